# build-phase A/B: GPU parity suite, then kernel-level build timing
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r20_pytest.log
tail -3 gpurun_out/r20_pytest.log
timeout 300 python tests/gpu_perf.py --reps 5 --tag micro > gpurun_out/r20_perf.log 2>&1
cat gpurun_out/r20_perf.log | tail -4
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/prof20 -o r20 -- python /root/repo/tests/gpu_perf.py --reps 3 > /root/repo/gpurun_out/r20_prof.log 2>&1
cd /root/repo
python - <<'PY'
import csv,glob
for f in glob.glob('gpurun_out/prof20/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        print("%-60s calls %5s total_us %12.1f avg_us %10.1f %6s%%" % (r['Name'][:60], r['Calls'], float(r['TotalDurationNs'])/1e3, float(r['AverageNs'])/1e3, r['Percentage']))
PY
