#!/bin/bash
# round 5, GPU call: counters and kernel statistics of the final kernels (profiles/r05_*), after the tests of this round's new pieces
O=gpurun_out/r05p; mkdir -p $O
R=$PWD
timeout 900 python -m pytest tests/test_gpu_round5.py -m gpu -q --durations=6 2>&1 | tail -14 > $O/pytest5.log; cat $O/pytest5.log | cut -c1-220
timeout 200 python tests/gpu_devfilter.py > $O/devfilter.log 2>&1; grep -E "DEVFILTER|calls|rror" $O/devfilter.log
# 1. FETCH_SIZE on known access patterns
( cd /tmp && export TMPDIR=/tmp && timeout 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/$O/calib -o calib -- $R/tools/fetch_calib > $R/$O/calib.log 2>&1 ); tail -1 $O/calib.log
python - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(list)
for p in glob.glob('gpurun_out/r05p/calib/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(p)):
        if r['Counter_Name'] == 'FETCH_SIZE': agg[r['Kernel_Name'].split('(')[0]].append(float(r['Counter_Value']))
known = {'stream16': 4 << 30, 'scatter16': (4 << 20) * 16, 'node80': (2 << 20) * 80}
for k, v in agg.items():
    kb = sum(v[1:]) / max(1, len(v) - 1)
    name = [n for n in known if n in k]
    if name: print("CALIB %-10s FETCH_SIZE %.4g KiB per launch = %.4g bytes; asked for %d bytes -> counter / asked = %.3f" % (name[0], kb, kb * 1024, known[name[0]], kb * 1024 / known[name[0]]))
PY
# 2. the bench command: kernel statistics + PMC passes of the closest-hit kernel (-> profiles/pmc_bench_latest.json)
bash tools/profile_round.sh r05 2>&1 | tail -3
# 3. PMC passes of the any-hit kernel on a configs[3] shard and of the closest-hit kernel on configs[4]
tools/pmc_run.sh $O/pmc_any python $R/tests/gpu_perf.py --shadow --reps 6 > $O/pmc_any.log 2>&1
python tools/pmc_summary.py $O/pmc_any "trace_kernel_q<true, false, false, false, 0>" $O/pmc_trace_any > $O/pmc_any_summary.log 2>&1; tail -3 $O/pmc_any_summary.log
tools/pmc_run.sh $O/pmc_pp python $R/tests/gpu_perf.py --powerplant --reps 6 > $O/pmc_pp.log 2>&1
python tools/pmc_summary.py $O/pmc_pp "trace_kernel_q<false, false, false, false, 0>" $O/pmc_trace_powerplant > $O/pmc_pp_summary.log 2>&1; tail -3 $O/pmc_pp_summary.log
timeout 100 python tests/gpu_perf.py --shadow --tag shadow 2>&1 | tail -2 > $O/perf_cfg.log
timeout 150 python tests/gpu_perf.py --powerplant --tag powerplant 2>&1 | tail -2 >> $O/perf_cfg.log
timeout 100 python tests/gpu_perf.py --tag crown 2>&1 | tail -2 >> $O/perf_cfg.log
cat $O/perf_cfg.log | cut -c1-330
# 4. small_build counters + the commit launch by launch
tools/pmc_run.sh $O/pmc_build python $R/tests/gpu_build_only.py "" 4 > $O/pmc_build.log 2>&1
python tools/pmc_summary.py $O/pmc_build "small_build" $O/pmc_small_build > $O/pmc_build_summary.log 2>&1; tail -3 $O/pmc_build_summary.log
( cd /tmp && export TMPDIR=/tmp && timeout 200 rocprofv3 --kernel-trace -d $R/$O/ktrace -o kt -- python $R/tests/gpu_build_only.py "" 3 > $R/$O/ktrace.log 2>&1 )
python tools/ktimeline.py $O/ktrace v > $O/commit_timeline_medium.txt 2>&1; tail -32 $O/commit_timeline_medium.txt
