timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
timeout 300 python tests/gpu_perf.py --reps 5 --tag scan 2>&1 | grep PERF | cut -c1-230

cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/prof26 -o r26 -- python /root/repo/tests/gpu_perf.py --reps 3 > /root/repo/gpurun_out/r26_prof.log 2>&1
cd /root/repo
python tools/kstats.py gpurun_out/prof26
