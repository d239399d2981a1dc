timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -2
timeout 300 python tests/gpu_perf.py --reps 3 --tag sweep4 2>&1 | grep PERF | cut -c1-120
timeout 300 python tests/gpu_perf.py --reps 3 --tag minleaf1 --config min_leaf=1 2>&1 | grep PERF | cut -c1-120
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/prof36 -o r36 -- python /root/repo/tests/gpu_perf.py --reps 2 > /dev/null 2>&1
cd /root/repo; python tools/kstats.py gpurun_out/prof36 | head -9
