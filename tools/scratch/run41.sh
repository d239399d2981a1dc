timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -8
timeout 300 python tests/gpu_perf.py --reps 5 --tag morton --low 2>&1 | grep PERF | cut -c1-330
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/prof41 -o r41 -- python /root/repo/tests/gpu_perf.py --reps 2 --low > /dev/null 2>&1
cd /root/repo; python tools/kstats.py gpurun_out/prof41 | head -16
