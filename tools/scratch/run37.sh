cd /tmp && export TMPDIR=/tmp
MI355_LIB=/root/repo/embree_amd/lib/variant_nomerge.so timeout 300 rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/prof37 -o r37 -- python /root/repo/tests/gpu_perf.py --reps 2 2>&1 | grep PERF | cut -c1-100
cd /root/repo; python tools/kstats.py gpurun_out/prof37 | grep -E "top_|small"
