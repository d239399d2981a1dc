for v in "" embree_amd/lib/variant_c1024.so embree_amd/lib/variant_c4096.so; do
echo "== $v"
env ${v:+MI355_LIB=/root/repo/$v} timeout 300 python tests/gpu_perf.py --reps 2 2>&1 | grep PERF | cut -c1-100
done
cd /tmp && export TMPDIR=/tmp
MI355_LIB=/root/repo/embree_amd/lib/variant_c1024.so timeout 300 rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/prof38 -o r38 -- python /root/repo/tests/gpu_perf.py --reps 2 2>&1 | grep PERF | cut -c1-100
cd /root/repo; python tools/kstats.py gpurun_out/prof38 | grep -E "top_|small"
