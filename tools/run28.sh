timeout 900 python bench.py > gpurun_out/bench3.json 2> gpurun_out/bench3.err; tail -2 gpurun_out/bench3.err; cat gpurun_out/bench3.json
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/prof28 -o r28 -- python /root/repo/bench.py --steps 12 --warmup 2 --no-cpu > /root/repo/gpurun_out/bench_prof3.json 2> /root/repo/gpurun_out/bench_prof3.err
timeout 600 rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/prof28s -o r28s -- python /root/repo/bench.py --steps 12 --warmup 2 --no-cpu --streams 1 > /root/repo/gpurun_out/bench_prof3s.json 2> /root/repo/gpurun_out/bench_prof3s.err
cd /root/repo
cat gpurun_out/bench_prof3.json; python tools/kstats.py gpurun_out/prof28 | head -8
cat gpurun_out/bench_prof3s.json; python tools/kstats.py gpurun_out/prof28s | head -8
