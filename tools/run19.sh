timeout 900 python bench.py --steps 20 --warmup 3 > gpurun_out/bench2.json 2> gpurun_out/bench2.err
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/prof19 -o r01b -- python /root/repo/bench.py --steps 10 --warmup 2 --no-cpu > /root/repo/gpurun_out/bench_prof2.json 2> /root/repo/gpurun_out/bench_prof2.err
cd /root/repo
tools/pmc_run.sh gpurun_out/pmc4 python /root/repo/bench.py --steps 5 --warmup 1 --no-cpu > gpurun_out/pmc4.log 2>&1
