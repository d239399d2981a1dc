# round-1 final measurements of this session: bench line, rocprofv3 kernel stats of the same command, PMC passes (serial launches)
timeout 900 python bench.py > gpurun_out/bench4.json 2> gpurun_out/bench4.err; tail -2 gpurun_out/bench4.err
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/prof32 -o r32 -- python /root/repo/bench.py --no-cpu > /root/repo/gpurun_out/bench4_prof.json 2> /root/repo/gpurun_out/bench4_prof.err
cd /root/repo
python tools/kstats.py gpurun_out/prof32 > gpurun_out/prof32_kstats.md
tools/pmc_run.sh gpurun_out/pmc32 python /root/repo/bench.py --steps 5 --warmup 1 --no-cpu --streams 1 > gpurun_out/pmc32.log 2>&1
python tools/pmc_summary.py gpurun_out/pmc32 trace_kernel_q gpurun_out/pmc32_trace >> gpurun_out/pmc32.log 2>&1
python tools/pmc_summary.py gpurun_out/pmc32 small_build gpurun_out/pmc32_small >> gpurun_out/pmc32.log 2>&1
cat gpurun_out/bench4.json; head -12 gpurun_out/prof32_kstats.md; tail -5 gpurun_out/pmc32.log
