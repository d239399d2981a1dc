for t in 1 6 12 20 32 48; do MI355_TRI_MIN=$t timeout 200 python tests/gpu_perf.py --reps 5 --tag "trimin$t" >> gpurun_out/perf5.log 2>&1; done
for t in 1 8 32 64; do MI355_REFILL_MIN=$t timeout 200 python tests/gpu_perf.py --reps 5 --tag "refill$t" >> gpurun_out/perf5.log 2>&1; done
tools/pmc_run.sh gpurun_out/pmc2 python /root/repo/tests/gpu_perf.py --reps 3 --tag pmc > gpurun_out/pmc2.log 2>&1
