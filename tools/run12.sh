timeout 300 python tests/gpu_debug.py basic soup > gpurun_out/debug12.log 2>&1; echo EXIT $? >> gpurun_out/debug12.log
for b in 5 4 3 2; do MI355_TRACE_BLOCKS_PER_CU=$b MI355_NUM_CURSORS=1 MI355_REFILL_MIN=32 timeout 200 python tests/gpu_perf.py --reps 5 --tag "v5e-bpc$b" >> gpurun_out/perf12.log 2>&1; done
for p in 4 6; do MI355_PUSH_ROUNDS=$p MI355_NUM_CURSORS=1 MI355_REFILL_MIN=32 timeout 200 python tests/gpu_perf.py --reps 5 --tag "v5e-push$p" >> gpurun_out/perf12.log 2>&1; done
