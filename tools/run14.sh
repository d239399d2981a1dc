timeout 300 python tests/gpu_debug.py basic soup crown > gpurun_out/debug14.log 2>&1; echo EXIT $? >> gpurun_out/debug14.log
for r in 8 16 32; do MI355_REFILL_MIN=$r timeout 200 python tests/gpu_perf.py --reps 5 --tag "v5f-G$r" >> gpurun_out/perf14.log 2>&1; done
MI355_REFILL_MIN=16 MI355_NUM_CURSORS=1 timeout 200 python tests/gpu_perf.py --reps 5 --tag "v5f-G16-1cursor" >> gpurun_out/perf14.log 2>&1
MI355_REFILL_MIN=16 MI355_PUSH_ROUNDS=3 timeout 200 python tests/gpu_perf.py --reps 5 --tag "v5f-G16-push3" >> gpurun_out/perf14.log 2>&1
MI355_REFILL_MIN=16 MI355_PUSH_ROUNDS=8 timeout 200 python tests/gpu_perf.py --reps 5 --tag "v5f-G16-push8" >> gpurun_out/perf14.log 2>&1
MI355_REFILL_MIN=16 MI355_TRACE_BLOCKS_PER_CU=4 timeout 200 python tests/gpu_perf.py --reps 5 --tag "v5f-G16-bpc4" >> gpurun_out/perf14.log 2>&1
