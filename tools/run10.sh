timeout 300 python tests/gpu_debug.py basic soup crown > gpurun_out/debug10.log 2>&1; echo EXIT $? >> gpurun_out/debug10.log
for c in 1; do MI355_NUM_CURSORS=$c MI355_REFILL_MIN=32 timeout 200 python tests/gpu_perf.py --reps 5 --tag "v5d-cull" >> gpurun_out/perf10.log 2>&1; done
MI355_NUM_CURSORS=1 MI355_REFILL_MIN=32 timeout 200 python tests/gpu_perf.py --reps 5 --tag "v5d-cull-primary" --primary >> gpurun_out/perf10.log 2>&1
MI355_NUM_CURSORS=1 MI355_REFILL_MIN=32 timeout 200 python tests/gpu_perf.py --reps 5 --tag "v5d-cull-any" --any >> gpurun_out/perf10.log 2>&1
