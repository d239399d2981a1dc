MI355_NUM_CURSORS=1 MI355_REFILL_MIN=32 timeout 200 python tests/gpu_perf.py --reps 5 --tag "v5d-retrace" --retrace >> gpurun_out/perf11.log 2>&1
