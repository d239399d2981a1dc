for c in 1 8 1 8 2 4; do MI355_NUM_CURSORS=$c MI355_REFILL_MIN=32 timeout 200 python tests/gpu_perf.py --reps 5 --tag "v5c-cursors$c" >> gpurun_out/perf9.log 2>&1; done
