for c in 8 1 2 4; do MI355_NUM_CURSORS=$c MI355_REFILL_MIN=16 timeout 200 python tests/gpu_perf.py --reps 5 --tag "v5g-G16-cursors$c" >> gpurun_out/perf15.log 2>&1; done
