for g in 16 24 32; do for p in 3 5; do MI355_REFILL_MIN=$g MI355_PUSH_ROUNDS=$p timeout 200 python tests/gpu_perf.py --reps 10 --tag "v5g-G$g-push$p" >> gpurun_out/perf16.log 2>&1; done; done
