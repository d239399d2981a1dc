timeout 300 python tests/gpu_debug.py basic soup crown > gpurun_out/debug17.log 2>&1; echo EXIT $? >> gpurun_out/debug17.log
for g in 8 16 24 32; do MI355_REFILL_MIN=$g timeout 200 python tests/gpu_perf.py --reps 10 --tag "v5h-G$g" >> gpurun_out/perf17.log 2>&1; done
MI355_REFILL_MIN=16 timeout 200 python tests/gpu_perf.py --reps 10 --tag "v5h-G16-any" --any >> gpurun_out/perf17.log 2>&1
