timeout 300 python tests/gpu_debug.py basic soup crown > gpurun_out/debug6.log 2>&1; echo EXIT $? >> gpurun_out/debug6.log
for v in 4 5; do MI355_TRACE_VARIANT=$v timeout 200 python tests/gpu_perf.py --reps 5 --tag "variant$v" >> gpurun_out/perf6.log 2>&1; done
for r in 8 16 32; do MI355_REFILL_MIN=$r timeout 200 python tests/gpu_perf.py --reps 5 --tag "v5-refill$r" >> gpurun_out/perf6.log 2>&1; done
timeout 200 python tests/gpu_perf.py --reps 5 --tag "v5-any" --any >> gpurun_out/perf6.log 2>&1
timeout 200 python tests/gpu_perf.py --reps 5 --tag "v5-primary" --primary >> gpurun_out/perf6.log 2>&1
