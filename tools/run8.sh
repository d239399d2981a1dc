/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -w -o /tmp/valu_bench tools/valu_bench.hip && timeout 120 /tmp/valu_bench > gpurun_out/valu_bench.log 2>&1
timeout 300 python tests/gpu_debug.py basic soup > gpurun_out/debug8.log 2>&1; echo EXIT $? >> gpurun_out/debug8.log
for r in 8 16 24 32; do MI355_REFILL_MIN=$r timeout 200 python tests/gpu_perf.py --reps 5 --tag "v5c-refill$r" >> gpurun_out/perf8.log 2>&1; done
for p in 4 6; do MI355_REFILL_MIN=16 MI355_PUSH_ROUNDS=$p timeout 200 python tests/gpu_perf.py --reps 5 --tag "v5c-r16-push$p" >> gpurun_out/perf8.log 2>&1; done
