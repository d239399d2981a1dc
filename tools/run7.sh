timeout 120 tools/valu_bench > gpurun_out/valu_bench.log 2>&1
for r in 16 32 48; do MI355_REFILL_MIN=$r timeout 200 python tests/gpu_perf.py --reps 5 --tag "v5-refill$r" >> gpurun_out/perf7.log 2>&1; done
for b in 2 3 4; do MI355_REFILL_MIN=32 MI355_TRACE_BLOCKS_PER_CU=$b timeout 200 python tests/gpu_perf.py --reps 5 --tag "v5-r32-bpc$b" >> gpurun_out/perf7.log 2>&1; done
