timeout 300 python tests/gpu_debug.py basic > gpurun_out/debug4.log 2>&1; echo EXIT $? >> gpurun_out/debug4.log
for cfg in "" "int_cost=0.5" "int_cost=0.3" "int_cost=2" "max_leaf=1" "max_leaf=2,int_cost=0.5" "trav_cost=0.3" "trav_cost=3"; do
  timeout 200 python tests/gpu_perf.py --reps 5 --tag "cw-v1" --config "$cfg" >> gpurun_out/perf4.log 2>&1
done
timeout 200 python tests/gpu_perf.py --reps 5 --tag "cw-v1-primary" --primary >> gpurun_out/perf4.log 2>&1
timeout 200 python tests/gpu_perf.py --reps 5 --tag "cw-v1-any" --any >> gpurun_out/perf4.log 2>&1
for b in 2 3 4; do MI355_TRACE_BLOCKS_PER_CU=$b timeout 200 python tests/gpu_perf.py --reps 5 --tag "cw-v1-bpc$b" >> gpurun_out/perf4.log 2>&1; done
tools/pmc_run.sh gpurun_out/pmc2 python tests/gpu_perf.py --reps 3 --tag pmc > gpurun_out/pmc2.log 2>&1
