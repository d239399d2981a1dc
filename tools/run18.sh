for cfg in "min_leaf=2" "min_leaf=3" "min_leaf=2,int_cost=0.3" "int_cost=0.2" ; do timeout 200 python tests/gpu_perf.py --reps 10 --tag "v5h" --config "$cfg" >> gpurun_out/perf18.log 2>&1; done
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/prof18 -o b -- python /root/repo/tests/gpu_perf.py --reps 2 --tag prof > /root/repo/gpurun_out/prof18.log 2>&1
