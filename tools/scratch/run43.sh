for c in "" "int_cost=0.5" "int_cost=2" "trav_cost=2" "trav_cost=4" "trav_cost=0.5" "max_leaf=2" "min_leaf=3,max_leaf=3" "sah_block_shift=1" "min_leaf=1,trav_cost=3"; do
timeout 300 python tests/gpu_perf.py --reps 6 --tag "x" --config "$c" 2>&1 | grep PERF | cut -c31-75,96-140,200-245
done
