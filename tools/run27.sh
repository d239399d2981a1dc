run() { echo "== $*"; env "$@" timeout 300 python tests/gpu_overlap.py 2>&1 | grep -E "streams=(1|4)" | sed 's/launches of 1048576 rays in//'; }
run X=0
run MI355_REFILL_MIN=16
run MI355_REFILL_MIN=24
run MI355_REFILL_MIN=48
run MI355_PUSH_ROUNDS=3
run MI355_PUSH_ROUNDS=8
run MI355_TRACE_BLOCKS_PER_CU=4
run MI355_TRACE_BLOCKS_PER_CU=3
