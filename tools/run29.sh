run() { echo "== $*"; env "$@" timeout 300 python tests/gpu_overlap.py 2>&1 | grep -E "streams=(1|4)" | sed 's/launches of 1048576 rays in//'; }
run EVENTS=0
run EVENTS=1
run EVENTS=1 MI355_EVENT_FLAGS=0
