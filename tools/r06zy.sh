#!/bin/bash
# round 6, session 5: the end_to_end leg INSIDE bench.py (the reference's worker pool of the cpu_baseline leg is alive in the process): packed / whole records x copy threads
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06zy; mkdir -p $O; rm -rf $O/*
for V in "1 2" "1 6" "0 6" "0 2" "1 12"; do set -- $V
  MI355_PACKED_LINK=$1 MI355_COPY_THREADS=$2 timeout 600 python bench.py --steps 10 --warmup 3 --sustain 0 > $O/b_$1_$2.json 2> $O/b_$1_$2.err
  python - <<PY >> $O/e2e.log
import json
d=json.loads(open("$O/b_$1_$2.json").read().strip().splitlines()[-1]); e=d["end_to_end"]
print("packed=$1 threads=$2: end_to_end %.1f Mrays/s %.3f ms | value %.0f" % (e["value"], e["ms"], d["value"]))
PY
done
cat $O/e2e.log
