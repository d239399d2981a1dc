#!/bin/bash
# round 6, last session: the CPU packing of the host-array path with streaming stores (MI355_PACK_NT=1, default) against plain memcpy (=0)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06_nt; mkdir -p $O; rm -rf $O/*
for rep in 1 2 3; do for NT in 0 1; do
  MI355_PACK_NT=$NT timeout 200 python tests/gpu_e2e_time.py "gpu=0" 2>&1 | grep -a "E2E\|rror" | sed "s/^/nt=$NT /" >> $O/e2e.log
done; done
for NT in 0 1 0 1; do
  MI355_PACK_NT=$NT timeout 400 python bench.py --steps 10 --warmup 3 --sustain 0 > $O/b_$NT.json 2> $O/b_$NT.err
  python - <<PY >> $O/e2e.log
import json
d=json.loads(open("$O/b_$NT.json").read().strip().splitlines()[-1]); e=d["end_to_end"]
print("bench nt=$NT: end_to_end %.1f Mrays/s %.3f ms" % (e["value"], e["ms"]))
PY
done
cat $O/e2e.log
