#!/bin/bash
# round 6, last session: ray records read (1) and hit records written (2) with the non-temporal hint (tools/patches/r06_ray_nt.patch) -- data that is used once should not take the caches from the tree
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06_raynt; mkdir -p $O; rm -rf $O/*
for round in 1 2; do for V in base raynt1 raynt2; do
  LIBV=embree_amd/lib/variant_$V.so; [ $V = base ] && LIBV=embree_amd/lib/libembree4_mi355.so
  MI355_LIB=$LIBV timeout 300 python tests/gpu_batch_sweep.py --lo 17 --hi 21 --reps 30 --tag $V 2>&1 | grep -a "SWEEP\|rror\|fault" >> $O/sweep.log
done; done
for V in base raynt1 raynt2 base raynt1 raynt2; do
  LIBV=embree_amd/lib/variant_$V.so; [ $V = base ] && LIBV=embree_amd/lib/libembree4_mi355.so
  MI355_LIB=$LIBV timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu --sustain 0 > $O/bench_$V.json 2> $O/bench_$V.err
  python - <<PY >> $O/bench.log
import json
try:
    d=json.loads(open("$O/bench_$V.json").read().strip().splitlines()[-1]); print("$V value %.0f pipelined %.0f small %s" % (d["value"], d["pipelined"]["value"], [(l["rays"], l["us"]) for l in d["small_batch"]["legs"]]))
except Exception as e: print("$V bench failed", e)
PY
done
cat $O/sweep.log; cat $O/bench.log
