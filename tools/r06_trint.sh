#!/bin/bash
# round 6, last session: triangle records fetched with the non-temporal hint (tools/patches/r06_tri_nt.patch, -DMI355_TRI_NT=1) against plain loads
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06_trint; mkdir -p $O; rm -rf $O/*
for round in 1 2; do for V in base trint; do
  LIBV=embree_amd/lib/variant_$V.so; [ $V = base ] && LIBV=embree_amd/lib/libembree4_mi355.so
  MI355_LIB=$LIBV timeout 300 python tests/gpu_batch_sweep.py --lo 16 --hi 21 --reps 30 --tag $V 2>&1 | grep -a "SWEEP\|rror\|fault" >> $O/sweep.log
done; done
for V in base trint base trint; do
  LIBV=embree_amd/lib/variant_$V.so; [ $V = base ] && LIBV=embree_amd/lib/libembree4_mi355.so
  MI355_LIB=$LIBV timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu --sustain 0 > $O/bench_$V.json 2> $O/bench_$V.err
  python - <<PY >> $O/bench.log
import json
try:
    d=json.loads(open("$O/bench_$V.json").read().strip().splitlines()[-1]); print("$V value %.0f pipelined %.0f build %.2f ms small %s bvh_bytes %s" % (d["value"], d["pipelined"]["value"], d["build"]["gpu_build_ms"], [(l["rays"], l["us"]) for l in d["small_batch"]["legs"]], d["build"].get("bvh_bytes")))
except Exception as e: print("$V bench failed", e)
PY
done
MI355_LIB=embree_amd/lib/variant_trint.so timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -2 >> $O/bench.log
cat $O/sweep.log; cat $O/bench.log
