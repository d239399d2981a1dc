#!/bin/bash
# round 6, last session: dry run of the driver's N = 2 and N = 8 commands on the 1-GPU box (ranks share the GPU, RCCL refuses duplicate GPUs: rccl_ranks 0) -- do they finish, is the line valid
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06_dry; mkdir -p $O; rm -rf $O/*
for N in 2 8; do
  ( time timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port $((29500 + N)) bench.py --gpus $N --steps 20 --warmup 5 > $O/bench_g$N.json 2> $O/bench_g$N.err ) 2>&1 | grep real
  python - <<PY
import json
try:
    d=json.loads(open("$O/bench_g$N.json").read().strip().splitlines()[-1])
    print("g$N value", d["value"], d["scaling"], "n_gpus", d["n_gpus"], "rccl", d["rccl_ranks"], "ms_per_step", d["ms_per_step"], "weak", d.get("weak", {}).get("value") if isinstance(d.get("weak"), dict) else d.get("weak"), "| metric:", d["metric"][:200])
    print("   step_timing", d.get("step_timing"), "gather", str(d.get("gather"))[:200])
except Exception as e: print("g$N parse failed", e)
PY
  tail -2 $O/bench_g$N.err | cut -c1-300
done
