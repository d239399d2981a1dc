#!/bin/bash
# round 6, last session: all five BASELINE.json configs at kernel level on the final tree (tests/gpu_configs.py) + the driver's bench command (end_to_end: best of 9)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06_configs; mkdir -p $O; rm -rf $O/*
timeout 900 python tests/gpu_configs.py > $O/configs.md 2> $O/configs.err; echo "configs rc=$?"
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2> $O/bench_driver_cmd.err; echo "bench rc $?"
cat $O/configs.md | cut -c1-220
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r06_configs/bench_driver_cmd.json").read().strip().splitlines()[-1])
print("value", d["value"], "pipelined", d["pipelined"]["value"], "e2e", d["end_to_end"]["value"], d["end_to_end"]["ms"], d["end_to_end"].get("ms_median"), "build", d["build"]["gpu_build_ms"])
PY
