#!/bin/bash
# round 6, last session: the GPU suite and the driver's bench command on the tree with the streaming-store packing
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06_final2; mkdir -p $O; rm -rf $O/*
timeout 1500 python -m pytest tests -m gpu -q -x > $O/suite.log 2>&1; echo "suite rc=$?"; tail -1 $O/suite.log
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2> $O/bench_driver_cmd.err; echo "bench rc $?"
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r06_final2/bench_driver_cmd.json").read().strip().splitlines()[-1])
print("value", d["value"], "pipelined", d["pipelined"]["value"], "e2e", d["end_to_end"]["value"], d["end_to_end"]["ms"], "build", d["build"]["gpu_build_ms"])
PY
