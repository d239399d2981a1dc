#!/bin/bash
# round 6, closing run: the GPU suite, smoke and the default bench on the last commit of the session
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06_final3; mkdir -p $O; rm -rf $O/*
timeout 1500 python -m pytest tests -m gpu -q -x > $O/suite.log 2>&1; echo "suite rc=$?"; tail -1 $O/suite.log
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc $?"
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r06_final3/bench_default.json").read().strip().splitlines()[-1])
print("value", d["value"], "steps", d["steps"], "pipelined", d["pipelined"]["value"], "e2e", d["end_to_end"]["value"], d["end_to_end"]["ms"], d["end_to_end"].get("ms_median"), "build", d["build"]["gpu_build_ms"], "frac", d["roofline"]["frac"])
PY
