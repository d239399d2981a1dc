#!/bin/bash
# round 6, session 6: the whole GPU suite and the driver's bench command on the tree of 3f81eea
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06zz; mkdir -p $O; rm -rf $O/*
timeout 1500 python -m pytest tests -m gpu -x -q > $O/suite.log 2>&1; echo "suite rc=$?" >> $O/suite.log
tail -5 $O/suite.log
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
tail -c 3000 $O/bench.json
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -3 $O/smoke.log
