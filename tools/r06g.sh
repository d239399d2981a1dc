#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06g; mkdir -p $O
( time python -m pytest tests -v -m gpu ) > $O/gpu_tests.log 2>&1; grep -v "^  File\|Extension modules" $O/gpu_tests.log | tail -15
