#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06f; mkdir -p $O
python -m pytest tests/test_gpu_round3.py -x -v -m gpu -k "several_gpus" > $O/a.log 2>&1; tail -30 $O/a.log | grep -v "^  File\|Extension modules"
echo "=== static off"
MI355_STATIC_RAYS=0 python -m pytest tests/test_gpu_round3.py -x -v -m gpu -k "several_gpus" > $O/b.log 2>&1; tail -8 $O/b.log | grep -v "^  File\|Extension modules"
