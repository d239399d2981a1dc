#!/bin/bash
# round 6, session 5: wide_plan without ds_bpermute in its two chains (payloads travel with the argmax)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06zo; mkdir -p $O; rm -rf $O/*
export TMPDIR=/tmp
{
TREEHASH=1 timeout 300 python tests/gpu_build_only.py "" 8 2>&1 | grep -a "TREEHASH\|BUILD\|rror\|fault"
TREEHASH=1 timeout 300 python tests/gpu_build_only.py "" 4 2 2>&1 | grep -a "TREEHASH\|BUILD\|rror\|fault"
TREEHASH=1 timeout 300 python tests/gpu_build_only.py "" 4 1 2>&1 | grep -a "TREEHASH\|BUILD\|rror\|fault"
PP=1 TREEHASH=1 timeout 300 python tests/gpu_build_only.py "" 4 2>&1 | grep -a "TREEHASH\|BUILD\|rror\|fault"
} > $O/hashes.log 2>&1
cat $O/hashes.log
( cd /tmp && timeout 300 rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/$O/prof -o medium -- python $GRAFT_REPO_ROOT/tests/gpu_build_only.py "" 6 > $GRAFT_REPO_ROOT/$O/prof.log 2>&1 )
python tools/ktimeline.py $O/prof v > $O/timeline_medium.txt 2>&1
tail -45 $O/timeline_medium.txt
rm -rf $O/prof
timeout 600 python -m pytest tests -m gpu -x -q -k "garbage or coincident or rebuild or outlier or powerplant_full or refit or invalid" 2>&1 | tail -5
