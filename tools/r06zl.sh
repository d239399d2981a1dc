#!/bin/bash
# round 6, session 5: top_setup with 64 sets and ONE chunk atomic per workgroup: tree hashes, times, HIGH timeline
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06zl; mkdir -p $O; rm -rf $O/*
export TMPDIR=/tmp
{
TREEHASH=1 timeout 300 python tests/gpu_build_only.py "" 8 2>&1 | grep -a "TREEHASH\|BUILD\|rror\|fault"
TREEHASH=1 timeout 300 python tests/gpu_build_only.py "" 6 2 2>&1 | grep -a "TREEHASH\|BUILD\|rror\|fault"
TREEHASH=1 timeout 300 python tests/gpu_build_only.py "" 4 0 2>&1 | grep -a "TREEHASH\|BUILD\|rror\|fault"
PP=1 TREEHASH=1 timeout 300 python tests/gpu_build_only.py "" 4 2>&1 | grep -a "TREEHASH\|BUILD\|rror\|fault"
PP=1 TREEHASH=1 timeout 300 python tests/gpu_build_only.py "" 3 2 2>&1 | grep -a "TREEHASH\|BUILD\|rror\|fault"
} > $O/hashes.log 2>&1
cat $O/hashes.log
( cd /tmp && timeout 300 rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/$O/prof -o high -- python $GRAFT_REPO_ROOT/tests/gpu_build_only.py "" 5 2 > $GRAFT_REPO_ROOT/$O/prof.log 2>&1 )
python tools/ktimeline.py $O/prof v > $O/timeline_high.txt 2>&1
tail -32 $O/timeline_high.txt
rm -rf $O/prof
timeout 600 python -m pytest tests -m gpu -x -q -k "garbage or coincident or rebuild or outlier or powerplant_full or refit or invalid or high or spatial or instance or refit or scene or commit" 2>&1 | tail -3
