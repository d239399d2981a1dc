#!/bin/bash
# round 6, session 5: small_build takes its sub-trees largest first (small_order) -- A/B against the list as it is, tree hashes, timeline
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06zp; mkdir -p $O; rm -rf $O/*
export TMPDIR=/tmp
{
echo "== list as it is"; MI355_SMALL_ORDER=0 TREEHASH=1 timeout 300 python tests/gpu_build_only.py "" 8 2>&1 | grep -a "TREEHASH\|BUILD\|rror\|fault"
echo "== largest first"; TREEHASH=1 timeout 300 python tests/gpu_build_only.py "" 8 2>&1 | grep -a "TREEHASH\|BUILD\|rror\|fault"
echo "== HIGH as it is"; MI355_SMALL_ORDER=0 TREEHASH=1 timeout 300 python tests/gpu_build_only.py "" 5 2 2>&1 | grep -a "TREEHASH\|BUILD\|rror\|fault"
echo "== HIGH largest first"; TREEHASH=1 timeout 300 python tests/gpu_build_only.py "" 5 2 2>&1 | grep -a "TREEHASH\|BUILD\|rror\|fault"
echo "== PP as it is"; MI355_SMALL_ORDER=0 PP=1 TREEHASH=1 timeout 300 python tests/gpu_build_only.py "" 4 2>&1 | grep -a "TREEHASH\|BUILD\|rror\|fault"
echo "== PP largest first"; PP=1 TREEHASH=1 timeout 300 python tests/gpu_build_only.py "" 4 2>&1 | grep -a "TREEHASH\|BUILD\|rror\|fault"
} > $O/ab.log 2>&1
cat $O/ab.log
( cd /tmp && timeout 300 rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/$O/prof -o medium -- python $GRAFT_REPO_ROOT/tests/gpu_build_only.py "" 6 > $GRAFT_REPO_ROOT/$O/prof.log 2>&1 )
python tools/ktimeline.py $O/prof v > $O/timeline_medium.txt 2>&1
grep -a "small_build\|small_order\|commit:" $O/timeline_medium.txt
rm -rf $O/prof
