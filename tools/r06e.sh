#!/bin/bash
# round 6: DPP block scans; then the whole GPU suite
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06e; mkdir -p $O; rm -f $O/ab.log
TREEHASH=1 timeout 300 python tests/gpu_build_only.py "" 8 2>&1 | grep -E "TREEHASH|BUILD|rror|fault" >> $O/ab.log
TREEHASH=1 timeout 300 python tests/gpu_build_only.py "" 4 2 2>&1 | grep -E "TREEHASH|BUILD|rror|fault" >> $O/ab.log
TREEHASH=1 timeout 300 python tests/gpu_build_only.py "" 4 0 2>&1 | grep -E "TREEHASH|BUILD|rror|fault" >> $O/ab.log
PP=1 TREEHASH=1 timeout 300 python tests/gpu_build_only.py "" 4 2>&1 | grep -E "TREEHASH|BUILD|rror|fault" >> $O/ab.log
cat $O/ab.log
( cd /tmp && export TMPDIR=/tmp && rm -rf $GRAFT_REPO_ROOT/$O/prof && rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof -o commit -- python $GRAFT_REPO_ROOT/tests/gpu_build_only.py "" 6 > $GRAFT_REPO_ROOT/$O/prof.log 2>&1 )
python tools/ktimeline.py $O/prof 2>&1 | head -34
( time python -m pytest tests -x -q -m gpu ) > $O/gpu_tests.log 2>&1; tail -6 $O/gpu_tests.log
