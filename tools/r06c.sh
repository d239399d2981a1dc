#!/bin/bash
# round 6: nano_build (one lane per set of <= 16 triangles) against the micro mode down to the leaves: tree hashes and commit times
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06c; mkdir -p $O; rm -f $O/ab.log
for NANO in 0 16 8; do
  echo "== MI355_BUILD_NANO=$NANO" >> $O/ab.log
  MI355_BUILD_NANO=$NANO TREEHASH=1 timeout 300 python tests/gpu_build_only.py "" 8 2>&1 | grep -E "TREEHASH|BUILD|rror|fault" >> $O/ab.log
  MI355_BUILD_NANO=$NANO TREEHASH=1 timeout 300 python tests/gpu_build_only.py "" 4 2 2>&1 | grep -E "TREEHASH|BUILD|rror|fault" >> $O/ab.log
  PP=1 MI355_BUILD_NANO=$NANO TREEHASH=1 timeout 300 python tests/gpu_build_only.py "" 4 2>&1 | grep -E "TREEHASH|BUILD|rror|fault" >> $O/ab.log
done
cat $O/ab.log
cd /tmp && export TMPDIR=/tmp
for NANO in 16 8; do
rm -rf $GRAFT_REPO_ROOT/$O/prof$NANO
MI355_BUILD_NANO=$NANO rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof$NANO -o commit -- python $GRAFT_REPO_ROOT/tests/gpu_build_only.py "" 6 > $GRAFT_REPO_ROOT/$O/prof$NANO.log 2>&1
( cd $GRAFT_REPO_ROOT; python tools/ktimeline.py $O/prof$NANO 2>&1 | head -8 )
done
