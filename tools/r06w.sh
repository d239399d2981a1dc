#!/bin/bash
# round 6: copies of the upper levels' bounds records (accTop) -- tree hashes of the four builds, times, timeline of the MEDIUM commit
cd "$GRAFT_REPO_ROOT" || exit 1
R=$PWD; O=gpurun_out/r06w; mkdir -p $O; rm -rf $O/*
TREEHASH=1 timeout 300 python tests/gpu_build_only.py "" 8 2>&1 | grep -E "TREEHASH|BUILD|rror|fault" >> $O/ab.log
TREEHASH=1 timeout 300 python tests/gpu_build_only.py "" 5 2 2>&1 | grep -E "TREEHASH|BUILD|rror|fault" >> $O/ab.log
PP=1 TREEHASH=1 timeout 300 python tests/gpu_build_only.py "" 5 2>&1 | grep -E "TREEHASH|BUILD|rror|fault" >> $O/ab.log
PP=1 TREEHASH=1 timeout 300 python tests/gpu_build_only.py "" 3 2 2>&1 | grep -E "TREEHASH|BUILD|rror|fault" >> $O/ab.log
cat $O/ab.log
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d $R/$O/prof_m -o commit -- python $R/tests/gpu_build_only.py "" 6 > $R/$O/prof_m.log 2>&1 )
python tools/ktimeline.py $O/prof_m v > $O/commit_timeline_medium.txt 2>&1; grep -n "top_partition\|top_bin" $O/commit_timeline_medium.txt | head -12 | awk '{print $3, $6}' | paste - - - -
tail -30 $O/commit_timeline_medium.txt | head -8
