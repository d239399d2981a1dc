#!/bin/bash
# round 6: binning in pairs of lanes (bins_add_copies) -- tree hashes, commit times with and without, the timeline
cd "$GRAFT_REPO_ROOT" || exit 1
R=$PWD; O=gpurun_out/r06s; mkdir -p $O; rm -rf $O/*
for V in product nopairs; do
  echo "== $V" >> $O/ab.log
  L=$R/embree_amd/lib/variant_$V.so; [ $V = product ] && L=$R/embree_amd/lib/libembree4_mi355.so
  MI355_LIB=$L TREEHASH=1 timeout 300 python tests/gpu_build_only.py "" 8 2>&1 | grep -E "TREEHASH|BUILD|rror|fault" >> $O/ab.log
  MI355_LIB=$L TREEHASH=1 PP=1 timeout 300 python tests/gpu_build_only.py "" 5 2>&1 | grep -E "TREEHASH|BUILD|rror|fault" >> $O/ab.log
  MI355_LIB=$L TREEHASH=1 timeout 300 python tests/gpu_build_only.py "" 4 2 2>&1 | grep -E "TREEHASH|BUILD|rror|fault" >> $O/ab.log
done
cat $O/ab.log
( cd /tmp && export TMPDIR=/tmp && rm -rf $R/$O/prof_m && rocprofv3 --kernel-trace --stats -d $R/$O/prof_m -o commit -- python $R/tests/gpu_build_only.py "" 6 > $R/$O/prof_m.log 2>&1 )
python tools/ktimeline.py $O/prof_m v > $O/commit_timeline_medium.txt 2>&1; tail -30 $O/commit_timeline_medium.txt | head -14; grep -n "top_bin" $O/commit_timeline_medium.txt | head -14
