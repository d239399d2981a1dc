#!/bin/bash
# round 6: wide collapse with one scan entry per workgroup (trees must keep their hashes); the radix sort: tile sizes and what the look-back / the write-out cost
cd "$GRAFT_REPO_ROOT" || exit 1
R=$PWD; O=gpurun_out/r06q; mkdir -p $O; rm -rf $O/*
TREEHASH=1 timeout 300 python tests/gpu_build_only.py "" 8 2>&1 | grep -E "TREEHASH|BUILD|rror|fault" >> $O/ab.log
TREEHASH=1 timeout 300 python tests/gpu_build_only.py "" 6 0 2>&1 | grep -E "TREEHASH|BUILD|rror|fault" >> $O/ab.log
TREEHASH=1 timeout 300 python tests/gpu_build_only.py "" 4 2 2>&1 | grep -E "TREEHASH|BUILD|rror|fault" >> $O/ab.log
PP=1 TREEHASH=1 timeout 300 python tests/gpu_build_only.py "" 5 2>&1 | grep -E "TREEHASH|BUILD|rror|fault" >> $O/ab.log
cat $O/ab.log
timeout 120 python tests/gpu_sort_time.py 4762764 12 check 2>&1 | grep -a "SORT\|rror" >> $O/sort.log
KEYS=random timeout 120 python tests/gpu_sort_time.py 4762764 12 check 2>&1 | grep -a "SORT\|rror" >> $O/sort.log
for V in k4 k12 k16; do MI355_LIB=$R/embree_amd/lib/variant_$V.so timeout 120 python tests/gpu_sort_time.py 4762764 12 check 2>&1 | grep -a "SORT\|rror" >> $O/sort.log; done
for V in nolook nowrite nolw; do MI355_LIB=$R/embree_amd/lib/variant_$V.so timeout 120 python tests/gpu_sort_time.py 4762764 12 2>&1 | grep -a "SORT\|rror" >> $O/sort.log; done
timeout 120 python tests/gpu_sort_time.py 64000000 5 2>&1 | grep -a "SORT\|rror" >> $O/sort.log
cat $O/sort.log
( cd /tmp && export TMPDIR=/tmp && rm -rf $R/$O/prof_m && rocprofv3 --kernel-trace --stats -d $R/$O/prof_m -o commit -- python $R/tests/gpu_build_only.py "" 6 > $R/$O/prof_m.log 2>&1 )
python tools/ktimeline.py $O/prof_m v > $O/commit_timeline_medium.txt 2>&1; tail -32 $O/commit_timeline_medium.txt
