#!/bin/bash
# round 6: the radix sort of the Morton build (build_sort.inl) -- against a stable argsort, the LOW trees' hashes (must stay 97a5042e.../617a8afd... and d090e12d.../aeaee02e...), the LOW parity tests, kernel times
cd "$GRAFT_REPO_ROOT" || exit 1
R=$PWD; O=gpurun_out/r06o; mkdir -p $O; rm -rf $O/*
timeout 600 python -m pytest tests/test_gpu_round6.py -q -s --capture=sys -m gpu -k "radix_sort or library_kernel" > $O/sort.log 2>&1; grep -a "passed\|failed\|radix sort of\|^E " $O/sort.log | head -30
TREEHASH=1 timeout 300 python tests/gpu_build_only.py "" 6 0 2>&1 | grep -E "TREEHASH|BUILD|rror|fault" >> $O/ab.log
PP=1 TREEHASH=1 timeout 300 python tests/gpu_build_only.py "" 4 0 2>&1 | grep -E "TREEHASH|BUILD|rror|fault" >> $O/ab.log
cat $O/ab.log
timeout 900 python -m pytest tests -q -m gpu -k "low_quality or morton or lbvh or LOW or quality" 2>&1 | tail -3
( cd /tmp && export TMPDIR=/tmp && rm -rf $R/$O/prof_l && rocprofv3 --kernel-trace --stats -d $R/$O/prof_l -o commit -- python $R/tests/gpu_build_only.py "" 6 0 > $R/$O/prof_l.log 2>&1 )
python tools/ktimeline.py $O/prof_l v > $O/commit_timeline_low.txt 2>&1; tail -22 $O/commit_timeline_low.txt
