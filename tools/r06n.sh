#!/bin/bash
# round 6: small_threshold 512 + margin 10, SMALL kernel instantiations: hashes, commit times, md5 of the bench rays, the GPU suite, PMC of the trace kernel again
cd "$GRAFT_REPO_ROOT" || exit 1
R=$PWD; O=gpurun_out/r06n; mkdir -p $O; rm -f $O/*
TREEHASH=1 timeout 300 python tests/gpu_build_only.py "" 8 2>&1 | grep -E "TREEHASH|BUILD|rror|fault" >> $O/ab.log
PP=1 TREEHASH=1 timeout 300 python tests/gpu_build_only.py "" 5 2>&1 | grep -E "TREEHASH|BUILD|rror|fault" >> $O/ab.log
TREEHASH=1 timeout 300 python tests/gpu_build_only.py "" 4 2 2>&1 | grep -E "TREEHASH|BUILD|rror|fault" >> $O/ab.log
cat $O/ab.log
( time python -m pytest tests -q --capture=sys -m gpu ) > $O/full.log 2>&1; grep -a -v "^  File\|Extension modules" $O/full.log | tail -6
bash tools/profile_round.sh r06 > $O/profile_round.log 2>&1; tail -12 $O/profile_round.log
timeout 900 python tests/gpu_batch_sweep.py --lo 12 --hi 20 --md > $O/sweep.log 2>&1; grep -a "^|\|SWEEP" $O/sweep.log
