#!/bin/bash
O=gpurun_out/r03f; mkdir -p $O; R=$PWD
for args in "--tag rel32" "--config top_split_rel=4 --tag rel4" "--config top_split_rel=256 --tag rel256" "--config top_split_rel=0.001 --tag relAll"; do
  timeout 300 python tests/gpu_perf.py $args --reps 6 2>&1 | grep -E "PERF|rror" | tee -a $O/sweep.log
done
python tests/gpu_build_only.py "" 6 2>&1 | tee $O/build_default.log
python tests/gpu_build_only.py "top_split_rel=4" 6 2>&1 | tee $O/build_rel4.log
PP=1 python tests/gpu_build_only.py "" 4 2>&1 | tee $O/build_pp.log
PP=1 python tests/gpu_build_only.py "top_splits=0" 4 2>&1 | tee $O/build_pp_nosplit.log
( cd /tmp && export TMPDIR=/tmp && MI355_BUILD_GRAPH=0 rocprofv3 --kernel-trace --stats -d $R/$O/prof_split -o p -- python $R/tests/gpu_build_only.py "" 5 > $R/$O/prof_split.log 2>&1 )
python tools/kstats.py $O/prof_split > $O/kstats_split.md 2>&1; head -14 $O/kstats_split.md
rm -rf $O/prof_split
