#!/bin/bash
O=gpurun_out/r03e; mkdir -p $O; R=$PWD
python tests/gpu_build_only.py "" 6 2>&1 | tee $O/build_default.log
python tests/gpu_build_only.py "top_splits=0" 6 2>&1 | tee $O/build_nosplit.log
python tests/gpu_build_only.py "top_split_min=262144" 6 2>&1 | tee $O/build_256k.log
MI355_BUILD_GRAPH=0 python tests/gpu_build_only.py "" 6 2>&1 | tee $O/build_nograph.log
( cd /tmp && export TMPDIR=/tmp && MI355_BUILD_GRAPH=0 rocprofv3 --kernel-trace --stats -d $R/$O/prof_split -o p -- python $R/tests/gpu_build_only.py "" 5 > $R/$O/prof_split.log 2>&1 )
python tools/kstats.py $O/prof_split > $O/kstats_split.md 2>&1; head -40 $O/kstats_split.md
( cd /tmp && export TMPDIR=/tmp && MI355_BUILD_GRAPH=0 rocprofv3 --kernel-trace --stats -d $R/$O/prof_nosplit -o p -- python $R/tests/gpu_build_only.py "top_splits=0" 5 > $R/$O/prof_nosplit.log 2>&1 )
python tools/kstats.py $O/prof_nosplit > $O/kstats_nosplit.md 2>&1; head -30 $O/kstats_nosplit.md
rm -rf $O/prof_split $O/prof_nosplit
