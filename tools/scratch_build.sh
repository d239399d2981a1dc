#!/bin/bash
# kernel times of MEDIUM and HIGH commits of the crown stand-in (in-tree library)
R=$PWD
for q in 0 2; do
  ( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d $R/gpurun_out/bq$q -o b -- python $R/tests/gpu_build_only.py "" 4 $( [ $q = 2 ] && echo 2 ) > $R/gpurun_out/bq$q.log 2>&1 )
  grep BUILD gpurun_out/bq$q.log
  python tools/kstats.py gpurun_out/bq$q 2>/dev/null | head -10 | cut -c1-90
done
