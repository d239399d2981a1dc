#!/bin/bash
R=$PWD
for v in "$@"; do
  ( cd /tmp && export TMPDIR=/tmp && MI355_LIB=$R/embree_amd/lib/variant_$v.so MI355_BUILD_STEPWISE=1 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/sb_$v -o sb -- python $R/tests/gpu_build_only.py "" 2 > $R/gpurun_out/sb_$v.log 2>&1 )
  echo "$v: $(python tools/kstats.py gpurun_out/sb_$v 2>/dev/null | grep small_build | cut -c1-70)"
done
