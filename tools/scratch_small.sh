#!/bin/bash
# small_build / top phase time against small_threshold (rocprofv3 kernel trace per variant)
R=$PWD
for th in 128 256 512 1024 2048; do
  ( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d $R/gpurun_out/sm_$th -o sm -- python $R/tests/gpu_build_only.py "small_threshold=$th" 4 > $R/gpurun_out/sm_$th.log 2>&1 )
  tail -1 gpurun_out/sm_$th.log
  python tools/kstats.py gpurun_out/sm_$th 2>/dev/null | grep -E "small_build|top_bin|top_partition|wide_plan" | cut -c1-90
done
