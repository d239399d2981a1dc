#!/bin/bash
# kernel times of a RTC_BUILD_QUALITY_HIGH commit of the crown stand-in
R=$PWD
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d $R/gpurun_out/high -o high -- python $R/tests/gpu_build_only.py "" 3 2 > $R/gpurun_out/high.log 2>&1 )
grep BUILD gpurun_out/high.log
python tools/kstats.py gpurun_out/high 2>/dev/null | head -24 | cut -c1-100
