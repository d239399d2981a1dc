#!/bin/bash
# round 5, GPU call: A/B of trace.hip variants on one box: tools/r05w.sh <variant> <variant> ...   (embree_amd/lib/variant_<name>.so, tools/trace_variant.sh)
O=gpurun_out/r05w; mkdir -p $O; rm -f $O/*.log
for rep in 1 2; do for v in "$@"; do
  MI355_LIB=$PWD/embree_amd/lib/variant_$v.so timeout 90 python tests/gpu_knobs.py $v 2>&1 | grep KNOBS | cut -c1-100 >> $O/knobs.log
done; done
cat $O/knobs.log
