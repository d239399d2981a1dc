#!/bin/bash
O=gpurun_out/r05m; mkdir -p $O; rm -f $O/leaf.log
for cfg in "" "min_leaf=3" "min_leaf=3,max_leaf=4" "min_leaf=1"; do
  timeout 100 python tests/gpu_build_only.py "$cfg" 6 2>&1 | grep BUILD >> $O/leaf.log
  CFG="$cfg" timeout 100 python tests/gpu_knobs.py "cfg:$cfg" 2>&1 | grep KNOBS | cut -c1-110 >> $O/leaf.log
done
cat $O/leaf.log | cut -c1-220
