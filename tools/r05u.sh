#!/bin/bash
O=gpurun_out/r05u; mkdir -p $O; rm -f $O/*.log
for rep in 1 2; do for b in 16 14 12 10; do
  MI355_TRACE_BLOCKS_PER_CU=$b timeout 90 python tests/gpu_knobs.py blocks$b 2>&1 | grep KNOBS | cut -c1-100 >> $O/knobs.log
done; done
cat $O/knobs.log
