#!/bin/bash
for m in 0 512 2048 8192 65536; do
  echo "== MI355_SPATIAL_MIN=$m"
  MI355_SPATIAL_MIN=$m PHI=158 python tests/gpu_sah.py 2>&1 | grep -E "mi355     HIGH  " | cut -c1-250
  MI355_SPATIAL_MIN=$m python tests/gpu_perf.py --high --tag high$m 2>&1 | grep PERF | cut -c1-60,120-330
done
