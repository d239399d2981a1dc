#!/bin/bash
O=gpurun_out/r03o; mkdir -p $O
for args in "--config top_splits=0 --tag nosplit" "--tag topsplit" "--config top_splits=0 --tess-room 4 --tag tess4" "--config top_splits=0 --tess-room 12 --tag tess12" "--config top_splits=0 --tess-room 32 --tag tess32" "--config top_splits=0 --tess-room 96 --tag tess96"; do
  timeout 300 python tests/gpu_perf.py $args --reps 6 2>&1 | grep -E "PERF|rror" | tee -a $O/sweep.log
done
