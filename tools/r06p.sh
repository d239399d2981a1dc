#!/bin/bash
# round 6: lanes that wait for a hand-out block help before the cursors are dry (MI355_TRACE_HELPERS=k: once k lanes are free) -- batch sweep 2^17 .. 2^20 per setting; the sort tests again
cd "$GRAFT_REPO_ROOT" || exit 1
R=$PWD; O=gpurun_out/r06p; mkdir -p $O; rm -rf $O/*
timeout 600 python -m pytest tests/test_gpu_round6.py -q -m gpu -k "radix_sort or library_kernel" 2>&1 | tail -2
for H in 1 2 4 8 16 32; do
  echo "== MI355_TRACE_HELPERS=$H" >> $O/sweep.log
  MI355_TRACE_HELPERS=$H timeout 600 python tests/gpu_batch_sweep.py --lo 17 --hi 20 --md --tag h$H 2>&1 | grep -a "^|\|SWEEP\|rror" >> $O/sweep.log
done
cat $O/sweep.log
