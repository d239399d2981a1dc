#!/bin/bash
# round 6: small batches on fewer waves than they could fill (refill instead of a fixed 64 rays per wave)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06b; mkdir -p $O
timeout 600 python tests/gpu_batch_sweep.py --lo 12 --hi 20 --tag auto > $O/sweep_auto.log 2>&1
for F in 4 5 6 7; do
  MI355_SMALL_FRAC8=$F timeout 600 python tests/gpu_batch_sweep.py --lo 16 --hi 18 --tag F$F > $O/sweep_F$F.log 2>&1
done
grep -h SWEEP $O/sweep_*.log
