#!/bin/bash
# round 6, session 5: runs of 2^k consecutive 16-ray blocks go to the same cursor (= XCD): MI355_CURSOR_SUPER = 0 (ships: block by block) .. 10
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06zt; mkdir -p $O; rm -rf $O/*
for P in 0 4 6 8 10 0; do
  echo "== MI355_CURSOR_SUPER=$P" >> $O/sweep.log
  MI355_CURSOR_SUPER=$P timeout 600 python tests/gpu_batch_sweep.py --lo 17 --hi 21 --md --tag super$P 2>&1 | grep -a "^|\|SWEEP\|rror\|fault\|differ" >> $O/sweep.log
done
cat $O/sweep.log
