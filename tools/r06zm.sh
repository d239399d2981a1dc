#!/bin/bash
# round 6, session 5: how many cursors a wave asks before it calls the batch handed out (MI355_PROBE_LIMIT = 8 ships so far): batch sweep 2^17 .. 2^21
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06zm; mkdir -p $O; rm -rf $O/*
for P in 8 1 2 4; do
  echo "== MI355_PROBE_LIMIT=$P" >> $O/sweep.log
  MI355_PROBE_LIMIT=$P timeout 600 python tests/gpu_batch_sweep.py --lo 17 --hi 21 --md --tag probe$P 2>&1 | grep -a "^|\|SWEEP\|rror\|fault\|differ" >> $O/sweep.log
done
cat $O/sweep.log
