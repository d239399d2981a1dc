#!/bin/bash
# round 6, session 5: the first P % of every cursor's blocks are a contiguous part of the batch, the rest is dealt block by block (MI355_CURSOR_CONTIG_PCT; 0 ships)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06zu; mkdir -p $O; rm -rf $O/*
for P in 0 50 70 85 95 0; do
  echo "== MI355_CURSOR_CONTIG_PCT=$P" >> $O/sweep.log
  MI355_CURSOR_CONTIG_PCT=$P timeout 600 python tests/gpu_batch_sweep.py --lo 18 --hi 21 --md --tag pct$P 2>&1 | grep -a "^|\|SWEEP\|rror\|fault\|differ" >> $O/sweep.log
done
cat $O/sweep.log
