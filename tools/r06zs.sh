#!/bin/bash
# round 6, session 5: ray blocks dealt to the cursors (= XCDs) interleaved (ships) against contiguous eighths of the batch (MI355_CURSOR_CONTIG=1)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06zs; mkdir -p $O; rm -rf $O/*
for P in 0 1 0 1; do
  echo "== MI355_CURSOR_CONTIG=$P" >> $O/sweep.log
  MI355_CURSOR_CONTIG=$P timeout 600 python tests/gpu_batch_sweep.py --lo 17 --hi 21 --md --tag contig$P 2>&1 | grep -a "^|\|SWEEP\|rror\|fault\|differ" >> $O/sweep.log
done
cat $O/sweep.log
