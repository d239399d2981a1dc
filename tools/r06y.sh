#!/bin/bash
# round 6, experiment: a pool of prepared rays per wave (MI355_POOL=1): batch sweep against the product, pop thresholds
cd "$GRAFT_REPO_ROOT" || exit 1
R=$PWD; O=gpurun_out/r06y; mkdir -p $O; rm -rf $O/*
echo "== product" >> $O/sweep.log
timeout 600 python tests/gpu_batch_sweep.py --lo 17 --hi 21 --md --tag product 2>&1 | grep -a "^|\|SWEEP\|rror\|fault" >> $O/sweep.log
for PM in 4 2 8 16; do
  echo "== pool, MI355_POP_MIN=$PM" >> $O/sweep.log
  MI355_LIB=$R/embree_amd/lib/variant_pool.so MI355_POP_MIN=$PM timeout 600 python tests/gpu_batch_sweep.py --lo 17 --hi 21 --md --tag pool$PM 2>&1 | grep -a "^|\|SWEEP\|rror\|fault\|differ" >> $O/sweep.log
done
cat $O/sweep.log
