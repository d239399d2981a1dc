#!/bin/bash
# round 6, first GPU call: the boundary tests + the batch-size sweep of the launch shapes
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06a; mkdir -p $O
python -m pytest tests/test_gpu_round6.py tests/test_gpu_round5.py -x -q -m gpu -s > $O/tests.log 2>&1; echo "tests rc $?" >> $O/tests.log
for R in 0 1 16 32 64; do
  MI355_STATIC_RAYS=$R timeout 600 python tests/gpu_batch_sweep.py --lo 12 --hi 21 --tag R$R > $O/sweep_R$R.log 2>&1
done
tail -5 $O/tests.log; grep -h SWEEP $O/sweep_*.log
