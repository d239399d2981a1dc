#!/bin/bash
# round 6: after "host memory never meets the GPU": the first four test files in order 14 times (was: 1 GPU memory fault in ~10), then the whole suite twice
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06j; mkdir -p $O; rm -f $O/*
fail=0
for i in $(seq 1 14); do
  python -m pytest tests/test_gpu_parity.py tests/test_gpu_reference_suite.py tests/test_gpu_round2.py tests/test_gpu_round3.py -x -q --capture=sys -m gpu > $O/seq.log 2>&1 || { fail=$((fail+1)); cp $O/seq.log $O/seqfail_$i.log; }
done
echo "staged host paths: $fail failures of 14"
for f in $O/seqfail*; do [ -f "$f" ] && { echo "== $f"; grep -a -v "^  File\|Extension modules" $f | tail -25; }; done
for i in 1 2; do ( time python -m pytest tests -q --capture=sys -m gpu ) > $O/full_$i.log 2>&1; grep -a -v "^  File\|Extension modules" $O/full_$i.log | tail -8; done
