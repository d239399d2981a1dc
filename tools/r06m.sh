#!/bin/bash
# round 6: lbvh_bounds with a workgroup-local phase (LOW build; hash must stay 97a5042e.../617a8afd...), and the small threshold again (sets of <= T triangles go to small_build)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06m; mkdir -p $O; rm -f $O/*
TREEHASH=1 timeout 300 python tests/gpu_build_only.py "" 6 0 2>&1 | grep -E "TREEHASH|BUILD|rror|fault" >> $O/ab.log
PP=1 TREEHASH=1 timeout 300 python tests/gpu_build_only.py "" 4 0 2>&1 | grep -E "TREEHASH|BUILD|rror|fault" >> $O/ab.log
for T in 1024 768 512 384 256; do
  echo "== small_threshold=$T" >> $O/ab.log
  TREEHASH=1 timeout 300 python tests/gpu_build_only.py "small_threshold=$T" 8 2>&1 | grep -E "TREEHASH|BUILD|rror|fault" >> $O/ab.log
  PP=1 TREEHASH=1 timeout 300 python tests/gpu_build_only.py "small_threshold=$T" 5 2>&1 | grep -E "TREEHASH|BUILD|rror|fault" >> $O/ab.log
done
cat $O/ab.log
python -m pytest tests/test_gpu_parity.py -q -m gpu -k "low_quality or morton or lbvh" 2>&1 | tail -3
