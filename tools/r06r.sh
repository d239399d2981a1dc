#!/bin/bash
# round 6: small_build with a register budget for four waves per SIMD and / or 9.9 KB of LDS per wave (16 waves per CU); the sort with the cheaper match and the four-wide look-back
cd "$GRAFT_REPO_ROOT" || exit 1
R=$PWD; O=gpurun_out/r06r; mkdir -p $O; rm -rf $O/*
timeout 300 python -m pytest tests/test_gpu_round6.py -q -m gpu -k "radix_sort" 2>&1 | tail -1
timeout 120 python tests/gpu_sort_time.py 4762764 12 check 2>&1 | grep -a "SORT\|rror" >> $O/sort.log
KEYS=random timeout 120 python tests/gpu_sort_time.py 4762764 12 check 2>&1 | grep -a "SORT\|rror" >> $O/sort.log
cat $O/sort.log
for V in product sw4 sw4lds lds; do
  echo "== $V" >> $O/ab.log
  L=$R/embree_amd/lib/variant_$V.so; [ $V = product ] && L=$R/embree_amd/lib/libembree4_mi355.so
  MI355_LIB=$L TREEHASH=1 timeout 300 python tests/gpu_build_only.py "" 8 2>&1 | grep -E "TREEHASH|BUILD|rror|fault" >> $O/ab.log
  MI355_LIB=$L TREEHASH=1 PP=1 timeout 300 python tests/gpu_build_only.py "" 5 2>&1 | grep -E "TREEHASH|BUILD|rror|fault" >> $O/ab.log
done
cat $O/ab.log
