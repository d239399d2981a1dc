#!/bin/bash
# round 6, session 5: small_build's large mode with 1 / 2 / 4 / 8 rounds of loads in flight (and 4 at a register budget of four waves per SIMD)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06zg; mkdir -p $O; rm -rf $O/*
for V in product d1 d2 d4w4 d8; do
  echo "== $V" >> $O/ab.log
  if [ $V = product ]; then L=""; else L=$PWD/embree_amd/lib/variant_$V.so; fi
  MI355_LIB=${L:-$PWD/embree_amd/lib/libembree4_mi355.so} TREEHASH=1 timeout 300 python tests/gpu_build_only.py "" 8 2>&1 | grep -a "TREEHASH\|BUILD\|rror\|fault" >> $O/ab.log
done
MI355_LIB=$PWD/embree_amd/lib/libembree4_mi355.so PP=1 TREEHASH=1 timeout 300 python tests/gpu_build_only.py "" 4 2>&1 | grep -a "TREEHASH\|BUILD\|rror\|fault" >> $O/ab.log
TREEHASH=1 timeout 300 python tests/gpu_build_only.py "" 4 2 2>&1 | grep -a "TREEHASH\|BUILD\|rror\|fault" >> $O/ab.log
cat $O/ab.log
