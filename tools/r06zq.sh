#!/bin/bash
# round 6, session 5: spatial_bin with 1 / 2 / 4 workgroups per chunk: HIGH commits of the crown and the powerplant stand-ins, tree hashes
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06zq; mkdir -p $O; rm -rf $O/*
for V in sp1 product sp4; do
  echo "== $V" >> $O/ab.log
  if [ $V = product ]; then L=$PWD/embree_amd/lib/libembree4_mi355.so; else L=$PWD/embree_amd/lib/variant_$V.so; fi
  MI355_LIB=$L TREEHASH=1 timeout 300 python tests/gpu_build_only.py "" 6 2 2>&1 | grep -a "TREEHASH\|BUILD\|rror\|fault" >> $O/ab.log
  MI355_LIB=$L PP=1 TREEHASH=1 timeout 300 python tests/gpu_build_only.py "" 3 2 2>&1 | grep -a "TREEHASH\|BUILD\|rror\|fault" >> $O/ab.log
done
cat $O/ab.log
