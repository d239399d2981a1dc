#!/bin/bash
# A/B of build variants on the GPU box: tools/ab_build.sh <outdir> <variant names...>   (libraries from tools/build_variant.sh)
O=$1; shift; mkdir -p $O
for v in "$@"; do
  echo "== $v" >> $O/ab.log
  MI355_LIB=embree_amd/lib/variant_$v.so TREEHASH=1 timeout 120 python tests/gpu_build_only.py "" 8 2>&1 | grep -E "TREEHASH|BUILD|Error|error|fault" >> $O/ab.log
done
cat $O/ab.log
