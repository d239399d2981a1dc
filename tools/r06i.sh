#!/bin/bash
# round 6, VERDICT r05 item 6: which of the kernel's hand-written pieces breaks the FILT == 2 kernels above -O1?  Every variant: the EMPTY callee (WHICH=1: the hits must be
# the plain hits) and the real rule (WHICH=0), fast and robust scene.
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06i; mkdir -p $O; rm -f $O/*
for v in o1 o3 o3_nosgpr o3_noscan o3_noundef o3_nopre o3_all o2_all; do
  for rob in "" 1; do for w in 1 0; do
    echo "== $v robust=${rob:-0} which=$w" >> $O/log.txt
    MI355_LIB=embree_amd/lib/variant_fp_$v.so ROBUST=$rob SMALL_ONLY=1 WHICH=$w timeout 120 python tests/gpu_devfilter.py 2>&1 | grep -a "calls /\|rror\|fault\|Abort" >> $O/log.txt
  done; done
done
cat $O/log.txt
