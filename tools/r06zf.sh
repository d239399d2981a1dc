#!/bin/bash
# round 6, session 5: where small_build's wave cycles go (SM_TIME / SM_STATS variant)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06zf; mkdir -p $O; rm -rf $O/*
MI355_LIB=$PWD/embree_amd/lib/variant_smtime.so timeout 300 python tests/gpu_build_only.py "" 3 2>&1 | grep -a "mi355 build\|BUILD\|rror\|fault" > $O/smtime.log
cat $O/smtime.log
