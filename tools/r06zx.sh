#!/bin/bash
# round 6, session 5: packed host-array path: copy threads x chunk sizes
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06zx; mkdir -p $O; rm -rf $O/*
{
for CH in 131072 262144 524288; do for T in 0 1 2 3; do MI355_COPY_THREADS=$T timeout 300 python tests/gpu_e2e_time.py "gpu=0,host_pipeline_chunk=$CH" 2>&1 | grep -a "E2E\|rror"; done; done
MI355_COPY_THREADS=2 timeout 300 python tests/gpu_e2e_time.py "gpu=0,packed_link=0" 2>&1 | grep -a "E2E\|rror"
MI355_COPY_THREADS=2 timeout 300 python tests/gpu_e2e_time.py "gpu=0,packed_link=0,host_pipeline_chunk=524288" 2>&1 | grep -a "E2E\|rror"
} > $O/e2e.log 2>&1
cat $O/e2e.log
