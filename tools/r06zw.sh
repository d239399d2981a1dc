#!/bin/bash
# round 6, session 5: packed host-array path: CPU copy threads and chunk sizes
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06zw; mkdir -p $O; rm -rf $O/*
{
for T in 2 4 6 10 14; do MI355_COPY_THREADS=$T timeout 300 python tests/gpu_e2e_time.py "gpu=0" 2>&1 | grep -a "E2E\|rror"; done
for CH in 65536 262144 524288; do timeout 300 python tests/gpu_e2e_time.py "gpu=0,host_pipeline_chunk=$CH" 2>&1 | grep -a "E2E\|rror"; done
nproc; cat /sys/fs/cgroup/cpu.max 2>/dev/null
} > $O/e2e.log 2>&1
cat $O/e2e.log
