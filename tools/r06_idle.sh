#!/bin/bash
# round 6, last session: what the idle lanes' root loads of step 3a cost the address path (tools/idle_lane_bench.hip)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06_idle; mkdir -p $O; rm -rf $O/*
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -w tools/idle_lane_bench.hip -o /tmp/idle_lane_bench || exit 1
for N in 32768 4000000 16000000; do timeout 120 /tmp/idle_lane_bench $N >> $O/idle.log 2>&1; done
cat $O/idle.log
