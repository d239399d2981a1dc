#!/bin/bash
# round 6, session 5: host-array queries that send 48 bytes per ray up and only the written fields down (packed_link, default) against whole records both ways
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06zv; mkdir -p $O; rm -rf $O/*
{
python - <<'PY'
import numpy as np, sys, os
sys.path.insert(0, os.getcwd())
from embree_amd import api, workloads as W
from embree_amd.rtypes import rays_of
meshes = W.synthetic_crown(num_phi=48)
out = {}
for cfg in ("gpu=0,packed_link=0", "gpu=0"):
    dev = api.Device(cfg); s = api.make_scene(dev, meshes)
    prim = W.crown_camera_rays(meshes, 1024, 1024); s.intersect1M(prim)
    rays = W.diffuse_bounce_rays(prim, meshes, seed=1)
    got = rays.copy(); s.intersect1M(got)
    sh = rays_of(rays); s.occluded1M(sh)
    out[cfg] = (prim.tobytes(), got.tobytes(), sh.tobytes())
    s.release()
a, b = out["gpu=0,packed_link=0"], out["gpu=0"]
print("PACKED vs WHOLE: primary identical", a[0] == b[0], "bounce identical", a[1] == b[1], "occluded identical", a[2] == b[2])
PY
for C in "gpu=0,packed_link=0" "gpu=0" "gpu=0,host_register=1"; do timeout 300 python tests/gpu_e2e_time.py "$C" 2>&1 | grep -a "E2E\|rror"; done
} > $O/e2e.log 2>&1
cat $O/e2e.log
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round2.py tests/test_gpu_round3.py -m gpu -x -q 2>&1 | tail -4
