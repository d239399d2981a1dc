"""GPU box tool (not a pytest file): the host-array entry point (rtcIntersect1M on a pageable array of 2^20 RTCRayHit) -- ms per call, best of 7, for the CPU copy
thread count of the environment (MI355_COPY_THREADS) and with the caller's array registered instead (host_register=1)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from embree_amd import api, workloads as W                       # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "gpu=0"
dev = api.Device(cfg)
meshes = W.synthetic_crown(num_phi=48)
s = api.make_scene(dev, meshes)
prim = W.crown_camera_rays(meshes, 1024, 1024)
s.intersect1M(prim)
rays = W.diffuse_bounce_rays(prim, meshes, seed=1)
want = rays.copy(); s.intersect1M(want)
t = []
for _ in range(7):
    r = rays.copy()
    t0 = time.perf_counter(); s.intersect1M(r); t.append((time.perf_counter() - t0) * 1e3)
assert r.tobytes() == want.tobytes()
print("E2E cfg=%s copy_threads=%s: best %.2f ms, median %.2f ms for %d rays = %.0f Mrays/s" % (cfg, os.environ.get("MI355_COPY_THREADS", "default"), min(t), sorted(t)[3], rays.shape[0], rays.shape[0] / min(t) / 1e3))
