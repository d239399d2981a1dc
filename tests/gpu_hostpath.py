"""Host-pointer entry points at the bench's size (GPU box, not a pytest file):  python tests/gpu_hostpath.py
rtcIntersect1M / rtcOccluded1M on a pageable numpy array of 2^20 rays: plain path (one upload, one launch, one download) against the pipelined path
(array pinned for the call, chunks alternating between two streams), several chunk sizes, and the in-place path (host_in_place=1: the array registered with
the device and traced where it lies: 48 bytes read and <= 52 written per ray over the host link instead of 96 each way).  Results must be identical."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from embree_amd import api, workloads as W                       # noqa: E402
from embree_amd.rtypes import rays_of                             # noqa: E402

meshes = W.synthetic_crown()
rays = None
ref = None
CONFIGS = ("host_pipeline_min=4000000000", "host_in_place=1", "host_pipeline_chunk=65536", "host_pipeline_chunk=131072", "host_pipeline_chunk=262144", "host_pipeline_chunk=524288")
for cfg in (sys.argv[1:] or CONFIGS):                          # (configs on the command line: only those)
    dev = api.Device(cfg)
    s = api.Scene(dev)
    for v, t in meshes:
        s.add_triangle_mesh(v, t, device_resident=True)
    s.commit()
    if rays is None:
        prim = W.crown_camera_rays(meshes, 1024, 1024)
        tr = prim.copy(); s.intersect1M(tr)
        rays = W.diffuse_bounce_rays(tr, meshes)
    best, besto = 1e9, 1e9
    for _ in range(6):
        a = rays.copy()
        t0 = time.perf_counter(); s.intersect1M(a); best = min(best, time.perf_counter() - t0)
        r = rays_of(rays)
        t0 = time.perf_counter(); s.occluded1M(r); besto = min(besto, time.perf_counter() - t0)
    if ref is None:
        ref = (a.tobytes(), r.tobytes())
    same = a.tobytes() == ref[0] and r.tobytes() == ref[1]
    print("HOSTPATH %-32s rtcIntersect1M %.2f ms = %.0f Mrays/s | rtcOccluded1M %.2f ms = %.0f Mrays/s | identical: %s"
          % (cfg, best * 1e3, rays.shape[0] / best / 1e6, besto * 1e3, rays.shape[0] / besto / 1e6, same))
    s.release(); dev.release()
