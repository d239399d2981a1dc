"""Throughput of back-to-back closest-hit launches on 1, 2 and 3 HIP streams (GPU box, not a pytest file).
The persistent kernel fills the chip, so a launch on a second stream starts as the blocks of the first one retire:
the tail of one batch (lanes without a ray) overlaps with the start of the next."""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from embree_amd import api, workloads as W                       # noqa: E402
from embree_amd.rtypes import RAYHIT_DTYPE                        # noqa: E402

L = api.load()
dev = api.Device("")
meshes = W.synthetic_crown(num_phi=int(os.environ.get("PHI", "158")))
s = api.Scene(dev)
for v, t in meshes:
    s.add_triangle_mesh(v, t, device_resident=True)
s.commit()
prim = W.crown_camera_rays(meshes, 1024, 1024)
d = api.DeviceArray.from_numpy(prim)
s.intersect1M_device(d.ptr, prim.shape[0])
L.mi355_device_synchronize(0)
rays = W.diffuse_bounce_rays(d.download(RAYHIT_DTYPE), meshes)
M = rays.shape[0]
K = 24
pristine = api.DeviceArray.from_numpy(rays)
bufs = [api.DeviceArray(rays.nbytes) for _ in range(K)]
ref = None
EVENTS = int(os.environ.get("EVENTS", "0"))
evs = [C.c_void_p() for _ in range(2 * K)]
for e in evs:
    L.mi355_event_create(C.byref(e))
for ns in (1, 2, 3, 4):
    streams = []
    for i in range(ns):
        st = C.c_void_p()
        L.mi355_stream_create(0, C.byref(st))
        streams.append(st)
    best = None
    for rep in range(3):
        for b in bufs:
            L.mi355_memcpy_d2d_async(b.ptr, pristine.ptr, rays.nbytes, None)
        L.mi355_device_synchronize(0)
        t0 = time.perf_counter()
        for k in range(K):
            if EVENTS:
                rc = L.mi355_trace_timed(s.bvh(), bufs[k].ptr, M, 96, 0, streams[k % ns], evs[2 * k], evs[2 * k + 1])
            else:
                rc = L.mi355_trace_closest(s.bvh(), bufs[k].ptr, M, 96, streams[k % ns])
            assert rc == 0
        L.mi355_device_synchronize(0)
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    out = bufs[-1].download(RAYHIT_DTYPE).tobytes()
    ref = ref or out
    print("OVERLAP streams=%d: %d launches of %d rays in %.3f ms -> %.1f Mrays/s (%.3f ms per batch) same_result=%s"
          % (ns, K, M, 1e3 * best, K * M / best / 1e6, 1e3 * best / K, out == ref), flush=True)
