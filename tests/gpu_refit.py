"""Refit timing on the crown stand-in (GPU box, not a pytest file):  python tests/gpu_refit.py
Build with the refit data kept, then rtcUpdateGeometryBuffer + rtcCommitScene without moving anything: the refitted tree must be the built tree
bit for bit (same boxes in, same quantiser), and the commit time is the refit alone."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from embree_amd import api, workloads as W                       # noqa: E402

L = api.load()
dev = api.Device("")
for name, meshes in (("crown stand-in", W.synthetic_crown()), ("powerplant stand-in", W.synthetic_powerplant())):
    s = api.Scene(dev)
    gids = [s.add_triangle_mesh(v, t, device_resident=True) for v, t in meshes]
    for g in gids:
        s.set_geometry_build_quality(g, api.RTC_BUILD_QUALITY_REFIT)
    s.commit()
    i0 = s.info()
    n0, t0 = s.download_bvh()
    best = None
    for _ in range(4):
        for g in gids:
            h = L.rtcGetGeometry(s.h, g)
            L.rtcUpdateGeometryBuffer(h, api.RTC_BUFFER_TYPE_VERTEX, 0)
            L.rtcCommitGeometry(h)
        w0 = time.perf_counter()
        s.commit()
        wall = (time.perf_counter() - w0) * 1e3
        i1 = s.info()
        best = (i1["build_ms"], wall) if best is None or i1["build_ms"] < best[0] else best
    n1, t1 = s.download_bvh()
    same = n0.tobytes() == n1.tobytes() and t0.tobytes() == t1.tobytes()
    print("REFIT %-20s tris %d nodes %d depth %d | build %.2f ms | refit %.3f ms GPU, %.3f ms wall (%d refits) -> %.0f Mprims/s | tree identical to the build: %s"
          % (name, i0["num_triangles"], i0["num_nodes"], i0["depth"], i0["build_ms"], best[0], best[1], i1["num_refits"], i0["num_triangles"] / best[0] / 1e3, same))
    s.release()
