"""Batch-size sweep of the closest-hit kernel on the bench's workload (GPU box tool, not a pytest file).
  [MI355_STATIC_RAYS=0|16|32|64] python tests/gpu_batch_sweep.py [--lo 14] [--hi 22] [--reps 30] [--tag x]
For every size 2^k: lone launches over the FIRST 2^k rays of the 2^20 diffuse-bounce rays of configs[2] (sizes above 2^20: the batch again with other seeds), HIP events
around each launch, and a byte comparison of the hit records with the same rays' records out of the full-size launch (the launch shape must not show in the answers)."""
import argparse
import ctypes as C
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from embree_amd import api, workloads as W                       # noqa: E402
from embree_amd.rtypes import RAYHIT_DTYPE                       # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--lo", type=int, default=14)
ap.add_argument("--hi", type=int, default=22)
ap.add_argument("--reps", type=int, default=30)
ap.add_argument("--phi", type=int, default=158)
ap.add_argument("--tag", default=os.environ.get("MI355_STATIC_RAYS", "auto"))
ap.add_argument("--md", action="store_true", help="print a markdown table row per size")
ap.add_argument("--census", action="store_true", help="lane-iteration census of the counting kernel per size (cursor hand-out whatever the setting)")
a = ap.parse_args()
L = api.load()
dev = api.Device("gpu=0")
meshes = W.synthetic_crown(num_phi=a.phi)
s = api.Scene(dev)
for v, t in meshes:
    s.add_triangle_mesh(v, t, device_resident=True)
s.commit()
prim = W.crown_camera_rays(meshes, 1024, 1024)
d = api.DeviceArray.from_numpy(prim)
s.intersect1M_device(d.ptr, prim.shape[0])
L.mi355_device_synchronize(0)
tr = d.download(RAYHIT_DTYPE)
d.free()
parts = [W.diffuse_bounce_rays(tr, meshes, seed=1)]
while len(parts) << 20 < 1 << a.hi:
    parts.append(W.diffuse_bounce_rays(tr, meshes, seed=1 + len(parts)))
rays = np.concatenate(parts)
pristine = api.DeviceArray.from_numpy(rays)
work = api.DeviceArray(rays.nbytes)
e0, e1 = C.c_void_p(), C.c_void_p()
L.mi355_event_create(C.byref(e0))
L.mi355_event_create(C.byref(e1))
rec = rays.dtype.itemsize


def run(n, reps):
    ms = []
    for i in range(reps + 3):
        L.mi355_memcpy_d2d_async(work.ptr, pristine.ptr, n * rec, None)
        rc = L.mi355_trace_timed(s.bvh(), work.ptr, n, rec, 0, None, e0, e1)
        assert rc == 0, L.mi355_last_error()
        t = C.c_float()
        L.mi355_event_elapsed_ms(e0, e1, C.byref(t))
        if i >= 3:
            ms.append(t.value)
    return np.array(ms), work.download(RAYHIT_DTYPE, n)


# the reference bytes: every 2^20 part in ONE launch of its own (more rays than lane slots: the cursor hand-out whatever the setting)
full = []
for k in range(len(parts)):
    L.mi355_memcpy_d2d_async(work.ptr, pristine.ptr + (k << 20) * rec, (1 << 20) * rec, None)
    assert L.mi355_trace_timed(s.bvh(), work.ptr, 1 << 20, rec, 0, None, e0, e1) == 0
    full.append(work.download(RAYHIT_DTYPE, 1 << 20))
full = np.concatenate(full)
print("SWEEP tag=%s md5(2^20)=%s" % (a.tag, hashlib.md5(full[: 1 << 20].tobytes()).hexdigest()[:8]), flush=True)
for k in range(a.lo, a.hi + 1):
    n = 1 << k
    ms, res = run(n, a.reps if k <= 20 else max(5, a.reps // 3))
    same = res.tobytes() == full[:n].tobytes()
    us_min, us_med = 1e3 * ms.min(), 1e3 * float(np.median(ms))
    if a.census:
        L.mi355_memcpy_d2d_async(work.ptr, pristine.ptr, n * rec, None)
        st = s.trace_stats(work.ptr, n, rec, False)
        li = 64.0 * max(1, st["wave_iters"])
        print("CENSUS rays=2^%-2d wave_iters %d (%.1f per 64 rays) lanes: node %.3f idle %.3f wait_batch %.3f wait_drain %.3f blocked %.3f | per ray nodes %.2f tris %.2f | tri batches filled %.2f"
              % (k, st["wave_iters"], st["wave_iters"] * 64.0 / n, st["nodes"] / li, st["lanes_idle"] / li, st["lanes_wait_batch"] / li, st["lanes_wait_drain"] / li,
                 st["lanes_blocked"] / li, st["nodes"] / n, st["tris"] / n, st["tris"] / max(1, 64 * st["tri_blocks"])), flush=True)
    if a.md:
        print("| %s | 2^%d | %.1f | %.1f | %.0f | %s |" % (a.tag, k, us_min, us_med, n / us_med, "yes" if same else "NO"), flush=True)
    else:
        print("SWEEP tag=%-5s rays=2^%-2d us min %.1f median %.1f -> %.0f Mrays/s (median) identical=%s" % (a.tag, k, us_min, us_med, n / us_med, same), flush=True)
