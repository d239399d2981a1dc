"""A/B timing helper for the GPU box (not a pytest file).
  [MI355_TRACE_VARIANT=1|2] [MI355_TRACE_BLOCKS_PER_CU=n] python tests/gpu_perf.py [--config k=v,..] [--phi 158] [--any]
Prints kernel time (HIP events around the kernel), visit statistics and a checksum of the results."""
import argparse
import ctypes as C
import hashlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from embree_amd import api, workloads as W                       # noqa: E402
from embree_amd.rtypes import RAYHIT_DTYPE, RAY_DTYPE, rays_of   # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--config", default="")
ap.add_argument("--phi", type=int, default=158)
ap.add_argument("--reps", type=int, default=10)
ap.add_argument("--any", action="store_true")
ap.add_argument("--robust", action="store_true", help="RTC_SCENE_FLAG_ROBUST scene")
ap.add_argument("--low", action="store_true", help="RTC_BUILD_QUALITY_LOW (Morton build)")
ap.add_argument("--high", action="store_true", help="RTC_BUILD_QUALITY_HIGH (presplit build)")
ap.add_argument("--sort", default="", help="experiment: reorder the rays on the host before the upload: origin | origin+octant | octant")
ap.add_argument("--powerplant", action="store_true", help="configs[4]: the 12.7 M triangle powerplant stand-in instead of the crown stand-in")
ap.add_argument("--primary", action="store_true")
ap.add_argument("--shadow", action="store_true", help="configs[3]: one rank's shard (2^21 rays) of the 16 Mi shadow rays, any hit")
ap.add_argument("--tag", default="")
ap.add_argument("--tess-room", type=int, default=0, help="experiment: the 12 room triangles of the crown stand-in replaced by K x K quads per wall (what cutting them into sphere-sized pieces would give)")
ap.add_argument("--retrace", action="store_true", help="trace once, then time the same rays with tfar preset to the hit distance (perfect-culling bound)")
a = ap.parse_args()
L = api.load()
dev = api.Device(a.config)
meshes = W.synthetic_powerplant() if a.powerplant else W.synthetic_crown(num_phi=a.phi)
if a.tess_room and not a.powerplant:
    v, t = meshes[-1]
    K = a.tess_room
    nv, nt = [], []
    for q in range(6):                                           # every wall = two triangles (a, b, c), (a, c, d)
        pa, pb, pc = v[t[2 * q]]
        pd = v[t[2 * q + 1][2]]
        base = len(nv)
        for j in range(K + 1):
            for i in range(K + 1):
                fu, fv = i / K, j / K
                nv.append((1 - fu) * (1 - fv) * pa + fu * (1 - fv) * pb + fu * fv * pc + (1 - fu) * fv * pd)
        for j in range(K):
            for i in range(K):
                p0 = base + j * (K + 1) + i
                nt += [(p0, p0 + 1, p0 + K + 2), (p0, p0 + K + 2, p0 + K + 1)]
    meshes = meshes[:-1] + [(np.array(nv, np.float32), np.array(nt, np.uint32))]
s = api.Scene(dev, api.RTC_SCENE_FLAG_ROBUST if a.robust else 0, api.RTC_BUILD_QUALITY_LOW if a.low else (api.RTC_BUILD_QUALITY_HIGH if a.high else None))
for v, t in meshes:
    s.add_triangle_mesh(v, t, device_resident=True)
s.commit()
s.touch(); s.commit()
info = s.info()
prim = W.crown_camera_rays(meshes, 1024, 1024)
d = api.DeviceArray.from_numpy(prim)
s.intersect1M_device(d.ptr, prim.shape[0])
L.mi355_device_synchronize(0)
tr = d.download(RAYHIT_DTYPE)
rays = prim if a.primary else W.diffuse_bounce_rays(tr, meshes)
if a.retrace:
    d2 = api.DeviceArray.from_numpy(rays)
    s.intersect1M_device(d2.ptr, rays.shape[0])
    L.mi355_device_synchronize(0)
    t2 = d2.download(RAYHIT_DTYPE)
    rays = rays.copy()
    rays["tfar"] = np.where(t2["geomID"] != 0xFFFFFFFF, t2["tfar"] * np.float32(1.000001), rays["tfar"])
    d2.free()
if a.sort:
    lo, hi = W.scene_bounds(meshes)
    org = np.stack([rays["org_x"], rays["org_y"], rays["org_z"]], -1)
    q = np.clip(((org - lo) / np.maximum(hi - lo, 1e-20) * 32).astype(np.int64), 0, 31)
    def spread(v):
        out = np.zeros_like(v)
        for b in range(5):
            out |= ((v >> b) & 1) << (3 * b)
        return out
    cell = spread(q[:, 0]) | (spread(q[:, 1]) << 1) | (spread(q[:, 2]) << 2)
    octant = (rays["dir_x"] < 0).astype(np.int64) | ((rays["dir_y"] < 0).astype(np.int64) << 1) | ((rays["dir_z"] < 0).astype(np.int64) << 2)
    key = {"origin": cell, "origin+octant": (cell << 3) | octant, "octant": octant, "octant+origin": (octant << 15) | cell}[a.sort]
    rays = rays[np.argsort(key, kind="stable")].copy()
if a.shadow:                                                     # (as tests/gpu_configs.py: 16 shadow rays per hit point of the first 2^17 bounce rays)
    db = api.DeviceArray.from_numpy(rays)
    s.intersect1M_device(db.ptr, rays.shape[0])
    L.mi355_device_synchronize(0)
    bt = db.download(RAYHIT_DTYPE)
    db.free()
    rays = W.shadow_rays(bt[: 1 << 17], meshes, samples=16)
    a.any = True
elif a.any:
    rays = rays_of(rays)
M, rec = rays.shape[0], rays.dtype.itemsize
pristine = api.DeviceArray.from_numpy(rays)
work = api.DeviceArray(rays.nbytes)
e0, e1 = C.c_void_p(), C.c_void_p()
L.mi355_event_create(C.byref(e0))
L.mi355_event_create(C.byref(e1))
ms = []
for i in range(a.reps + 2):
    L.mi355_memcpy_d2d_async(work.ptr, pristine.ptr, rays.nbytes, None)
    rc = L.mi355_trace_timed(s.bvh(), work.ptr, M, rec, int(a.any), None, e0, e1)
    assert rc == 0, L.mi355_last_error()
    t = C.c_float()
    L.mi355_event_elapsed_ms(e0, e1, C.byref(t))
    if i >= 2:
        ms.append(t.value)
res = work.download(rays.dtype)
L.mi355_memcpy_d2d_async(work.ptr, pristine.ptr, rays.nbytes, None)
st = s.trace_stats(work.ptr, M, rec, a.any)
alg = M * 48 + (M * 4 if a.any else int((res["geomID"] != 0xFFFFFFFF).sum()) * 52) + st["nodes"] * 80 + st["tris"] * 48
best = min(ms)
print("PERF %-24s cfg='%s' build=%.2fms nodes=%d leaves=%d depth=%d sah=%.1f | kernel min %.3f avg %.3f ms -> %.1f Mrays/s | "
      "alg %.0f B/ray %.0f GB/s frac %.3f | per ray: nodes %.2f tris %.2f | wave_iters %d util node %.2f tri %.2f | spills %d stack %d | md5 %s"
      % (a.tag, a.config, info["build_ms"], info["num_nodes"], info["num_leaves"], info["depth"], info["sah"], best,
         float(np.mean(ms)), M / best / 1e3, alg / M, alg / best / 1e6, alg / best / 1e6 / 8000.0, st["nodes"] / M,
         st["tris"] / M, st["wave_iters"], st["nodes"] / max(1, 64 * st["node_blocks"]), st["tris"] / max(1, 64 * st["tri_blocks"]),
         st["spills"], st["max_depth"], hashlib.md5(res.tobytes()).hexdigest()[:10]), flush=True)
li = 64.0 * max(1, st["wave_iters"])
print("     lane-iterations: node %.3f idle %.3f wait_batch %.3f wait_drain %.3f blocked %.3f" % (st["nodes"] / li, st["lanes_idle"] / li,
      st["lanes_wait_batch"] / li, st["lanes_wait_drain"] / li, st["lanes_blocked"] / li) +
      " | per ray: nodes with no hit %.2f, groups culled at pop %.2f" % (st["empty_nodes"] / M, st["culled_groups"] / M), flush=True)
