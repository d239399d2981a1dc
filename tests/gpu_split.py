"""Does ONE batch finish sooner when it is issued as k sub-batches on k streams (their tails overlap)?  GPU box script."""
import ctypes as C, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
from embree_amd import api, workloads as W
from embree_amd.rtypes import RAYHIT_DTYPE
L = api.load()
dev = api.Device("")
meshes = W.synthetic_crown()
s = api.Scene(dev)
for v, t in meshes: s.add_triangle_mesh(v, t, device_resident=True)
s.commit()
prim = W.crown_camera_rays(meshes, 1024, 1024)
d = api.DeviceArray.from_numpy(prim); s.intersect1M_device(d.ptr, prim.shape[0]); L.mi355_device_synchronize(0)
rays = W.diffuse_bounce_rays(d.download(RAYHIT_DTYPE), meshes)
M = rays.shape[0]
pristine = api.DeviceArray.from_numpy(rays); work = api.DeviceArray.from_numpy(rays)
streams = []
for _ in range(8):
    st = C.c_void_p(); assert L.mi355_stream_create(0, C.byref(st)) == 0; streams.append(st)
    L.mi355_trace_prepare(s.bvh(), st)
def run(parts, reps=12):
    """parts: list of (first, count) issued on streams 0..len-1 at once; returns best wall ms of the whole batch (host clock around enqueue + sync of all)"""
    best = 1e9
    for r in range(reps):
        L.mi355_memcpy_d2d_async(work.ptr, pristine.ptr, rays.nbytes, None); L.mi355_device_synchronize(0)
        t0 = time.perf_counter()
        for k, (f, c) in enumerate(parts):
            assert L.mi355_trace_closest(s.bvh(), C.c_void_p(work.ptr + f * 96), c, 96, streams[k]) == 0
        for k in range(len(parts)): L.mi355_synchronize(streams[k])
        best = min(best, (time.perf_counter() - t0) * 1e3)
    return best
def even(k): return [(i * (M // k), M // k) for i in range(k)]
def geo(fr):
    out, f = [], 0
    for x in fr: c = int(M * x) // 64 * 64; out.append((f, c)); f += c
    out[-1] = (out[-1][0], M - out[-1][0]); return out
for name, parts in (("1 x M", even(1)), ("2 x M/2", even(2)), ("4 x M/4", even(4)), ("8 x M/8", even(8)),
                    ("0.4,0.3,0.2,0.1", geo([0.4, 0.3, 0.2, 0.1])), ("0.5,0.25,0.125,0.125", geo([0.5, 0.25, 0.125, 0.125]))):
    ms = run(parts)
    print("SPLIT %-22s %.3f ms -> %.1f Mrays/s (host clock, enqueue to last sync)" % (name, ms, M / ms / 1e3), flush=True)
