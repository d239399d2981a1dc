"""Lone-launch and 4-in-flight rate of the bench workload (crown stand-in, 2^20 diffuse rays) for the kernel's environment knobs (GPU box, one process per setting):
    MI355_REFILL_MIN=12 python tests/gpu_knobs.py [tag]"""
import ctypes as C, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
from embree_amd import api, workloads as W
from embree_amd.rtypes import RAYHIT_DTYPE
L = api.load()
dev = api.Device(os.environ.get("CFG", ""))
m = W.synthetic_crown()
s = api.Scene(dev)
for v, t in m:
    s.add_triangle_mesh(v, t, device_resident=True)
s.commit()
prim = W.crown_camera_rays(m, 1024, 1024)
d = api.DeviceArray.from_numpy(prim)
s.intersect1M_device(d.ptr, prim.shape[0]); L.mi355_device_synchronize(0)
rays = W.diffuse_bounce_rays(d.download(RAYHIT_DTYPE), m, seed=1)
streams = []
for _ in range(4):
    st = C.c_void_p(); L.mi355_stream_create(0, C.byref(st)); streams.append(st)
M, reps = rays.shape[0], 24
pristine = api.DeviceArray.from_numpy(rays)
bufs = [api.DeviceArray(rays.nbytes) for _ in range(reps)]
out = []
for ns in (1, 4):
    best = None
    for _ in range(4):
        for b in bufs:
            L.mi355_memcpy_d2d_async(b.ptr, pristine.ptr, rays.nbytes, None)
        L.mi355_device_synchronize(0)
        t0 = time.perf_counter()
        for k in range(reps):
            assert L.mi355_trace_closest(s.bvh(), bufs[k].ptr, M, 96, streams[k % ns]) == 0
        L.mi355_device_synchronize(0)
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    out.append(reps * M / best / 1e6)
import hashlib
print("KNOBS %-28s lone %7.1f  4-in-flight %7.1f Mrays/s  md5 %s  env %s" % (sys.argv[1] if len(sys.argv) > 1 else "", out[0], out[1], hashlib.md5(bufs[0].download(RAYHIT_DTYPE).tobytes()).hexdigest()[:8],
      {k: v for k, v in os.environ.items() if k.startswith("MI355_")}), flush=True)
