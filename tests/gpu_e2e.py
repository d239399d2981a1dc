"""rtcIntersect1M on a pageable host array of 2^20 rays (crown stand-in) for a few device configs (GPU box script)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
from embree_amd import api, workloads as W
from embree_amd.rtypes import RAYHIT_DTYPE
meshes = W.synthetic_crown()
base = None
for cfg in sys.argv[1:] or [""]:
    dev = api.Device(cfg); L = api.load()
    s = api.Scene(dev)
    for v, t in meshes: s.add_triangle_mesh(v, t, device_resident=True)
    s.commit()
    if base is None:
        prim = W.crown_camera_rays(meshes, 1024, 1024)
        d = api.DeviceArray.from_numpy(prim); s.intersect1M_device(d.ptr, prim.shape[0]); L.mi355_device_synchronize(0)
        base = W.diffuse_bounce_rays(d.download(RAYHIT_DTYPE), meshes)
    best = 1e9
    for r in range(6):
        a = base.copy()
        t0 = time.perf_counter(); s.intersect1M(a); best = min(best, time.perf_counter() - t0)
    print("E2E cfg=%-50r %.3f ms -> %.1f Mrays/s  (hits %d)" % (cfg, best * 1e3, base.shape[0] / best / 1e6, int((a["geomID"] != 0xFFFFFFFF).sum())), flush=True)
    s.release(); dev.release()
