"""Wave clocks per step of the traversal loop (a scratch build with s_memtime stamps between the steps: tools/patches/r05_segclk.patch):
    MI355_LIB=embree_amd/lib/variant_prof.so python tests/gpu_segclk.py [--shadow|--powerplant]"""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from embree_amd import api, workloads as W
from embree_amd.rtypes import RAYHIT_DTYPE
L = api.load()
dev = api.Device("")
m = W.synthetic_crown()
s = api.Scene(dev)
for v, t in m:
    s.add_triangle_mesh(v, t, device_resident=True)
s.commit()
prim = W.crown_camera_rays(m, 1024, 1024)
d = api.DeviceArray.from_numpy(prim)
s.intersect1M_device(d.ptr, prim.shape[0]); L.mi355_device_synchronize(0)
rays = W.diffuse_bounce_rays(d.download(RAYHIT_DTYPE), m, seed=1)
work = api.DeviceArray.from_numpy(rays)
out = (C.c_uint64 * 32)()
assert L.mi355_trace_stats(s.bvh(), work.ptr, rays.shape[0], 96, 0, out) == 0
names = ["0 prefetch issue", "1 hand-out", "1b tail", "2 pop", "3a node loads", "4 triangle block", "3b node tests", "5 queueing + loop end"]
tot = float(out[16])
print("SEGCLK loop clocks %.4g, iterations %d, per iteration %.0f" % (tot, out[5], tot * 64 / max(1, out[5]) / 64))
for i, n in enumerate(names):
    print("SEGCLK %-24s %.4f of the loop clocks" % (n, out[18 + i] / tot))
