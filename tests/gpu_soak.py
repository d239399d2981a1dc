"""Leak / stability soak (GPU box script): repeated commits of every build quality, host-array and device queries, scene / device churn; prints the drift of
free device memory (hipMemGetInfo) -- it must settle after the first rounds."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from embree_amd import api, workloads as W
from embree_amd.rtypes import rays_of
hip = C.CDLL("libamdhip64.so")
def free_mb():
    f, t = C.c_size_t(), C.c_size_t(); hip.hipMemGetInfo(C.byref(f), C.byref(t)); return f.value / 2**20
meshes = W.synthetic_crown(num_phi=50)
rays = W.incoherent_rays(200000, [0, 1, 0], seed=1)
marks = []
for rnd in range(12):
    dev = api.Device("gpu=0")
    for q in (None, api.RTC_BUILD_QUALITY_LOW, api.RTC_BUILD_QUALITY_HIGH):
        s = api.make_scene(dev, meshes, quality=q, flags=4 if rnd % 2 else 0)
        for k in range(3):
            s.touch(); s.commit()
        a = rays.copy(); s.intersect1M(a)
        r = rays_of(rays); s.occluded1M(r)
        d = api.DeviceArray.from_numpy(rays); s.intersect1M_device(d.ptr, rays.shape[0]); api.load().mi355_device_synchronize(0); d.free()
        s.release()
    dev.release()
    marks.append(free_mb())
    print("round %2d: free %.0f MB (hits %d)" % (rnd, marks[-1], int((a["geomID"] != 0xFFFFFFFF).sum())), flush=True)
drift = marks[3] - marks[-1]
print("SOAK drift after round 3: %.1f MB" % drift)
assert abs(drift) < 64.0, "device memory keeps shrinking"
