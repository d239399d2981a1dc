"""debug helper (GPU box): whole-batch device launch vs chunked host-array path on the crown stand-in"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from embree_amd import api, workloads as W
from embree_amd.rtypes import RAYHIT_DTYPE
cfg = sys.argv[1] if len(sys.argv) > 1 else ""
L = api.load()
dev = api.Device(cfg)
meshes = W.synthetic_crown(num_phi=int(os.environ.get("PHI", "158")))
s = api.Scene(dev)
for v, t in meshes:
    s.add_triangle_mesh(v, t, device_resident=True)
s.commit()
prim = W.crown_camera_rays(meshes, 1024, 1024)
d = api.DeviceArray.from_numpy(prim)
s.intersect1M_device(d.ptr, prim.shape[0]); L.mi355_device_synchronize(0)
tr = d.download(RAYHIT_DTYPE)
rays = W.diffuse_bounce_rays(tr, meshes, seed=1)
d2 = api.DeviceArray.from_numpy(rays)
s.intersect1M_device(d2.ptr, rays.shape[0]); L.mi355_device_synchronize(0)
a = d2.download(RAYHIT_DTYPE)
d3 = api.DeviceArray.from_numpy(rays)
st = s.trace_stats(d3.ptr, rays.shape[0], 96)
c = d3.download(RAYHIT_DTYPE)
b = rays.copy(); s.intersect1M(b)
for name, x in (("stats", c), ("host", b)):
    diff = np.nonzero(x.view(np.uint8).reshape(-1, 96) != a.view(np.uint8).reshape(-1, 96))[0]
    idx = np.unique(diff)
    print(cfg, name, "rays differing from the whole-batch device launch:", idx.shape[0], idx[:8])
    for i in idx[:4]:
        print("  ray", i, "dev", a[i]["tfar"], a[i]["primID"], a[i]["geomID"], a[i]["u"], a[i]["v"], "|", name, x[i]["tfar"], x[i]["primID"], x[i]["geomID"], x[i]["u"], x[i]["v"])
print("info", s.info()["num_presplit"], s.info()["num_triangles"])
