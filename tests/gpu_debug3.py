"""debug: host packet calls vs the batch path"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from embree_amd import api, workloads as W
from embree_amd.rtypes import INVALID_ID
from oracle import restate
from tests.test_gpu_round2 import to_packets, from_packets, _aligned
dev = api.Device("gpu=0")
L = api.load()
meshes = W.synthetic_crown(num_phi=24)
s = api.make_scene(dev, meshes)
o = restate.OracleScene()
for v, t in meshes: o.add_mesh(v, t)
o.commit()
prim = W.crown_camera_rays(meshes, 64, 64)
o.intersect1(prim)
rays = W.diffuse_bounce_rays(prim, meshes, seed=5)
big = rays.copy(); s.intersect1M(big)
for K in (4, 8):
  for pattern in ("all", "fifth"):
    valid = np.ones(rays.shape[0], np.int32)
    if pattern == "fifth": valid[::5] = 0
    act = valid != 0
    nbad = 0
    for p in range(64):
        sel = slice(p * K, (p + 1) * K)
        buf = _aligned(to_packets(rays[sel], K, 21)[0])
        v = _aligned(np.where(act[sel], -1, 0).astype(np.int32))
        getattr(L, "rtcIntersect%d" % K)(v.ctypes.data, s.h, buf.ctypes.data, None)
        one = from_packets(buf[None], rays[sel])
        exp = np.where(act[sel], big[sel]["primID"], rays[sel]["primID"])
        d = one["primID"] != exp
        if d.any():
            nbad += 1
            if nbad <= 3:
                print(K, pattern, "packet", p, "act", act[sel].astype(int).tolist(), "got prim", one["primID"].tolist(), "exp", exp.tolist(),
                      "tfar", one["tfar"].tolist(), "exp tfar", big[sel]["tfar"].tolist())
    print(K, pattern, "bad packets", nbad, "err", dev.get_error())
