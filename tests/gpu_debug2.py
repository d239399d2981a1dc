"""debug: small launches vs one big launch on the same rays (helpers on / off)"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from embree_amd import api, workloads as W
from embree_amd.rtypes import INVALID_ID
from oracle import restate
dev = api.Device("gpu=0")
meshes = W.synthetic_crown(num_phi=24)
s = api.make_scene(dev, meshes)
o = restate.OracleScene()
for v, t in meshes: o.add_mesh(v, t)
o.commit()
prim = W.crown_camera_rays(meshes, 64, 64)
o.intersect1(prim)
rays = W.diffuse_bounce_rays(prim, meshes, seed=5)
want = rays.copy(); o.intersect1(want)
big = rays.copy(); s.intersect1M(big)
print("big vs oracle: id mismatches", int(((big["primID"] != want["primID"]) | (big["geomID"] != want["geomID"])).sum()))
for helpers in ("1", "0"):
    os.environ["MI355_TRACE_HELPERS"] = helpers
    for G in (1, 3, 4, 16, 64, 200):
        bad = 0; lost = 0; first = None
        for b in range(0, 1024, G):
            r = rays[b:b + G].copy()
            s.intersect1M(r)
            d = (r["primID"] != want["primID"][b:b + G]) | (r["geomID"] != want["geomID"][b:b + G])
            bad += int(d.sum()); lost += int(((r["geomID"] == INVALID_ID) & (want["geomID"][b:b + G] != INVALID_ID)).sum())
            if d.any() and first is None: first = (b, np.nonzero(d)[0][:4].tolist())
        print("helpers", helpers, "group", G, "mismatches", bad, "lost hits", lost, "first", first)
