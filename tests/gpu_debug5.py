"""debug: HIGH vs MEDIUM robust mismatch on the pipe-run scene"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from embree_amd import api, workloads as W
from embree_amd.rtypes import INVALID_ID
from tests.test_gpu_round2 import _spatial_scenes, _aimed_rays
from tests import bvh_check
dev = api.Device("gpu=0")
meshes = _spatial_scenes()["powerplant_200k"]
rays = _aimed_rays(meshes)
flags = 4
med = api.make_scene(dev, meshes, flags=flags); high = api.make_scene(dev, meshes, flags=flags, quality=2)
a, b = rays.copy(), rays.copy()
high.intersect1M(a); med.intersect1M(b)
d = np.nonzero((a["tfar"] != b["tfar"]))[0]
print("differ in t", d)
nodes, tris = high.download_bvh()
for i in d[:2]:
    r = rays[i]
    o = np.array([r["org_x"], r["org_y"], r["org_z"]], np.float64); dr = np.array([r["dir_x"], r["dir_y"], r["dir_z"]], np.float64)
    print("ray", i, o, dr, "high", a["primID"][i], repr(a["tfar"][i]), "med", b["primID"][i], repr(b["tfar"][i]))
    for p in (int(a["primID"][i]), int(b["primID"][i])):
        v, t = meshes[0]; tv = v[t[p]].astype(np.float64)
        n = np.cross(tv[1] - tv[0], tv[2] - tv[0]); tt = np.dot(n, tv[0] - o) / np.dot(n, dr)
        print("  prim", p, "fp64 t", tt, "point", o + tt * dr)
    want = int(b["primID"][i])
    # walk: all root-to-leaf paths that end in a record of `want`
    def visit(idx, path):
        nd = nodes[idx]; lo, hi = bvh_check.decode_child_boxes(nd); imask = int(nd["imask"]); rank = 0
        for s in range(8):
            m = int(nd["meta"][s])
            if m == 0: continue
            with np.errstate(divide="ignore", invalid="ignore"):
                t0 = (lo[s] - o) / dr; t1 = (hi[s] - o) / dr
            tn = np.minimum(t0, t1).max(); tf = np.maximum(t0, t1).min()
            ent = path + [(idx, s, tn, tf)]
            if (imask >> s) & 1:
                visit(int(nd["childBase"]) + rank, ent); rank += 1
            else:
                bits, ofs = m >> 5, m & 31; cnt = {1: 1, 3: 2, 7: 3}[bits]; first = int(nd["triBase"]) + ofs
                if want in tris["primID"][first:first + cnt].tolist():
                    print("  leaf with prim", want, "box", lo[s], hi[s])
                    for e in ent: print("      node %d slot %d  entry %.9f exit %.9f %s" % (e[0], e[1], e[2], e[3], "MISS" if e[2] > e[3] else ""))
    sys.setrecursionlimit(10000)
    visit(0, [])
