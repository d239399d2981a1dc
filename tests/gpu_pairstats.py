"""How many triangles of the leaf slots could share a record with a neighbour (two triangles with two common vertices = four vertices = 48 bytes)?
GPU-box script (not a pytest file):  python tests/gpu_pairstats.py [--powerplant] [--config k=v,..]
Downloads the committed tree and looks at every leaf slot: 1 / 2 / 3 triangles, and whether two of them share two vertex POSITIONS (bitwise).
Prints the fraction over all leaf slots and weighted by the slot's box area (the chance that a ray enters it)."""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from embree_amd import api, workloads as W   # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--config", default="")
ap.add_argument("--phi", type=int, default=158)
ap.add_argument("--powerplant", action="store_true")
a = ap.parse_args()
dev = api.Device(a.config)
meshes = W.synthetic_powerplant() if a.powerplant else W.synthetic_crown(num_phi=a.phi)
s = api.Scene(dev)
for v, t in meshes:
    s.add_triangle_mesh(v, t, device_resident=True)
s.commit()
nodes, tris = s.download_bvh()
# canonical vertex ids by position (a seam may repeat a position under two indices)
allv = np.concatenate([np.asarray(v, np.float32) for v, _ in meshes])
vbase = np.cumsum([0] + [len(v) for v, _ in meshes])
_, canon = np.unique(allv.view(np.uint32).reshape(-1, 3), axis=0, return_inverse=True)
canon = canon.reshape(-1)
tbase = np.cumsum([0] + [len(t) for _, t in meshes])
allt = np.concatenate([np.asarray(t, np.int64) + vbase[g] for g, (_, t) in enumerate(meshes)])
tri_verts = canon[allt]                                           # [global triangle][3] canonical vertex ids
gtri = tbase[tris["geomID"].astype(np.int64)] + (tris["primID"].astype(np.int64) & 0x7FFFFFFF)   # record -> global triangle
rv = tri_verts[gtri]                                              # [record][3]

meta = nodes["meta"].astype(np.uint32)                            # [node][8]
inner = ((nodes["imask"][:, None].astype(np.uint32) >> np.arange(8)) & 1).astype(bool)
leaf = (meta != 0) & ~inner
cnt = np.zeros_like(meta)
for b in (5, 6, 7):
    cnt += (meta >> b) & 1
first = nodes["triBase"][:, None].astype(np.int64) + (meta & 31)
scale = (nodes["exp"].astype(np.uint32) << 23).view(np.float32).astype(np.float64)                # [node][3]
ext = (nodes["qhi"].astype(np.float64) - nodes["qlo"].astype(np.float64)) * scale[:, :, None]     # [node][3][8]
area = ext[:, 0] * (ext[:, 1] + ext[:, 2]) + ext[:, 1] * ext[:, 2]                                # [node][8]


def shared(i, j):
    x, y = rv[i], rv[j]
    return (x[:, :, None] == y[:, None, :]).any(2).sum(1)


tot_w = tot = 0.0
rep = {}
for c in (1, 2, 3):
    m = leaf & (cnt == c)
    f = first[m]
    w = area[m]
    if c == 1:
        pair = np.zeros(len(f), bool)
    elif c == 2:
        pair = shared(f, f + 1) >= 2
    else:
        pair = (shared(f, f + 1) >= 2) | (shared(f, f + 2) >= 2) | (shared(f + 1, f + 2) >= 2)
    rep[c] = (len(f), float(pair.mean()) if len(f) else 0.0, float((w * pair).sum() / max(w.sum(), 1e-300)))
    tot += len(f)
    tot_w += w.sum()
ntri = sum(c * rep[c][0] for c in rep)
nrec = sum((c - rep[c][1]) * rep[c][0] for c in rep)                      # a pair saves one record
# area-weighted: triangles tested per entered slot vs records tested
wt = wr = 0.0
for c in (1, 2, 3):
    m = leaf & (cnt == c)
    w = area[m].sum()
    wt += c * w
    wr += (c - rep[c][2]) * w
print("PAIRS %s: %d nodes, %d leaf slots, %d triangles | slots with 1/2/3 triangles: %d / %d / %d | pairable 2-slots %.3f (area-weighted %.3f), 3-slots %.3f (%.3f)"
      % ("powerplant" if a.powerplant else "crown", len(nodes), int(tot), ntri, rep[1][0], rep[2][0], rep[3][0], rep[2][1], rep[2][2], rep[3][1], rep[3][2]))
print("      records / triangles = %.3f (unweighted), %.3f (area-weighted: what rays meet)" % (nrec / max(ntri, 1), wr / max(wt, 1e-300)))
used = (meta != 0).sum(1)
print("      used slots per node: " + " ".join("%d:%.3f" % (k, float((used == k).mean())) for k in range(1, 9)))
