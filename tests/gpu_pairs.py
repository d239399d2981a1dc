"""How many leaf slots of the crown stand-in's tree hold triangles that share an edge (what a 4-vertex pair record could serve)?  GPU box script."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from embree_amd import api, workloads as W
dev = api.Device(os.environ.get("CFG", "gpu=0"))
meshes = W.synthetic_powerplant(target_tris=2_000_000) if os.environ.get("PP") else W.synthetic_crown(num_phi=int(os.environ.get("PHI", "158")))
s = api.make_scene(dev, meshes)
nodes, tris = s.download_bvh()
meta = nodes["meta"].astype(np.int64); imask = nodes["imask"].astype(np.int64)
leaf = (meta != 0) & (((imask[:, None] >> np.arange(8)[None, :]) & 1) == 0)
bits = meta >> 5; cnt = np.where(leaf, np.where(bits == 1, 1, np.where(bits == 3, 2, 3)), 0)
first = nodes["triBase"].astype(np.int64)[:, None] + (meta & 31)
v0 = tris["v0"].astype(np.float32); v1 = v0 - tris["e1"]; v2 = v0 + tris["e2"]       # (v1, v2 recovered to an ulp: compare the ORIGINAL vertices instead)
# original vertices by id
vt = {}
P = np.zeros((tris.shape[0], 3, 3), np.float32)
for g, (v, t) in enumerate(meshes):
    m = tris["geomID"] == g
    P[m] = np.asarray(v, np.float32)[np.asarray(t)[tris["primID"][m]]]
key = P.view(np.uint32).astype(np.uint64)
vid = (key[..., 0] * np.uint64(0x9E3779B97F4A7C15) ^ key[..., 1] * np.uint64(0xC2B2AE3D27D4EB4F) ^ key[..., 2] * np.uint64(0x165667B19E3779F9))   # [tri][3] vertex hashes
def shared(a, b):
    return (vid[a][:, :, None] == vid[b][:, None, :]).any(2).sum(1)
h = {k: int((cnt == k).sum()) for k in (1, 2, 3)}
print("leaf slots by size:", h, "triangles", tris.shape[0], "nodes", nodes.shape[0])
f2 = first[cnt == 2]; s2 = shared(f2, f2 + 1)
print("2-triangle leaves sharing >= 2 vertices: %.3f (sharing exactly 1: %.3f)" % ((s2 >= 2).mean(), (s2 == 1).mean()))
f3 = first[cnt == 3]
s01, s02, s12 = shared(f3, f3 + 1), shared(f3, f3 + 2), shared(f3 + 1, f3 + 2)
anyp = (s01 >= 2) | (s02 >= 2) | (s12 >= 2)
fan = ((s01 >= 2).astype(int) + (s02 >= 2) + (s12 >= 2)) >= 2
print("3-triangle leaves with a pair sharing an edge: %.3f ; strips/fans (two shared edges): %.3f" % (anyp.mean(), fan.mean()))
rec_now = h[1] + 2 * h[2] + 3 * h[3]
rec_pair = h[1] + (s2 >= 2).sum() + 2 * (s2 < 2).sum() + 2 * anyp.sum() + 3 * (~anyp).sum()
print("records: %d -> %d (%.3f)" % (rec_now, rec_pair, rec_pair / rec_now))
# orientation: shared edge traversed in opposite directions (consistent winding)?
a, b = f2[s2 >= 2], f2[s2 >= 2] + 1
opp = 0
va, vb = vid[a], vid[b]
for i in range(3):
    for j in range(3):
        opp += ((va[:, i] == vb[:, (j + 1) % 3]) & (va[:, (i + 1) % 3] == vb[:, j])).sum()
print("of the sharing 2-leaves, shared edge in opposite direction: %.3f" % (opp / max(1, a.shape[0])))
# ray-weighted: tests per ray by leaf size
from embree_amd.rtypes import RAYHIT_DTYPE
prim = W.crown_camera_rays(meshes, 512, 512)
d = api.DeviceArray.from_numpy(prim); s.intersect1M_device(d.ptr, prim.shape[0]); api.load().mi355_device_synchronize(0)
rays = W.diffuse_bounce_rays(d.download(RAYHIT_DTYPE), meshes)
st = s.trace_stats(api.DeviceArray.from_numpy(rays).ptr, rays.shape[0], 96)
print("per ray: nodes %.2f tris %.2f" % (st["nodes"] / rays.shape[0], st["tris"] / rays.shape[0]))
