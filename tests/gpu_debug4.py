"""debug: which side misses in the powerplant full-size comparison"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from embree_amd import api, workloads as W
from embree_amd.rtypes import INVALID_ID, rays_of
from oracle import restate, refembree
dev = api.Device("gpu=0")
m = W.synthetic_powerplant()
s = api.make_scene(dev, m, device_resident=True)
R = refembree.RefScene("threads=64")
for v, t in m: R.add_mesh(v, t)
R.commit()
o = restate.OracleScene()
for v, t in m: o.add_mesh(v, t)
lo, hi = W.scene_bounds(m)
rays = W.incoherent_rays(1 << 20, (lo + hi) / 2, seed=11)
want, got = rays.copy(), rays.copy()
R.intersect1(want, 64)
s.intersect1M(got)
rob = api.make_scene(dev, m, flags=api.RTC_SCENE_FLAG_ROBUST, device_resident=True)
gr = rays.copy(); rob.intersect1M(gr)
diff = (got["primID"] != want["primID"]) | (got["geomID"] != want["geomID"])
idx = np.nonzero(diff)[0]
print("differing rays", idx.size)
gh, wh = got["geomID"][idx] != INVALID_ID, want["geomID"][idx] != INVALID_ID
print("gpu miss / ref hit", int((~gh & wh).sum()), " gpu hit / ref miss", int((gh & ~wh).sum()), " both hit", int((gh & wh).sum()))
b = idx[gh & wh]
tg = o.triangle_t(rays[b], got["geomID"][b], got["primID"][b])
tw = o.triangle_t(rays[b], want["geomID"][b], want["primID"][b])
rel = np.abs(tg - tw) / np.maximum(np.abs(tw), 1e-30)
tie = rel <= 1e-4
print("both hit: ties", int(tie.sum()), "gpu closer", int(((tg < tw) & ~tie).sum()), "ref closer", int(((tg > tw) & ~tie).sum()))
bad = b[~tie]
for i in bad[:12]:
    print(i, "gpu", got["geomID"][i], got["primID"][i], got["tfar"][i], "ref", want["geomID"][i], want["primID"][i], want["tfar"][i],
          "robustGPU", gr["geomID"][i], gr["primID"][i], gr["tfar"][i], "dir", rays["dir_x"][i], rays["dir_y"][i], rays["dir_z"][i])
# the ROBUST reference as a third opinion
R2 = refembree.RefScene("threads=64", flags=4)
for v, t in m: R2.add_mesh(v, t)
R2.commit()
w2 = rays.copy(); R2.intersect1(w2, 64)
for name, a in (("gpu fast", got), ("ref fast", want), ("gpu robust", gr)):
    d2 = (a["primID"] != w2["primID"]) | (a["geomID"] != w2["geomID"])
    closer = (a["tfar"] < w2["tfar"] * (1 - 1e-4)); farther = (a["tfar"] > w2["tfar"] * (1 + 1e-4))
    print(name, "vs robust reference: id diffs", int(d2.sum()), "t closer", int(closer.sum()), "t farther (missed something)", int(farther.sum()))
