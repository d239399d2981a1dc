"""Commits the crown stand-in a few times with the given device config (profiling helper, GPU box):  python tests/gpu_build_only.py "small_threshold=512" [reps] [quality]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from embree_amd import api, workloads as W
cfg = sys.argv[1] if len(sys.argv) > 1 else ""
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
q = int(sys.argv[3]) if len(sys.argv) > 3 else None
dev = api.Device(cfg)
meshes = W.synthetic_powerplant() if os.environ.get("PP") else W.synthetic_crown(num_phi=int(os.environ.get("PHI", "158")))
s = api.Scene(dev, 0, q)
for v, t in meshes:
    s.add_triangle_mesh(v, t, device_resident=True)
import time
ms = []; wall = []; ls = []
for i in range(reps):
    s.touch(); t0 = time.perf_counter(); s.commit(); wall.append((time.perf_counter() - t0) * 1e3)
    ii = s.info(); ms.append(ii["build_ms"]); ls.append("%d/%d" % (ii["num_launches"], ii["num_host_syncs"]))
    if os.environ.get("SLEEP"): time.sleep(float(os.environ["SLEEP"]))
print("WALL", " ".join("%.2f" % w for w in wall), "| launches/syncs", " ".join(ls))
i = s.info()
if os.environ.get("TREEHASH"):
    import hashlib
    nodes, tris = s.download_bvh()
    print("TREEHASH nodes %s tris %s" % (hashlib.sha256(nodes.tobytes()).hexdigest()[:16], hashlib.sha256(tris.tobytes()).hexdigest()[:16]))
print("BUILD cfg=%r q=%s: %s ms | nodes %d sah %.2f launches %d syncs %d top_levels %d depth %d" % (cfg, q, " ".join("%.2f" % m for m in ms), i["num_nodes"], i["sah"], i["num_launches"], i["num_host_syncs"], i.get("top_levels", -1), i.get("depth", -1)))
