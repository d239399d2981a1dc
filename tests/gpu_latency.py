import sys, time, numpy as np
sys.path.insert(0, '.')
from embree_amd import api, workloads as W
from embree_amd.rtypes import make_rayhits
for cfg in ("", "small_in_place=0"):
    dev = api.Device(cfg)
    m = W.synthetic_crown(num_phi=32)
    s = api.make_scene(dev, m)
    r = make_rayhits(np.float32([[0.1, 0.2, 5.0]]), np.float32([[0, 0, -1]]))
    for _ in range(20):
        q = r.copy(); s.intersect1(q)
    ts = []
    for _ in range(300):
        q = r.copy(); t0 = time.perf_counter(); s.intersect1(q); ts.append(time.perf_counter() - t0)
    q16 = np.repeat(r, 16)
    t16 = []
    for _ in range(100):
        q = q16.copy(); t0 = time.perf_counter(); s.intersect1M(q); t16.append(time.perf_counter() - t0)
    print("cfg %-18r rtcIntersect1 median %.1f us min %.1f us | 16 rays median %.1f us | hit geom %d prim %d t %.6f" % (cfg, 1e6 * np.median(ts), 1e6 * min(ts), 1e6 * np.median(t16), q["geomID"][0], q["primID"][0], q["tfar"][0]))
    s.release(); dev.release()
