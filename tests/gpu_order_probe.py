"""How much does the ORDER of the rays in a batch matter to a lone launch? (GPU box tool, not a pytest file.)
The bench's 2^20 diffuse-bounce rays lie in image order (neighbouring rays start at neighbouring pixels: similar cost, shared nodes).  Times lone launches over the same rays
in their own order, reversed, with the 16-ray hand-out blocks shuffled, with 64-ray blocks shuffled, and fully shuffled; per-ray node visits of the first and the last
eighth say whether the batch ends on expensive rays.    python tests/gpu_order_probe.py [--reps 30]"""
import argparse, ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from embree_amd import api, workloads as W                       # noqa: E402
from embree_amd.rtypes import RAYHIT_DTYPE                       # noqa: E402
ap = argparse.ArgumentParser(); ap.add_argument("--reps", type=int, default=30); a = ap.parse_args()
L = api.load(); dev = api.Device("gpu=0")
meshes = W.synthetic_crown(num_phi=158)
s = api.Scene(dev)
for v, t in meshes:
    s.add_triangle_mesh(v, t, device_resident=True)
s.commit()
prim = W.crown_camera_rays(meshes, 1024, 1024)
d = api.DeviceArray.from_numpy(prim); s.intersect1M_device(d.ptr, prim.shape[0]); L.mi355_device_synchronize(0)
tr = d.download(RAYHIT_DTYPE); d.free()
rays = W.diffuse_bounce_rays(tr, meshes, seed=1)
n = rays.shape[0]; rec = rays.dtype.itemsize
e0, e1 = C.c_void_p(), C.c_void_p(); L.mi355_event_create(C.byref(e0)); L.mi355_event_create(C.byref(e1))
work = api.DeviceArray(rays.nbytes)
rng = np.random.default_rng(5)
def blocks(k):
    p = rng.permutation(n // k); return (p[:, None] * k + np.arange(k)[None, :]).reshape(-1)
orders = {"image order": np.arange(n), "reversed": np.arange(n)[::-1].copy(), "16-ray blocks shuffled": blocks(16), "64-ray blocks shuffled": blocks(64),
          "4096-ray blocks shuffled": blocks(4096), "rays shuffled": rng.permutation(n)}
for name, idx in orders.items():
    r = np.ascontiguousarray(rays[idx]); pristine = api.DeviceArray.from_numpy(r); ms = []
    for i in range(a.reps + 3):
        L.mi355_memcpy_d2d_async(work.ptr, pristine.ptr, n * rec, None)
        assert L.mi355_trace_timed(s.bvh(), work.ptr, n, rec, 0, None, e0, e1) == 0
        t = C.c_float(); L.mi355_event_elapsed_ms(e0, e1, C.byref(t))
        if i >= 3: ms.append(t.value)
    L.mi355_memcpy_d2d_async(work.ptr, pristine.ptr, n * rec, None)
    st = s.trace_stats(work.ptr, n, rec, False)
    ms = np.array(ms)
    print("ORDER %-26s us min %.1f median %.1f -> %.0f Mrays/s | wave iterations %d, nodes / ray %.2f" % (name, 1e3 * ms.min(), 1e3 * float(np.median(ms)), n / (1e3 * float(np.median(ms))), st["wave_iters"], st["nodes"] / n), flush=True)
    pristine.free()
