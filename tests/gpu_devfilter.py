"""Device filter function smoke run (GPU box, not a pytest file; run under LD_PRELOAD=tools/segv_trace.so to get a backtrace of a host-side crash)."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from embree_amd import api
from embree_amd.rtypes import RAYHIT_DTYPE
from tests.test_gpu_round3 import _rule_scene_meshes, _rule_rays
L = api.load()
lib = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "_bin", "libdevfilter.so"))
lib.devfilter_address.restype = C.c_uint64
lib.devfilter_address_of.restype = C.c_uint64
fn = lib.devfilter_address_of(int(os.environ.get("WHICH", "0")))
print("device function address: 0x%x" % fn, flush=True)
meshes, rays = _rule_scene_meshes(), _rule_rays()
dev = api.Device("gpu=0,device_filter_functions=1")
FLAGS = 4 if os.environ.get("ROBUST") else 0
s = api.make_scene(dev, meshes, flags=FLAGS)
plain = rays.copy(); s.intersect1M(plain)
counters = api.DeviceArray.from_numpy(np.zeros(3, np.uint64))
d = api.DeviceArray.from_numpy(rays)
qa = api.QueryArguments(None, api.RTC_RAY_QUERY_FLAG_INVOKE_ARGUMENT_FILTER)
qa.filter, qa.context = C.c_void_p(fn), C.c_void_p(counters.ptr)
print("launching", flush=True)
s.intersect1M_device(d.ptr, rays.shape[0], args=qa)
L.mi355_device_synchronize(0)
got = d.download(RAYHIT_DTYPE)
print("calls / rejected / userptr sum:", counters.download(np.uint64), "changed rays:", int((got["primID"] != plain["primID"]).sum()), "hits", int((got["geomID"] != 0xFFFFFFFF).sum()), "plain hits", int((plain["geomID"] != 0xFFFFFFFF).sum()), flush=True)
# the same rule on the host (what the function implements): reject (primID + 2 geomID) % 5 == 1 -- via the host-array entry point with a host callback
FILTER_FN = C.CFUNCTYPE(None, C.c_void_p)
def host_rule(p):
    a = C.cast(p, C.POINTER(api.FilterArgs)).contents if hasattr(api, "FilterArgs") else None
if os.environ.get("ROBUST") or os.environ.get("SMALL_ONLY"):
    sys.exit(0)

# ---- what the call costs: the bench workload (crown stand-in, 2^20 diffuse rays) without a function, and with this one enforced on every candidate
import time
from embree_amd import workloads as W
m = W.synthetic_crown()
sc = api.make_scene(dev, m)
prim = W.crown_camera_rays(m, 1024, 1024)
sc.intersect1M(prim)
big = W.diffuse_bounce_rays(prim, m, seed=1)
pr = api.DeviceArray.from_numpy(big)
wk = api.DeviceArray(big.nbytes)
for label, flags_, f in (("no function", 0, None), ("function enforced on every candidate", api.RTC_RAY_QUERY_FLAG_INVOKE_ARGUMENT_FILTER, fn)):
    a2 = api.QueryArguments(None, flags_)
    if f:
        a2.filter, a2.context = C.c_void_p(f), None
    best = 1e9
    for _ in range(6):
        L.mi355_memcpy_d2d_async(wk.ptr, pr.ptr, big.nbytes, None); L.mi355_device_synchronize(0)
        t0 = time.perf_counter()
        sc.intersect1M_device(wk.ptr, big.shape[0], args=a2)
        L.mi355_device_synchronize(0)
        best = min(best, time.perf_counter() - t0)
    print("DEVFILTER %-40s %.3f ms = %.0f Mrays/s" % (label, best * 1e3, big.shape[0] / best / 1e6), flush=True)
