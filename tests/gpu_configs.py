"""Kernel-level numbers for every BASELINE.json config on one MI355X (GPU box, not a pytest file):
    python tests/gpu_configs.py > gpurun_out/configs.md
Closest hit / any hit through mi355_trace_timed (HIP events around the kernel, rays resident in HBM), lone launches and 4 launches in flight,
commit through rtcCommitScene.  The parity of the same configs is what tests/test_gpu_parity.py checks; this file only measures."""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
from embree_amd import api, workloads as W                       # noqa: E402
from embree_amd.rtypes import RAYHIT_DTYPE, rays_of               # noqa: E402

L = api.load()
dev = api.Device("")
streams = []
for _ in range(4):
    st = C.c_void_p()
    L.mi355_stream_create(0, C.byref(st))
    streams.append(st)


def rate(scene, rays, any_hit, reps=12, coherent=False):
    """(lone launches Mrays/s, 4 in flight Mrays/s); coherent: RTC_RAY_QUERY_FLAG_COHERENT = the wave-packet kernel"""
    M, rec = rays.shape[0], rays.dtype.itemsize
    pristine = api.DeviceArray.from_numpy(rays)
    bufs = [api.DeviceArray(rays.nbytes) for _ in range(reps)]
    out = []
    for ns in (1, 4):
        best = None
        for _ in range(3):
            for b in bufs:
                L.mi355_memcpy_d2d_async(b.ptr, pristine.ptr, rays.nbytes, None)
            L.mi355_device_synchronize(0)
            t0 = time.perf_counter()
            for k in range(reps):
                rc = L.mi355_trace_query(scene.bvh(), bufs[k].ptr, M, rec, int(any_hit), api.RTC_RAY_QUERY_FLAG_COHERENT if coherent else 0, streams[k % ns])
                assert rc == 0
            L.mi355_device_synchronize(0)
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        out.append(reps * M / best / 1e6)
    for b in bufs:
        b.free()
    pristine.free()
    return out


def commit_ms(scene, reps=3):
    ms = []
    for _ in range(reps):
        scene.touch()                                          # (a commit of an unmodified scene returns at once)
        scene.commit()
        ms.append(scene.info()["build_ms"])
    return min(ms)


def scene_of(meshes, quality=None):
    s = api.Scene(dev, 0, quality)
    for v, t in meshes:
        s.add_triangle_mesh(v, t, device_resident=True)
    s.commit()
    return s


rows = []
# configs[0]/[1]: cube + plane plumbing case and the Cornell box, coherent primary rays
m = W.cube_and_plane()
s = scene_of(m)
r = W.cube_camera_rays(1024, 1024)
a, b = rate(s, r, False)
rows.append(("configs[0] scene (cube + plane, 14 triangles), 2^20 coherent primary rays (the config itself is 1k rays on the CPU)", W.num_triangles(m), commit_ms(s), "closest", a, b))
a, b = rate(s, r, False, coherent=True)
rows.append(("  same, RTC_RAY_QUERY_FLAG_COHERENT (wave-packet kernel)", W.num_triangles(m), commit_ms(s), "closest", a, b))
s.release()
m = W.cornell_box()
s = scene_of(m)
r = W.cornell_camera_rays(1024, 1024)
a, b = rate(s, r, False)
rows.append(("configs[1] Cornell box, 2^20 coherent primary rays", W.num_triangles(m), commit_ms(s), "closest", a, b))
a, b = rate(s, r, False, coherent=True)
rows.append(("  same, RTC_RAY_QUERY_FLAG_COHERENT (wave-packet kernel)", W.num_triangles(m), commit_ms(s), "closest", a, b))
a, b = rate(s, rays_of(r), True, coherent=False)
rows.append(("  same rays, rtcOccluded", W.num_triangles(m), commit_ms(s), "any hit", a, b))
a, b = rate(s, rays_of(r), True, coherent=True)
rows.append(("  same rays, rtcOccluded, RTC_RAY_QUERY_FLAG_COHERENT", W.num_triangles(m), commit_ms(s), "any hit", a, b))
s.release()
# configs[2]/[3]: crown stand-in
m = W.synthetic_crown()
s = scene_of(m)
prim = W.crown_camera_rays(m, 1024, 1024)
tr = prim.copy()
s.intersect1M(tr)
bounce = W.diffuse_bounce_rays(tr, m)
cms = commit_ms(s)
a, b = rate(s, bounce, False)
rows.append(("configs[2] crown stand-in, 2^20 incoherent diffuse rays (the bench.py workload)", W.num_triangles(m), cms, "closest", a, b))
a, b = rate(s, prim, False)
rows.append(("  same scene, 2^20 coherent primary rays", W.num_triangles(m), cms, "closest", a, b))
a, b = rate(s, prim, False, coherent=True)
rows.append(("  same rays, RTC_RAY_QUERY_FLAG_COHERENT (wave-packet kernel: 4.5 triangles per pixel, the packets diverge)", W.num_triangles(m), cms, "closest", a, b))
bt = bounce.copy()
s.intersect1M(bt)
sh = W.shadow_rays(bt[: 1 << 17], m, samples=16)                # one rank's 2 Mi shadow-ray shard of configs[3]
a, b = rate(s, sh, True, reps=8)
rows.append(("configs[3] crown stand-in, one rank's shard of the 16 Mi shadow rays (2^21 rays)", W.num_triangles(m), cms, "any hit", a, b))
a, b = rate(s, sh, True, reps=8, coherent=True)
rows.append(("  same rays, RTC_RAY_QUERY_FLAG_COHERENT (16 consecutive rays share a hit point)", W.num_triangles(m), cms, "any hit", a, b))
s.release()
for q, qn in ((2, "RTC_BUILD_QUALITY_HIGH (spatial splits)"), (0, "RTC_BUILD_QUALITY_LOW (Morton codes)")):   # the other two build qualities on the bench's scene and rays
    s = scene_of(m, q)
    cq = commit_ms(s, 4)
    a, b = rate(s, bounce, False)
    rows.append(("  configs[2] scene and rays, %s" % qn, W.num_triangles(m), cq, "closest", a, b))
    s.release()
# configs[4]: powerplant stand-in
m = W.synthetic_powerplant()
s = scene_of(m)
prim = W.crown_camera_rays(m, 1024, 1024)
tr = prim.copy()
s.intersect1M(tr)
bounce = W.diffuse_bounce_rays(tr, m)
cms = commit_ms(s)
a, b = rate(s, bounce, False)
rows.append(("configs[4] powerplant stand-in, 2^20 incoherent diffuse rays", W.num_triangles(m), cms, "closest", a, b))
s.release()
s = scene_of(m, 2)
cq = commit_ms(s, 3)
a, b = rate(s, bounce, False)
rows.append(("  configs[4] scene and rays, RTC_BUILD_QUALITY_HIGH (spatial splits)", W.num_triangles(m), cq, "closest", a, b))
s.release()

print("| config | triangles | rtcCommitScene GPU ms | Mprims/s | query | Mrays/s, lone launches | Mrays/s, 4 launches in flight |")
print("|---|---:|---:|---:|---|---:|---:|")
for name, n, ms, q, a, b in rows:
    print("| %s | %d | %.2f | %.0f | %s | %.0f | %.0f |" % (name, n, ms, n / ms / 1e3, q, a, b))
