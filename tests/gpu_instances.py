"""Instancing at the bench's scale (GPU box, not a pytest file):  python tests/gpu_instances.py
The crown stand-in's lattice, but ONE noisy sphere (99,224 triangles) instanced 48 times (uniform scale + rotation about y + translation) inside the
room (the scene's own geometry), against the same scene flattened on the host (every instance's vertices transformed, 4.76 M triangles): 2^20 camera
rays and 2^20 diffuse bounce rays, closest hit and any hit.  Hits must agree (same instance / geometry, t within 1e-4; the flattened scene computes
with world-space vertices, so bits differ)."""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
from embree_amd import api, workloads as W                       # noqa: E402
from embree_amd.rtypes import rays_of, INVALID_ID                 # noqa: E402

L = api.load()
dev = api.Device("")
nx, ny, nz = 4, 4, 3
rng = np.random.default_rng(5)
sv, st = W.triangle_sphere(np.zeros(3, np.float32), 1.0, 158, noise=0.15, seed=3)
room = W._box_room([-0.25, -0.25, -0.25], [nx + 0.25, ny + 0.25, nz + 0.25])
xf, flat = [], []
for ix in range(nx):
    for iy in range(ny):
        for iz in range(nz):
            c = np.array([ix + 0.5, iy + 0.5, iz + 0.5]) + (rng.random(3) - 0.5) * 0.3
            r, a = 0.30 + 0.12 * rng.random(), rng.random() * 6.28
            m = np.array([[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]]) * r
            x = np.concatenate([m.T.reshape(9), c]).astype(np.float32)
            xf.append(x)
            M = x[:9].reshape(3, 3).T
            flat.append(((sv.astype(np.float32) @ M.T.astype(np.float32) + x[9:]).astype(np.float32), st))
flat.append(room)

obj = api.Scene(dev)
obj.add_triangle_mesh(sv, st, device_resident=True)
obj.commit()
top = api.Scene(dev)
for x in xf:
    top.add_instance(obj, x)
room_id = top.add_triangle_mesh(room[0], room[1], device_resident=True)
t0 = time.perf_counter(); top.commit(); wall = (time.perf_counter() - t0) * 1e3
ti = top.info()
fs = api.Scene(dev)
for v, t in flat:
    fs.add_triangle_mesh(v, t, device_resident=True)
fs.commit()
fi = fs.info()
print("INST object build %.2f ms (%d tris) | top commit %.2f ms wall (top tree %.3f ms GPU, %d instances + own geometry; combined %d nodes, %.1f MB) | flat build %.2f ms (%d tris, %.1f MB)"
      % (obj.info()["build_ms"], obj.info()["num_triangles"], wall, ti["build_ms"], len(xf), ti["num_nodes"], (ti["bytes_nodes"] + ti["bytes_triangles"]) / 1e6,
         fi["build_ms"], fi["num_triangles"], (fi["bytes_nodes"] + fi["bytes_triangles"]) / 1e6))

prim = W.crown_camera_rays(flat, 1024, 1024)
tr = prim.copy(); fs.intersect1M(tr)
bounce = W.diffuse_bounce_rays(tr, flat)
e0, e1 = C.c_void_p(), C.c_void_p()
L.mi355_event_create(C.byref(e0)); L.mi355_event_create(C.byref(e1))


def rate(scene, rays, any_hit):
    d = api.DeviceArray.from_numpy(rays)
    best = 1e9
    for _ in range(5):
        L.mi355_memcpy_h2d(d.ptr, rays.ctypes.data, rays.nbytes)
        L.mi355_trace_timed(scene.bvh(), d.ptr, rays.shape[0], rays.dtype.itemsize, int(any_hit), None, e0, e1)
        ms = C.c_float(); L.mi355_event_elapsed_ms(e0, e1, C.byref(ms)); best = min(best, ms.value)
    out = d.download(rays.dtype)
    d.free()
    return rays.shape[0] / best / 1e3, out


for name, rays in (("primary", prim), ("diffuse", bounce)):
    ri, gi = rate(top, rays, False)
    rf, gf = rate(fs, rays, False)
    hit = gf["geomID"] != INVALID_ID
    assert ((gi["geomID"] != INVALID_ID) == hit).mean() > 0.9999
    both = hit & (gi["geomID"] != INVALID_ID)
    inst_of = np.where(gi["instID"] == INVALID_ID, len(xf), gi["instID"])          # flat geomID k = instance k, room = last
    agree = (inst_of[both] == gf["geomID"][both]) & (np.abs(gi["tfar"][both] - gf["tfar"][both]) <= 1e-4 * np.abs(gf["tfar"][both]) + 1e-6)
    oi, _ = rate(top, rays_of(rays), True)
    of, _ = rate(fs, rays_of(rays), True)
    st_i = top.trace_stats(api.DeviceArray.from_numpy(rays).ptr, rays.shape[0], 96)
    st_f = fs.trace_stats(api.DeviceArray.from_numpy(rays).ptr, rays.shape[0], 96)
    for tag, q in (("inst", st_i), ("flat", st_f)):
        li = 64.0 * q["wave_iters"]
        print("CENSUS %s %-8s wave_iters %d node-step util %.2f | lanes: node %.3f idle %.3f wait_batch %.3f wait_drain %.3f blocked %.3f | spills %d depth %d"
              % (tag, name, q["wave_iters"], q["nodes"] / max(1, 64 * q["node_blocks"]), q["nodes"] / li, q["lanes_idle"] / li, q["lanes_wait_batch"] / li,
                 q["lanes_wait_drain"] / li, q["lanes_blocked"] / li, q["spills"], q["max_depth"]))
    print("INST %-8s closest %.0f Mrays/s (flat %.0f) | any hit %.0f (flat %.0f) | same primitive and t: %.5f of %d hits | nodes/ray %.1f tris/ray %.1f"
          % (name, ri, rf, oi, of, agree.mean(), both.sum(), st_i["nodes"] / st_i["rays"], st_i["tris"] / st_i["rays"]))
