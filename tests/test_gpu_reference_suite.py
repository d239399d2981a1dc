"""More of the reference's own test programme (tutorials/verify/verify.cpp), through the C ABI on the GPU (pytest -m gpu):
QuadHitTest :2549, UpdateTest :1835, GarbageGeometryTest :1915, UserGeometryIDTest :1591, NewDeleteGeometryTest :1523 / IntensiveRegressionTest :5298
(reduced), MultipleDevicesTest :764, GetUserDataTest :865, EmptyGeometryTest :1086, OverlappingGeometryTest :1209, DisableAndDetachGeometryTest :1700.
Expected values are the reference's (16 ulp on t/u/v/Ng where it states them); everything else is "no error, no hang, right geometry hit"."""
import ctypes as C

import numpy as np
import pytest

from embree_amd import workloads as W
from embree_amd.rtypes import make_rayhits, rays_of, INVALID_ID

pytestmark = pytest.mark.gpu
ULP = np.float32(1.1920929e-07)


@pytest.fixture(scope="module")
def api():
    from embree_amd import api as A
    A.load()
    assert A.load().mi355_device_count() > 0, "no HIP device: the product has no CPU fallback"
    return A


@pytest.fixture(scope="module")
def dev(api):
    d = api.Device("gpu=0")
    yield d
    d.release()


@pytest.mark.parametrize("flags", [0, 4])
def test_quad_hit(api, dev, flags):
    """QuadHitTest: unit quad, 256 rays from (0,0,-1) to v0 + u (v1-v0) + v (v3-v0): primID 0, u, v, t = 1, Ng = (0,0,1) within 16 ulp."""
    rng = np.random.default_rng(3)
    u, v = rng.random(256, dtype=np.float32), rng.random(256, dtype=np.float32)
    edge = (u < 0.001) | (v < 0.001) | (u > 0.999) | (v > 0.999)
    u[edge], v[edge] = 0.333, 0.333
    s = api.Scene(dev, flags)
    s.add_quad_mesh(np.array([[0, 0, 0], [1, 0, 0], [1, 1, 0], [0, 1, 0]], np.float32), np.array([[0, 1, 2, 3]], np.uint32))
    s.commit()
    org = np.tile(np.array([[0, 0, -1]], np.float32), (256, 1))
    to = np.stack([u, v, np.zeros_like(u)], -1)
    rh = make_rayhits(org, to - org)
    s.intersect1M(rh)
    assert (rh["primID"] == 0).all() and (rh["geomID"] == 0).all()
    assert np.abs(rh["u"] - u).max() <= 16 * ULP and np.abs(rh["v"] - v).max() <= 16 * ULP and np.abs(rh["tfar"] - 1).max() <= 16 * ULP
    assert np.abs(rh["Ng_x"]).max() <= 16 * ULP and np.abs(rh["Ng_y"]).max() <= 16 * ULP and np.abs(rh["Ng_z"] - 1).max() <= 16 * ULP
    r = rays_of(make_rayhits(org, to - org))
    s.occluded1M(r)
    assert np.isneginf(r["tfar"]).all()
    s.release()


def test_update_geometry_buffers(api, dev):
    """UpdateTest: a triangle sphere and a quad sphere whose vertex buffers are moved in place (rtcGetGeometryBufferData, rtcUpdateGeometryBuffer,
    rtcCommitGeometry, rtcCommitScene); after every step a ray from above must hit the geometry it is aimed at, closest hit and occlusion."""
    L = api.load()
    s = api.Scene(dev)
    num_phi = 10
    tv, tt = W.triangle_sphere([-10, 0, -10], 1.0, num_phi)
    # quad sphere: the same rings as quads (poles stay triangles in the reference; a quad with a repeated vertex is legal)
    nt = 2 * num_phi
    q = []
    for ph in range(num_phi):
        for th in range(nt):
            a, b = ph * nt + th, ph * nt + (th + 1) % nt
            q.append([a, b, b + nt, a + nt])
    qv, _ = W.triangle_sphere([-10, 0, 10], 1.0, num_phi)
    geoms, pos = [], [np.array([-10, 0, -10], np.float32), np.array([-10, 0, 10], np.float32)]
    for kind, (v, idx) in (("tri", (tv, tt)), ("quad", (qv, np.array(q, np.uint32)))):
        g = L.rtcNewGeometry(dev.h, api.RTC_GEOMETRY_TYPE_TRIANGLE if kind == "tri" else api.RTC_GEOMETRY_TYPE_QUAD)
        pv = L.rtcSetNewGeometryBuffer(g, api.RTC_BUFFER_TYPE_VERTEX, 0, api.RTC_FORMAT_FLOAT3, 12, v.shape[0])
        pi = L.rtcSetNewGeometryBuffer(g, api.RTC_BUFFER_TYPE_INDEX, 0, api.RTC_FORMAT_UINT3 if kind == "tri" else api.RTC_FORMAT_UINT4,
                                       12 if kind == "tri" else 16, idx.shape[0])
        dev.check()
        C.memmove(pv, np.ascontiguousarray(v).ctypes.data, v.nbytes)
        C.memmove(pi, np.ascontiguousarray(idx).ctypes.data, idx.nbytes)
        L.rtcCommitGeometry(g)
        gid = L.rtcAttachGeometry(s.h, g)
        L.rtcReleaseGeometry(g)
        geoms.append((gid, v.shape[0]))
    for step in range(8):
        for k, (gid, nv) in enumerate(geoms):
            if step & (1 << k):
                h = L.rtcGetGeometry(s.h, gid)
                p = L.rtcGetGeometryBufferData(h, api.RTC_BUFFER_TYPE_VERTEX, 0)
                arr = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_float)), shape=(nv, 3))
                arr += np.array([2, 0.1, 2], np.float32)
                L.rtcUpdateGeometryBuffer(h, api.RTC_BUFFER_TYPE_VERTEX, 0)
                L.rtcCommitGeometry(h)
                pos[k] = pos[k] + np.array([2, 0.1, 2], np.float32)
        s.commit()
        org = np.stack([p + np.array([0.1, 10, 0.1], np.float32) for p in pos] * 25)
        rh = make_rayhits(org, np.tile(np.array([[0, -1, 0]], np.float32), (org.shape[0], 1)))
        s.intersect1M(rh)
        assert (rh["geomID"] == np.array([g for g, _ in geoms] * 25, np.uint32)).all(), step
        r = rays_of(make_rayhits(org, np.tile(np.array([[0, -1, 0]], np.float32), (org.shape[0], 1))))
        s.occluded1M(r)
        assert np.isneginf(r["tfar"]).all()
    s.release()


def test_garbage_geometry(api, dev):
    """GarbageGeometryTest: meshes filled with random BITS (NaNs, infinities, denormals, huge values, out-of-range indices), triangles and quads,
    fast / robust / low-quality scenes: commit and queries must come back without an error and without hanging."""
    rng = np.random.default_rng(23565)
    for i in range(24):
        s = api.Scene(dev, api.RTC_SCENE_FLAG_ROBUST if i % 3 == 1 else 0, api.RTC_BUILD_QUALITY_LOW if i % 4 == 3 else None)
        total = 0
        for j in range(8):
            n = int(rng.integers(0, 256))
            nv = max(3 * n, 4)
            v = rng.integers(0, 2 ** 32, size=(nv, 3), dtype=np.uint64).astype(np.uint32).view(np.float32)
            if j % 2:
                idx = rng.integers(0, 2 ** 32, size=(n, 4 if j % 4 == 1 else 3), dtype=np.uint64).astype(np.uint32)      # garbage topology too
            else:
                idx = rng.integers(0, nv, size=(n, 4 if j % 4 == 0 else 3), dtype=np.uint64).astype(np.uint32)
            (s.add_quad_mesh if idx.shape[1] == 4 else s.add_triangle_mesh)(v, idx)
            total += n
        s.commit()
        rays = W.incoherent_rays(2000, [0, 0, 0], seed=i)
        s.intersect1M(rays)
        r = rays_of(W.incoherent_rays(2000, [0, 0, 0], seed=i))
        s.occluded1M(r)
        assert s.info()["num_triangles"] <= 2 * total
        s.release()
    dev.check()


def test_attach_by_id_user_data_and_detach(api, dev):
    """UserGeometryIDTest (rtcAttachGeometryByID), GetUserDataTest, DisableAndDetachGeometryTest: IDs chosen by the caller come back in the hits,
    user data round-trips, a detached / disabled geometry is not hit after the next commit and its ID can be reused."""
    L = api.load()
    L.rtcSetGeometryUserData.argtypes = [C.c_void_p, C.c_void_p]
    L.rtcGetGeometryUserData.restype = C.c_void_p
    L.rtcGetGeometryUserData.argtypes = [C.c_void_p]
    s = api.Scene(dev)
    keep, ids = [], [7, 3, 1000, 42]
    for k, gid in enumerate(ids):
        v = np.array([[0, 0, k], [1, 0, k], [0, 1, k], [0, 0, 0]], np.float32)
        t = np.array([[0, 1, 2]], np.uint32)
        keep += [v, t]
        g = L.rtcNewGeometry(dev.h, api.RTC_GEOMETRY_TYPE_TRIANGLE)
        L.rtcSetSharedGeometryBuffer(g, api.RTC_BUFFER_TYPE_VERTEX, 0, api.RTC_FORMAT_FLOAT3, v.ctypes.data, 0, 12, 3)
        L.rtcSetSharedGeometryBuffer(g, api.RTC_BUFFER_TYPE_INDEX, 0, api.RTC_FORMAT_UINT3, t.ctypes.data, 0, 12, 1)
        L.rtcSetGeometryUserData(g, C.c_void_p(0x1000 + gid))
        L.rtcCommitGeometry(g)
        L.rtcAttachGeometryByID(s.h, g, gid)
        L.rtcReleaseGeometry(g)
    dev.check()
    s.commit()
    for gid in ids:
        assert L.rtcGetGeometryUserData(L.rtcGetGeometry(s.h, gid)) == 0x1000 + gid
    org = np.array([[0.2, 0.2, k - 0.5] for k in range(4)], np.float32)
    rh = make_rayhits(org, np.tile(np.array([[0, 0, 1]], np.float32), (4, 1)))
    s.intersect1M(rh)
    assert (rh["geomID"] == np.array(ids, np.uint32)).all() and np.allclose(rh["tfar"], 0.5)
    L.rtcDetachGeometry(s.h, 3)                                   # the triangle at z = 1 goes away ...
    L.rtcDisableGeometry(L.rtcGetGeometry(s.h, 1000))             # ... and the one at z = 2 is switched off
    s.commit()
    rh = make_rayhits(org, np.tile(np.array([[0, 0, 1]], np.float32), (4, 1)))
    s.intersect1M(rh)
    assert list(rh["geomID"]) == [7, 42, 42, 42] and np.allclose(rh["tfar"], [0.5, 2.5, 1.5, 0.5])
    assert s.add_triangle_mesh(keep[0], keep[1]) in (0, 3)         # a free ID is handed out again (lowest free one)
    dev.check()
    s.release()


def test_many_scenes_devices_and_empty_geometry(api, dev):
    """MultipleDevicesTest + NewDeleteGeometryTest / IntensiveRegressionTest (reduced) + EmptyGeometryTest + OverlappingGeometryTest: several devices
    and scenes alive at once, geometries created, committed, detached and re-created in a loop, empty geometries, 20,000 overlapping copies."""
    L = api.load()
    devs = [api.Device("gpu=0") for _ in range(3)]
    scenes = []
    rng = np.random.default_rng(9)
    for d in devs:
        s = api.Scene(d)
        g = L.rtcNewGeometry(d.h, api.RTC_GEOMETRY_TYPE_TRIANGLE)          # EmptyGeometryTest: no buffers at all
        L.rtcCommitGeometry(g)
        L.rtcAttachGeometry(s.h, g)
        L.rtcReleaseGeometry(g)
        s.add_triangle_mesh(np.zeros((0, 3), np.float32), np.zeros((0, 3), np.uint32))   # zero-sized buffers
        s.commit()
        d.check()
        assert s.info()["num_triangles"] == 0
        scenes.append(s)
    for it in range(12):                                             # geometry churn on every device
        for d, s in zip(devs, scenes):
            n = int(rng.integers(1, 400))
            c = rng.random((n, 1, 3), dtype=np.float32)
            v = (c + 0.05 * rng.random((n, 3, 3), dtype=np.float32)).reshape(-1, 3).astype(np.float32)
            gid = s.add_triangle_mesh(v, np.arange(3 * n, dtype=np.uint32).reshape(-1, 3))
            if it % 3 == 2:
                L.rtcDetachGeometry(s.h, gid)
            s.commit()
            r = W.incoherent_rays(512, [0.5, 0.5, 0.5], seed=it)
            s.intersect1M(r)
            d.check()
    base = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0]], np.float32)   # OverlappingGeometryTest
    s = api.Scene(devs[0])
    for _ in range(4):
        s.add_triangle_mesh(np.tile(base, (5000, 1)), np.arange(15000, dtype=np.uint32).reshape(-1, 3))
    s.commit()
    rh = make_rayhits(np.array([[0.2, 0.2, -1]], np.float32), np.array([[0, 0, 1]], np.float32))
    s.intersect1M(rh)
    assert rh["geomID"][0] != INVALID_ID and rh["tfar"][0] == 1.0
    s.release()
    for s in scenes:
        s.release()
    for d in devs:
        d.release()


def test_memory_monitor(api):
    """MemoryMonitorTest (verify.cpp:5378): rtcSetDeviceMemoryMonitorFunction sees every allocation the library keeps (+bytes) and every release
    (-bytes): the sum is zero once everything is released; a callback answering false makes the request fail with RTC_ERROR_OUT_OF_MEMORY, leaves
    the accounting balanced and the previously committed tree usable."""
    L = api.load()
    MON = C.CFUNCTYPE(C.c_bool, C.c_void_p, C.c_ssize_t, C.c_bool)
    L.rtcSetDeviceMemoryMonitorFunction.argtypes = [C.c_void_p, MON, C.c_void_p]
    state = dict(used=0, calls=0, break_at=None, broke=0)

    def monitor(ptr, nbytes, post):
        state["calls"] += 1
        if nbytes > 0 and state["break_at"] is not None and state["calls"] >= state["break_at"]:
            state["broke"] += 1
            return False
        state["used"] += nbytes
        return True
    cb = MON(monitor)
    d = api.Device("gpu=0")
    L.rtcSetDeviceMemoryMonitorFunction(d.h, cb, None)
    meshes = W.synthetic_crown(num_phi=12)
    s = api.Scene(d)
    for v, t in meshes[:6]:
        s.add_triangle_mesh(v, t, shared=False)                  # library-owned buffers (rtcSetNewGeometryBuffer)
    s.commit()
    d.check()
    info = s.info()
    assert state["used"] >= info["bytes_nodes"] + info["bytes_triangles"] > 0
    calls_ok = state["calls"]
    rays = W.incoherent_rays(2000, [2, 2, 1.5], seed=1)
    want = rays.copy()
    s.intersect1M(want)
    # now refuse the next allocation: the commit fails, the old tree keeps answering
    s.touch()                                                  # (a commit of an unmodified scene returns at once and allocates nothing)
    state["break_at"] = state["calls"] + 1
    L.rtcCommitScene(s.h)
    assert d.get_error() == api.RTC_ERROR_OUT_OF_MEMORY and state["broke"] == 1
    state["break_at"] = None
    got = rays.copy()
    s.intersect1M(got)
    assert got.tobytes() == want.tobytes()
    s.release()
    assert state["used"] == 0 and state["calls"] > calls_ok
    # a refused buffer allocation
    state["break_at"] = state["calls"] + 1
    g = L.rtcNewGeometry(d.h, api.RTC_GEOMETRY_TYPE_TRIANGLE)
    p = L.rtcSetNewGeometryBuffer(g, api.RTC_BUFFER_TYPE_VERTEX, 0, api.RTC_FORMAT_FLOAT3, 12, 1000)
    assert not p and d.get_error() == api.RTC_ERROR_OUT_OF_MEMORY
    state["break_at"] = None
    L.rtcReleaseGeometry(g)
    assert state["used"] == 0
    L.rtcSetDeviceMemoryMonitorFunction(d.h, C.cast(None, MON), None)
    d.release()


def _wobble(meshes, step):
    """the same topology, moved vertices: every sphere breathes and drifts (a deforming-mesh frame)"""
    out = []
    for k, (v, t) in enumerate(meshes):
        c = v.mean(0)
        s = np.float32(1.0 + 0.25 * np.sin(0.9 * step + k))
        d = np.array([np.sin(step + k), np.cos(1.3 * step + 2 * k), np.sin(0.7 * step - k)], np.float32) * np.float32(0.35)
        out.append((((v - c) * s + c + d).astype(np.float32), t))
    return out


def _same_hits(a, b):
    """two trees over the same triangles: identical records except where two triangles tie on t (which one is reported depends on the leaf order)"""
    same = (a["geomID"] == b["geomID"]) & (a["primID"] == b["primID"])
    assert (a["tfar"][~same] == b["tfar"][~same]).all(), "different triangle without a tie on t"
    assert (~same).sum() <= max(4, a.shape[0] // 2000)
    for f in ("tfar", "u", "v", "Ng_x", "Ng_y", "Ng_z"):
        assert (a[f][same] == b[f][same]).all(), f


@pytest.mark.parametrize("flags", [0, 4])
def test_refit_equals_rebuild(api, dev, flags):
    """RTC_BUILD_QUALITY_REFIT (rtcSetGeometryBuildQuality.md; BVHNRefitT, kernels/bvh/bvh_refit.cpp): after rtcUpdateGeometryBuffer(VERTEX) the
    commit keeps the topology of the tree and refits it.  The refitted scene must answer like a scene built from scratch on the moved vertices --
    bit for bit, closest hit and any hit -- the scene bounds must follow, and what cannot be refitted (new index data, a triangle gone invalid)
    must fall back to a rebuild."""
    L = api.load()
    base = [W.triangle_sphere(np.array([x * 2.2, 0.4 * x, 0.3 * (x % 2)], np.float32), 1.0, 40, noise=0.1, seed=7 + x) for x in range(5)]
    s = api.Scene(dev, flags)
    gids = [s.add_triangle_mesh(v, t, shared=False) for v, t in base]
    for g in gids[:4]:
        s.set_geometry_build_quality(g, api.RTC_BUILD_QUALITY_REFIT)
    s.commit()
    i0 = s.info()
    assert i0["num_refits"] == 0 and i0["bytes_refit"] == 8 * i0["num_triangles"]
    rng = np.random.default_rng(11)
    org = (rng.random((60000, 3), dtype=np.float32) - 0.5) * np.array([16, 8, 8], np.float32) + np.array([4.4, 0.8, 0], np.float32)
    tgt = (rng.random((60000, 3), dtype=np.float32) - 0.5) * np.array([10, 2, 2], np.float32) + np.array([4.4, 0.8, 0], np.float32)
    refits = 0
    for step in range(1, 4):
        moved = _wobble(base, step)
        moved[4] = base[4]                                      # the MEDIUM-quality mesh stays: only REFIT meshes changed
        for g in gids[:4]:
            s.update_vertices(g, moved[g][0])
        s.commit()
        i1 = s.info()
        refits += 1
        assert i1["num_refits"] == refits and i1["num_nodes"] == i0["num_nodes"], "the commit rebuilt instead of refitting"
        fresh = api.make_scene(dev, moved, flags=flags)
        lo, hi = s.bounds(); flo, fhi = fresh.bounds()
        assert (lo == flo).all() and (hi == fhi).all()
        a, b = make_rayhits(org, tgt - org), make_rayhits(org, tgt - org)
        s.intersect1M(a); fresh.intersect1M(b)
        assert (a["geomID"] != INVALID_ID).sum() > 20000
        _same_hits(a, b)
        ra, rb = rays_of(make_rayhits(org, tgt - org)), rays_of(make_rayhits(org, tgt - org))
        s.occluded1M(ra); fresh.occluded1M(rb)
        assert (np.isneginf(ra["tfar"]) == np.isneginf(rb["tfar"])).all()
        fresh.release()
    # a mesh of MEDIUM quality moved: full rebuild (num_refits restarts with the new tree)
    moved = _wobble(base, 5)
    s.update_vertices(gids[4], moved[4][0])
    s.commit()
    assert s.info()["num_refits"] == 0
    # a triangle that becomes invalid cannot be refitted: the commit falls back to the builder, which skips it (scene_triangle_mesh.h:195-215)
    bad = moved[0][0].copy(); bad[base[0][1][17, 0]] = np.nan
    s.update_vertices(gids[0], bad)
    s.commit()
    i2 = s.info()
    assert i2["num_refits"] == 0 and i2["num_triangles"] < i0["num_triangles"]
    ref = [(bad, base[0][1])] + [(moved[k][0] if k == 4 else _wobble(base, 3)[k][0], base[k][1]) for k in range(1, 5)]
    fresh = api.make_scene(dev, ref, flags=flags)
    a, b = make_rayhits(org, tgt - org), make_rayhits(org, tgt - org)
    s.intersect1M(a); fresh.intersect1M(b)
    _same_hits(a, b)
    fresh.release()
    s.release()


def _instanced_scenes(api, dev, g, flags):
    """the scene of tests/golden/ref_instances.npz through the C API: (top, [object scenes])"""
    oa, ob = api.Scene(dev, flags), api.Scene(dev, flags)
    oa.add_triangle_mesh(g["a_v"], g["a_t"]); oa.commit()
    ob.add_triangle_mesh(g["cube_v"], g["cube_t"]); ob.add_quad_mesh(g["qv"], g["qq"]); ob.commit()
    top = api.Scene(dev, flags)
    assert top.add_triangle_mesh(g["ground_v"], g["ground_t"]) == 0 and top.add_triangle_mesh(g["sphere_v"], g["sphere_t"]) == 1
    for i in range(g["xfm"].shape[0]):
        assert top.add_instance(ob if g["inst_obj"][i] else oa, g["xfm"][i], int(g["inst_mask"][i])) == 2 + i
    top.commit()
    return top, [oa, ob]


def _compare_instanced(got, want, rays, label):
    RT = 1e-4
    rel = lambda a, b: np.abs(a - b) <= RT * np.maximum(np.abs(a), np.abs(b)) + 1e-30
    gh, wh = got["geomID"] != INVALID_ID, want["geomID"] != INVALID_ID
    assert (gh == wh).all(), "%s: %d hit/miss disagreements" % (label, (gh != wh).sum())
    same = (got["geomID"] == want["geomID"]) & (got["primID"] == want["primID"]) & (got["instID"] == want["instID"])
    ties = ~same
    assert rel(got["tfar"][ties], want["tfar"][ties]).all(), "%s: different primitive at a different t" % label
    assert ties.sum() <= 40, "%s: %d ties" % (label, ties.sum())
    m = same & wh
    assert (got["instPrimID"][m] == want["instPrimID"][m]).all()
    assert rel(got["tfar"][m], want["tfar"][m]).all()
    ng = np.sqrt(want["Ng_x"][m] ** 2 + want["Ng_y"][m] ** 2 + want["Ng_z"][m] ** 2)
    for f in ("Ng_x", "Ng_y", "Ng_z"):
        assert (np.abs(got[f][m] - want[f][m]) <= RT * ng + 1e-30).all(), f
    assert (np.abs(got["u"][m] - want["u"][m]) <= 1e-4).all() and (np.abs(got["v"][m] - want["v"][m]) <= 1e-4).all()
    miss = ~wh
    assert got[miss].tobytes() == rays[miss].tobytes(), "%s: a missed ray was modified" % label
    exact = sum(int((got[f][m].view(np.uint32) == want[f][m].view(np.uint32)).all()) for f in ("tfar", "u", "v", "Ng_x", "Ng_y", "Ng_z"))
    return int(ties.sum()), exact


@pytest.mark.parametrize("flags", [0, 4])
def test_instances_vs_golden(api, dev, flags):
    """RTC_GEOMETRY_TYPE_INSTANCE (tutorials/instanced_geometry; InstanceIntersector1, kernels/geometry/instance_intersector.cpp): 24 instances of two object
    scenes (triangles; triangles + quads) under rotation / non-uniform scale / shear, geometry masks, next to the scene's own geometry -- against the
    REAL reference's outputs (tests/golden/ref_instances.npz): geomID / primID / instID[0] / instPrimID[0], object-space Ng, u, v, t, occlusion, bounds."""
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_instances.npz"))
    top, objs = _instanced_scenes(api, dev, g, flags)
    suffix = "_robust" if flags else ""
    rays, want = g["rays"], g["hits" + suffix]
    got = rays.copy()
    top.intersect1M(got)
    ties, exact = _compare_instanced(got, want, rays, "instances flags=%d" % flags)
    assert (want["instID"] != INVALID_ID).sum() > 6000
    r = rays_of(rays)
    top.occluded1M(r)
    assert (np.isneginf(r["tfar"]) == np.isneginf(g["occl" + suffix])).all()
    assert (r["tfar"][~np.isneginf(r["tfar"])] == rays["tfar"][~np.isneginf(r["tfar"])]).all()
    lo, hi = top.bounds()
    assert (lo == g["bounds_lo"]).all() and (hi == g["bounds_hi"]).all()
    # packets go through the same kernel: rtcIntersect8 on eight rays that hit instances
    L = api.load()
    from embree_amd.rtypes import RAYHIT_DTYPE
    sel = np.nonzero(want["instID"] != INVALID_ID)[0][:8]
    fields = list(RAYHIT_DTYPE.names[:21])
    raw = np.zeros(21 * 8 + 16, np.uint32)
    ofs = (-raw.ctypes.data % 32) // 4
    pk = raw[ofs:ofs + 21 * 8].reshape(21, 8)
    for fi, f in enumerate(fields):
        pk[fi] = rays[sel][f].view(np.uint32)
    valid = np.full(8, -1, np.int32)
    L.rtcIntersect8(valid.ctypes.data, top.h, pk.ctypes.data, None)
    dev.check()
    for fi, f in enumerate(fields):
        assert (pk[fi] == got[sel][f].view(np.uint32)).all(), "rtcIntersect8 on an instanced scene: " + f
    # the object scenes may be released: the committed top scene holds its own copy of their trees
    for o in objs:
        o.release()
    again = rays.copy()
    top.intersect1M(again)
    assert again.tobytes() == got.tobytes()
    # transform round trip through the three matrix formats (rtcore.cpp:1408-1439)
    geom = L.rtcGetGeometry(top.h, 2)
    out = np.zeros(16, np.float32)
    L.rtcGetGeometryTransform(geom, 0.0, api.RTC_FORMAT_FLOAT3X4_COLUMN_MAJOR, out.ctypes.data)
    assert (out[:12] == g["xfm"][0]).all()
    L.rtcGetGeometryTransform(geom, 0.0, api.RTC_FORMAT_FLOAT3X4_ROW_MAJOR, out.ctypes.data)
    assert (out[:12].reshape(3, 4) == g["xfm"][0].reshape(4, 3).T).all()
    L.rtcGetGeometryTransform(geom, 0.0, api.RTC_FORMAT_FLOAT4X4_COLUMN_MAJOR, out.ctypes.data)
    assert (out.reshape(4, 4)[:, :3] == g["xfm"][0].reshape(4, 3)).all() and (out.reshape(4, 4)[:, 3] == [0, 0, 0, 1]).all()
    top.release()
    print("instances flags=%d: %d ties, %d/6 float fields bit-identical to the reference" % (flags, ties, exact))


def test_instances_edge_cases(api, dev):
    """an instance of an empty scene is skipped; a disabled instance disappears; only instances (no own geometry); a masked-out instance is invisible;
    committing the top scene before the object scene is an error; one object scene instanced 4096 times."""
    L = api.load()
    empty, obj = api.Scene(dev), api.Scene(dev)
    empty.commit()
    v, t = W.triangle_sphere(np.zeros(3, np.float32), 1.0, 6)
    obj.add_triangle_mesh(v, t); obj.commit()
    ident = np.array([1, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0], np.float32)
    top = api.Scene(dev)
    a = top.add_instance(empty, ident)
    x1 = ident.copy(); x1[9] = 5.0
    b = top.add_instance(obj, x1, mask=2)
    top.commit()
    rh = make_rayhits([[5, 0, -4], [0, 0, -4], [5, 0, -4]], [[0, 0, 1]] * 3)
    rh["mask"] = [0xFFFFFFFF, 0xFFFFFFFF, 1]
    top.intersect1M(rh)
    assert rh["instID"][0] == b and rh["geomID"][0] == 0 and abs(rh["tfar"][0] - 3.0) < 1e-3
    assert rh["geomID"][1] == INVALID_ID and rh["geomID"][2] == INVALID_ID
    L.rtcDisableGeometry(L.rtcGetGeometry(top.h, b)); top.commit()
    rh = make_rayhits([[5, 0, -4]], [[0, 0, 1]])
    top.intersect1M(rh)
    assert rh["geomID"][0] == INVALID_ID
    r = rays_of(rh); top.occluded1M(r)
    assert not np.isneginf(r["tfar"]).any()
    top.release()
    # object not committed yet
    late = api.Scene(dev)
    late.add_triangle_mesh(v, t)
    top = api.Scene(dev)
    top.add_instance(late, ident)
    L.rtcCommitScene(top.h)
    assert L.rtcGetDeviceError(dev.h) == 3                      # RTC_ERROR_INVALID_OPERATION
    late.commit(); top.commit()
    rh = make_rayhits([[0, 0, -4]], [[0, 0, 1]]); top.intersect1M(rh)
    assert rh["instID"][0] == 0 and abs(rh["tfar"][0] - 3.0) < 1e-3
    top.release(); late.release()
    # many instances of one object: a 64 x 64 grid, one ray down onto each
    top = api.Scene(dev)
    n = 64
    for i in range(n * n):
        x = ident.copy() * 0.4; x[9], x[10], x[11] = (i % n) * 1.5, 0.0, (i // n) * 1.5
        top.add_instance(obj, x)
    top.commit()
    org = np.stack([(np.arange(n * n) % n) * 1.5, np.full(n * n, 3.0), (np.arange(n * n) // n) * 1.5], -1).astype(np.float32)
    rh = make_rayhits(org, np.tile(np.array([[0.01, -1, 0.02]], np.float32), (n * n, 1)))
    top.intersect1M(rh)
    assert (rh["instID"] == np.arange(n * n)).all() and (np.abs(rh["tfar"] - 2.6) < 0.02).all()
    top.release(); obj.release(); empty.release()


def _sticks(n, seed):
    """long thin diagonal triangles: the case pre-splitting is for (their boxes are mostly empty)"""
    rng = np.random.default_rng(seed)
    a = rng.random((n, 3), dtype=np.float32)
    d = (rng.random((n, 3), dtype=np.float32) - 0.5) * np.float32(1.2)
    w = (rng.random((n, 3), dtype=np.float32) - 0.5) * np.float32(0.02)
    v = np.stack([a, a + d, a + d * 0.5 + w], 1).reshape(-1, 3).astype(np.float32)
    return v, np.arange(3 * n, dtype=np.uint32).reshape(-1, 3)


@pytest.mark.parametrize("flags", [0, 4])
@pytest.mark.parametrize("form", ["spatial", "presplits"])
def test_high_quality_presplit(api, dev, flags, form):
    """rtcSetSceneBuildQuality(HIGH), both forms of the reference: spatial splits inside the recursion (the default: kernels/builders/heuristic_spatial_array.h
    under BVHBuilderBinnedFastSpatialSAH) and, with the device config "presplits=1" (state.cpp:443), the presplit builder (kernels/builders/
    primrefgen_presplit.h + the binned SAH build): big triangles enter the tree as several references with clipped boxes.  Hits do not depend on the tree:
    they must equal the MEDIUM scene's bit for bit (closest hit, any hit, masks); the reference count stays within max_spatial_split_replications; the SAH
    cost drops; rebuilds are bit-identical; refit data is not kept."""
    own = api.Device("gpu=0,presplits=1") if form == "presplits" else None
    if own is not None: dev = own
    # few big triangles among many small ones: the budget (20 % of all references) goes to the big ones (a scene of ONLY big triangles gets no
    # splits at all: every relative priority is below 1, primrefgen_presplit.h:301-305)
    meshes = [_sticks(300, 5), W.triangle_sphere(np.array([0.5, 0.5, 0.5], np.float32), 0.25, 80, noise=0.1, seed=2)]
    ntri = sum(t.shape[0] for _, t in meshes)
    plain = api.Device("gpu=0,top_splits=0")                  # the yardstick: a MEDIUM tree that cuts nothing (by default MEDIUM cuts outlier references itself)
    med = api.make_scene(plain, meshes, masks=[1, 2], flags=flags)
    blobs = []
    for rep in range(2):
        high = api.make_scene(dev, meshes, masks=[1, 2], flags=flags, quality=api.RTC_BUILD_QUALITY_HIGH)
        nodes, tris = high.download_bvh()
        blobs.append(nodes.tobytes() + tris.tobytes())
        if rep == 0:
            ih, im = high.info(), med.info()
            assert im["num_presplit"] == 0 and im["num_triangles"] == ntri
            assert 0 < ih["num_presplit"] <= int(0.2 * ntri) and ih["num_triangles"] == ntri + ih["num_presplit"]
            assert ih["sah"] < 0.8 * im["sah"], (ih["sah"], im["sah"])
            assert ih["bytes_refit"] == 0
            ids = tris["geomID"].astype(np.uint64) << 32 | (tris["primID"] & 0x7FFFFFFF)
            assert np.unique(ids).shape[0] == ntri                  # every triangle is still there; some are named more than once
            org = np.random.default_rng(3).random((80000, 3), dtype=np.float32) * 1.6 - 0.3
            tgt = np.random.default_rng(4).random((80000, 3), dtype=np.float32)
            rays = make_rayhits(org, tgt - org)
            rays["mask"] = np.where(np.arange(80000) % 3 == 0, 1, np.where(np.arange(80000) % 3 == 1, 2, 3)).astype(np.uint32)
            a, b = rays.copy(), rays.copy()
            high.intersect1M(a); med.intersect1M(b)
            assert (b["geomID"] != INVALID_ID).sum() > 20000
            _same_hits(a, b)
            sh, sm = high.trace_stats(api.DeviceArray.from_numpy(rays).ptr, rays.shape[0], 96), med.trace_stats(api.DeviceArray.from_numpy(rays).ptr, rays.shape[0], 96)
            assert sh["nodes"] < sm["nodes"]                       # the point of it: fewer node visits for the same answers
            ra, rb = rays_of(rays), rays_of(rays)
            high.occluded1M(ra); med.occluded1M(rb)
            assert (np.isneginf(ra["tfar"]) == np.isneginf(rb["tfar"])).all()
            print("HIGH (%s) flags=%d: %d + %d references, SAH %.2f -> %.2f, nodes/ray %.2f -> %.2f, tris/ray %.2f -> %.2f" % (form, flags, ntri, ih["num_presplit"], im["sah"], ih["sah"], sm["nodes"] / 8e4, sh["nodes"] / 8e4, sm["tris"] / 8e4, sh["tris"] / 8e4))
        high.release()
    assert blobs[0] == blobs[1]
    med.release(); plain.release()
    # the golden scenes answer the same through a HIGH-quality tree
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_cornell_4k.npz"))
    s = api.make_scene(dev, W.cornell_box(), flags=flags, quality=api.RTC_BUILD_QUALITY_HIGH)
    got = g["rays"].copy()
    s.intersect1M(got)
    same = (got["geomID"] == g["hits"]["geomID"]) & (got["primID"] == g["hits"]["primID"])
    assert same.mean() > 0.995 and np.allclose(got["tfar"][same], g["hits"]["tfar"][same], rtol=1e-4)
    s.release()
    if own is not None: own.release()


def test_interpolate(api, dev):
    """rtcInterpolate / rtcInterpolateN (tutorials/interpolation; InterpolateTrianglesTest verify.cpp:1385): vertex positions and a 5-float vertex attribute
    of a triangle mesh and a quad mesh at hit points; P must be the point the ray hit (org + t dir), dPdu / dPdv the edges, second derivatives 0;
    the N-variant writes SoA and honours the valid mask."""
    L = api.load()
    sv, st = W.triangle_sphere(np.zeros(3, np.float32), 1.0, 12)
    k = 6
    gy, gx = np.meshgrid(np.arange(k + 1, dtype=np.float32), np.arange(k + 1, dtype=np.float32), indexing="ij")
    qv = np.stack([gx / k * 2 + 2, gy / k * 2 - 1, 0.3 * np.sin(gx + gy)], -1).reshape(-1, 3).astype(np.float32)
    ii = (np.arange(k)[:, None] * (k + 1) + np.arange(k)[None, :]).ravel()
    qq = np.stack([ii, ii + 1, ii + k + 2, ii + k + 1], -1).astype(np.uint32)
    s = api.Scene(dev)
    gt, gq = s.add_triangle_mesh(sv, st), s.add_quad_mesh(qv, qq)
    attr = {gt: (np.arange(sv.shape[0] * 5, dtype=np.float32).reshape(-1, 5) * 0.25), gq: (np.cos(np.arange(qv.shape[0] * 5, dtype=np.float32)).reshape(-1, 5))}
    keep = []
    for gid, a in attr.items():
        g = L.rtcGetGeometry(s.h, gid)
        L.rtcSetGeometryVertexAttributeCount(g, 1)
        pad = np.concatenate([a.ravel(), np.zeros(4, np.float32)]); keep.append(pad)
        L.rtcSetSharedGeometryBuffer(g, api.RTC_BUFFER_TYPE_VERTEX_ATTRIBUTE, 0, 0x9005, pad.ctypes.data, 0, 20, a.shape[0])   # RTC_FORMAT_FLOAT5
        L.rtcCommitGeometry(g)
    dev.check()
    s.commit()
    rng = np.random.default_rng(9)
    org = np.stack([rng.uniform(-1, 4, 4000), rng.uniform(-1, 1, 4000), np.full(4000, -5.0)], -1).astype(np.float32)
    rh = make_rayhits(org, np.tile(np.array([[0.01, 0.02, 1]], np.float32), (4000, 1)))
    s.intersect1M(rh)
    hit = np.nonzero(rh["geomID"] != INVALID_ID)[0]
    assert (rh["geomID"][hit] == gt).sum() > 300 and (rh["geomID"][hit] == gq).sum() > 300
    verts = {gt: (sv, st), gq: (qv, qq)}
    for i in hit[:600]:
        gid, pid, u, v = int(rh["geomID"][i]), int(rh["primID"][i]), float(rh["u"][i]), float(rh["v"][i])
        g = L.rtcGetGeometry(s.h, gid)
        P, du, dv, z0, z1, z2 = (np.full(3, 7, np.float32) for _ in range(6))
        a = api.RTCInterpolateArguments(g, pid, u, v, api.RTC_BUFFER_TYPE_VERTEX, 0, P.ctypes.data, du.ctypes.data, dv.ctypes.data, z0.ctypes.data, z1.ctypes.data, z2.ctypes.data, 3)
        L.rtcInterpolate(C.byref(a))
        want = np.array([rh["org_x"][i], rh["org_y"][i], rh["org_z"][i]]) + rh["tfar"][i] * np.array([rh["dir_x"][i], rh["dir_y"][i], rh["dir_z"][i]])
        assert np.abs(P - want).max() < 2e-5, (gid, pid, P, want)
        assert (z0 == 0).all() and (z1 == 0).all() and (z2 == 0).all()
        vv, idx = verts[gid]
        p = vv[idx[pid]]
        if gid == gt or u + v <= 1.0:
            assert np.allclose(du, p[1] - p[0], atol=1e-6) and np.allclose(dv, p[-1 if gid == gq else 2] - p[0], atol=1e-6)
        A = np.zeros(5, np.float32)
        a = api.RTCInterpolateArguments(g, pid, u, v, api.RTC_BUFFER_TYPE_VERTEX_ATTRIBUTE, 0, A.ctypes.data, None, None, None, None, None, 5)
        L.rtcInterpolate(C.byref(a))
        at = attr[gid][idx[pid]]
        if gid == gt:
            ref = (1 - u - v) * at[0] + u * at[1] + v * at[2]
        elif u + v <= 1.0:
            ref = (1 - u - v) * at[0] + u * at[1] + v * at[3]
        else:
            ref = (u + v - 1) * at[2] + (1 - u) * at[3] + (1 - v) * at[1]
        assert np.allclose(A, ref, rtol=1e-5, atol=1e-5)
    dev.check()
    # N-variant on the triangle mesh: SoA outputs, masked entries untouched
    sel = hit[rh["geomID"][hit] == gt][:64]
    n = sel.shape[0]
    valid = np.where(np.arange(n) % 5 == 0, 0, -1).astype(np.int32)
    pids, uu, vv_ = rh["primID"][sel].copy(), rh["u"][sel].copy(), rh["v"][sel].copy()
    P = np.full((3, n), 9, np.float32)
    a = api.RTCInterpolateNArguments(L.rtcGetGeometry(s.h, gt), valid.ctypes.data, pids.ctypes.data, uu.ctypes.data, vv_.ctypes.data, n, api.RTC_BUFFER_TYPE_VERTEX, 0,
                                     P.ctypes.data, None, None, None, None, None, 3)
    L.rtcInterpolateN(C.byref(a))
    dev.check()
    want = np.stack([rh["org_x"][sel] + rh["tfar"][sel] * rh["dir_x"][sel], rh["org_y"][sel] + rh["tfar"][sel] * rh["dir_y"][sel], rh["org_z"][sel] + rh["tfar"][sel] * rh["dir_z"][sel]])
    on = valid != 0
    assert np.abs(P[:, on] - want[:, on]).max() < 2e-5 and (P[:, ~on] == 9).all()
    # errors: bad slot, instance geometry
    a = api.RTCInterpolateArguments(L.rtcGetGeometry(s.h, gt), 0, 0.1, 0.1, api.RTC_BUFFER_TYPE_VERTEX_ATTRIBUTE, 3, P.ctypes.data, None, None, None, None, None, 3)
    L.rtcInterpolate(C.byref(a))
    assert L.rtcGetDeviceError(dev.h) == 2                      # RTC_ERROR_INVALID_ARGUMENT
    s.release()


def test_large_host_arrays_take_the_pipelined_path(api):
    """rtcIntersect1M / rtcOccluded1M on host arrays of >= host_pipeline_min rays: the array is pinned for the call and cut into chunks on two streams.
    Same answers as one device-resident launch (ragged last chunk, 3 chunk sizes, a strided RTCRayHit array), and the array is left unpinned."""
    L = api.load()
    meshes = W.synthetic_crown(num_phi=20)
    rays = W.incoherent_rays(300001, [2, 2, 1.5], seed=4)
    ref_dev = api.Device("host_pipeline_min=4000000000")
    s0 = api.make_scene(ref_dev, meshes)
    want = rays.copy(); s0.intersect1M(want)
    wo = rays_of(rays); s0.occluded1M(wo)
    s0.release(); ref_dev.release()
    for chunk in (1024, 65536, 131072):
        d = api.Device("host_pipeline_min=100000,host_pipeline_chunk=%d" % chunk)
        s = api.make_scene(d, meshes)
        got = rays.copy(); s.intersect1M(got)
        assert got.tobytes() == want.tobytes(), chunk
        go = rays_of(rays); s.occluded1M(go)
        assert go.tobytes() == wo.tobytes(), chunk
        # byteStride 128: records 128 bytes apart inside a larger array (the gaps must come back untouched)
        wide = np.zeros((rays.shape[0], 32), np.uint32)
        wide[:, :24] = rays.view(np.uint32).reshape(-1, 24)
        wide[:, 24:] = 0xABCD1234
        L.rtcIntersect1M(s.h, wide.ctypes.data, rays.shape[0], 128, None)
        d.check()
        assert wide[:, :24].tobytes() == want.tobytes() and (wide[:, 24:] == 0xABCD1234).all()
        got2 = rays.copy(); s.intersect1M(got2)                 # the same array again: it was unpinned, pinning it again must work
        assert got2.tobytes() == want.tobytes()
        s.release(); d.release()


def test_host_device_buffer_calls_and_scene_getters(api, dev):
    """Embree 4.4's host/device buffer API on this library's device copies (rtcore_buffer.h:42-65, rtcore_geometry.h:175-189): rtcNewBufferHostDevice +
    rtcCommitBuffer + rtcGetBufferDataDevice, rtcSetNewGeometryBufferHostDevice, rtcGetGeometryBufferDataDevice; rtcGetGeometryUserDataFromScene and
    rtcGetGeometryTransformFromScene; entry points outside the path record RTC_ERROR_INVALID_OPERATION instead of failing to link."""
    L = api.load()
    vp = C.c_void_p
    L.rtcNewBufferHostDevice.restype = vp; L.rtcNewBufferHostDevice.argtypes = [vp, C.c_size_t]
    L.rtcGetBufferData.restype = vp; L.rtcGetBufferData.argtypes = [vp]
    L.rtcGetBufferDataDevice.restype = vp; L.rtcGetBufferDataDevice.argtypes = [vp]
    L.rtcCommitBuffer.argtypes = [vp]; L.rtcReleaseBuffer.argtypes = [vp]
    L.rtcSetGeometryBuffer.argtypes = [vp, C.c_int, C.c_uint, C.c_int, vp, C.c_size_t, C.c_size_t, C.c_size_t]
    L.rtcSetNewGeometryBufferHostDevice.argtypes = [vp, C.c_int, C.c_uint, C.c_int, C.c_size_t, C.c_size_t, C.POINTER(vp), C.POINTER(vp)]
    L.rtcGetGeometryBufferDataDevice.restype = vp; L.rtcGetGeometryBufferDataDevice.argtypes = [vp, C.c_int, C.c_uint]
    L.rtcGetGeometryUserDataFromScene.restype = vp; L.rtcGetGeometryUserDataFromScene.argtypes = [vp, C.c_uint]
    L.rtcGetGeometryTransformFromScene.argtypes = [vp, C.c_uint, C.c_float, C.c_int, vp]
    L.rtcSetGeometryUserData.argtypes = [vp, vp]
    v, t = W.triangle_sphere(np.zeros(3, np.float32), 1.0, 8)
    # vertex buffer: an RTCBuffer created host/device, filled on the host, committed, bound to the geometry
    vb = L.rtcNewBufferHostDevice(dev.h, v.nbytes + 16)
    C.memmove(L.rtcGetBufferData(vb), v.ctypes.data, v.nbytes)
    L.rtcCommitBuffer(vb)
    dptr = L.rtcGetBufferDataDevice(vb)
    assert dptr
    back = np.zeros_like(v)
    assert L.mi355_memcpy_d2h(back.ctypes.data, dptr, v.nbytes) == 0 and (back == v).all()
    g = L.rtcNewGeometry(dev.h, api.RTC_GEOMETRY_TYPE_TRIANGLE)
    L.rtcSetGeometryBuffer(g, api.RTC_BUFFER_TYPE_VERTEX, 0, api.RTC_FORMAT_FLOAT3, vb, 0, 12, v.shape[0])
    L.rtcReleaseBuffer(vb)
    # index buffer: host and device pointer handed out together
    hp, dp = vp(), vp()
    L.rtcSetNewGeometryBufferHostDevice(g, api.RTC_BUFFER_TYPE_INDEX, 0, api.RTC_FORMAT_UINT3, 12, t.shape[0], C.byref(hp), C.byref(dp))
    dev.check()
    assert hp.value and dp.value
    C.memmove(hp.value, t.ctypes.data, t.nbytes)
    L.rtcSetGeometryUserData(g, 0x1234)
    L.rtcCommitGeometry(g)
    assert L.rtcGetGeometryBufferDataDevice(g, api.RTC_BUFFER_TYPE_INDEX, 0) == dp.value
    tb = np.zeros_like(t)
    assert L.mi355_memcpy_d2h(tb.ctypes.data, dp.value, t.nbytes) == 0 and (tb == t).all()      # rtcCommitGeometry brought the indices over
    s = api.Scene(dev)
    gid = L.rtcAttachGeometry(s.h, g)
    L.rtcReleaseGeometry(g)
    obj = api.Scene(dev); obj.add_triangle_mesh(v, t); obj.commit()
    x = np.array([2, 0, 0, 0, 2, 0, 0, 0, 2, 5, 0, 0], np.float32)
    iid = s.add_instance(obj, x)
    s.commit()
    assert L.rtcGetGeometryUserDataFromScene(s.h, gid) == 0x1234
    out = np.zeros(12, np.float32)
    L.rtcGetGeometryTransformFromScene(s.h, iid, 0.0, api.RTC_FORMAT_FLOAT3X4_COLUMN_MAJOR, out.ctypes.data)
    assert (out == x).all()
    rh = make_rayhits([[0, 0, -4], [5, 0, -4]], [[0, 0, 1]] * 2)
    s.intersect1M(rh)
    assert rh["geomID"][0] == gid and rh["instID"][0] == INVALID_ID and abs(rh["tfar"][0] - 3.0) < 1e-3
    assert rh["instID"][1] == iid and abs(rh["tfar"][1] - 2.0) < 1e-3
    dev.check()
    assert L.rtcGetDeviceError(None) == 0                    # nothing stale in the calling thread's device-less slot (tests/conftest.py checks this after every test)
    for name in ("rtcForwardIntersect1", "rtcSetGeometryTransformQuaternion", "rtcNewBVH", "rtcPointQuery4"):
        getattr(L, name).restype = vp
        getattr(L, name)(None, None, None, None)
        assert L.rtcGetDeviceError(None) == 3, name              # RTC_ERROR_INVALID_OPERATION, recorded for the calling thread (no device in these calls)
    s.release(); obj.release()
