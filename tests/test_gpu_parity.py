"""GPU parity tests (run on the MI355X box: pytest -m gpu).  Everything goes through the C ABI of
libembree4_mi355.so (include/embree4/rtcore.h + include/embree_amd_hip.h); the oracle (oracle/restate.c,
oracle/_ref) is only the checker.

Modelled on the reference's own tests (tutorials/verify/verify.cpp): TriangleHitTest :2462, RayMasksTest :2626,
InactiveRaysTest :3553, NaNTest :3813 / InfTest :3884, EmptySceneTest :1054, GetBoundsTest :785,
BufferStrideTest :915, EnableDisableGeometryTest, SmallTriangleHitTest :3692, WatertightTest :3611.
Bar: IDs bit-exact (exact-t ties classified), t / Ng within 1e-4 relative (tests/helpers.py).
"""
import ctypes as C
import os

import numpy as np
import pytest

from embree_amd import workloads as W
from embree_amd.rtypes import make_rayhits, rays_of, INVALID_ID, RAYHIT_DTYPE, RAY_DTYPE
from tests import bvh_check
from tests.helpers import compare_closest, compare_occluded, RTOL

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


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


@pytest.fixture(scope="module")
def restate():
    from oracle import restate as R
    assert R.available(), "oracle/librestate.so missing (make -C oracle)"
    return R


def oracle_scene(R, meshes, masks=None):
    o = R.OracleScene()
    for i, (v, t) in enumerate(meshes):
        o.add_mesh(v, t, 1 if masks is None else masks[i])
    o.commit()
    return o


def soup(n, seed, size=0.08):
    rng = np.random.default_rng(seed)
    c = rng.random((n, 3), dtype=np.float32)
    v = (c[:, None, :] + (rng.random((n, 3, 3), dtype=np.float32) - 0.5) * size).reshape(-1, 3)
    return v.astype(np.float32), np.arange(3 * n, dtype=np.uint32).reshape(-1, 3)


# ------------------------------------------------------------------------------------------- golden vectors
@pytest.mark.parametrize("name,meshes,masks", [("ref_cube_1k.npz", W.cube_and_plane, None),
                                               ("ref_cornell_4k.npz", W.cornell_box, None)])
def test_golden_reference_vectors(api, dev, restate, golden_dir, name, meshes, masks):
    """HIP path vs outputs of the REAL reference committed under tests/golden/."""
    g = np.load(os.path.join(golden_dir, name))
    m = meshes()
    s = api.make_scene(dev, m, masks)
    o = oracle_scene(restate, m, masks)
    got = g["rays"].copy()
    s.intersect1M(got)
    compare_closest(got, g["hits"], g["rays"], o.triangle_t, max_tie_frac=0.02, label=name)
    r = rays_of(g["rays"])
    s.occluded1M(r)
    compare_occluded(r["tfar"], g["occluded_tfar"], rays_of(g["rays"])["tfar"], label=name)
    lo, hi = s.bounds()                                   # GetBoundsTest
    assert (lo == g["bounds_lo"]).all() and (hi == g["bounds_hi"]).all()
    s.release()


def test_golden_soup_with_masks(api, dev, restate, golden_dir):
    g = np.load(os.path.join(golden_dir, "ref_soup_8k.npz"))
    m = [(g["v0"], g["t0"]), (g["v1"], g["t1"])]
    s = api.make_scene(dev, m, [1, 2])
    o = oracle_scene(restate, m, [1, 2])
    got = g["rays"].copy()
    s.intersect1M(got)
    compare_closest(got, g["hits"], g["rays"], o.triangle_t, label="soup masks")
    r = rays_of(g["rays"])
    s.occluded1M(r)
    compare_occluded(r["tfar"], g["occluded_tfar"], rays_of(g["rays"])["tfar"], label="soup masks")
    s.release()


def test_triangle_hit_known_answer(api, dev, golden_dir):
    """TriangleHitTest: |u-u0|, |v-v0|, |t-1| <= 16 ulp, Ng == (0,0,1) +- 16 ulp, IDs 0/0."""
    g = np.load(os.path.join(golden_dir, "ref_trianglehit.npz"))
    tv = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0]], np.float32)
    s = api.make_scene(dev, [(tv, np.array([[0, 1, 2]], np.uint32))])
    rh = g["rays"].copy()
    s.intersect1M(rh)
    ulp = np.finfo(np.float32).eps
    assert (rh["geomID"] == 0).all() and (rh["primID"] == 0).all()
    assert (np.abs(rh["u"] - g["u0"]) <= 16 * ulp).all() and (np.abs(rh["v"] - g["v0"]) <= 16 * ulp).all()
    assert (np.abs(rh["tfar"] - 1.0) <= 16 * ulp).all()
    assert (np.abs(rh["Ng_x"]) <= 16 * ulp).all() and (np.abs(rh["Ng_y"]) <= 16 * ulp).all() and (np.abs(rh["Ng_z"] - 1) <= 16 * ulp).all()
    assert (rh["instID"] == INVALID_ID).all()
    # hit point consistency: org + t*dir == v0 + u*(v1-v0) + v*(v2-v0)
    P = np.stack([rh["org_x"] + rh["tfar"] * rh["dir_x"], rh["org_y"] + rh["tfar"] * rh["dir_y"], rh["org_z"] + rh["tfar"] * rh["dir_z"]], -1)
    Q = np.stack([rh["u"], rh["v"], np.zeros_like(rh["u"])], -1)
    assert np.abs(P - Q).max() < 1e-5
    s.release()


# --------------------------------------------------------------------------------- oracle parity + BVH checks
@pytest.mark.parametrize("n,seed", [(1, 1), (5, 2), (64, 3), (1000, 4), (1025, 5), (30000, 6)])
def test_soup_parity_and_tree(api, dev, restate, n, seed):
    m = [soup(n, seed)]
    s = api.make_scene(dev, m)
    info = s.info()
    nodes, tris = s.download_bvh()
    bvh_check.validate(nodes, tris, info["root_ref"], m, max_leaf=info["max_leaf"])
    o = oracle_scene(restate, m)
    rays = W.incoherent_rays(20000, [0.5, 0.5, 0.5], seed=seed)
    want, got = rays.copy(), rays.copy()
    o.intersect1(want)
    s.intersect1M(got)
    compare_closest(got, want, rays, o.triangle_t, label=f"soup{n}")
    wr, gr = rays_of(rays), rays_of(rays)
    o.occluded1(wr)
    s.occluded1M(gr)
    compare_occluded(gr["tfar"], wr["tfar"], rays_of(rays)["tfar"], label=f"soup{n}")
    s.release()


def test_degenerate_and_duplicate_geometry(api, dev, restate):
    """OverlappingGeometryTest / GarbageGeometryTest spirit: all centroids identical (median fallback split),
    zero-area triangles kept, out-of-range / NaN / huge vertices skipped."""
    base = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0]], np.float32)
    v = np.tile(base, (3000, 1))
    t = np.arange(9000, dtype=np.uint32).reshape(-1, 3)
    bad_v = np.array([[np.nan, 0, 0], [3e18, 0, 0], [2, 2, 2], [2, 2, 2], [2, 2, 2]], np.float32)   # last three: zero-area
    v2 = np.concatenate([base + np.float32(5), bad_v])
    t2 = np.array([[0, 1, 2], [0, 1, 3], [0, 1, 4], [5, 6, 7], [0, 1, 99]], np.uint32)
    m = [(v, t), (v2, t2)]
    s = api.make_scene(dev, m)
    info = s.info()
    assert info["num_triangles"] == 3000 + 2
    nodes, tris = s.download_bvh()
    bvh_check.validate(nodes, tris, info["root_ref"], m, max_leaf=info["max_leaf"])
    o = oracle_scene(restate, m)
    rays = make_rayhits([[0.2, 0.2, -1], [5.2, 5.2, 4], [9, 9, 9]], [[0, 0, 1], [0, 0, 1], [0, 0, 1]])
    want, got = rays.copy(), rays.copy()
    o.intersect1(want)
    s.intersect1M(got)
    # 3000 coincident triangles: every one of them is an exact tie
    compare_closest(got, want, rays, o.triangle_t, max_tie_frac=1.0, label="duplicates")
    assert got["geomID"][0] == 0 and got["geomID"][1] == 1 and got["primID"][1] == 0 and got["geomID"][2] == INVALID_ID
    s.release()


def test_crown_small_bounce_and_shadow(api, dev, restate):
    meshes = W.synthetic_crown(num_phi=20)
    s = api.make_scene(dev, meshes)
    info = s.info()
    nodes, tris = s.download_bvh()
    bvh_check.validate(nodes, tris, info["root_ref"], meshes, max_leaf=info["max_leaf"], allow_splits=True)   # 72,972 triangles: the first levels may cut some (top splits)
    o = oracle_scene(restate, meshes)
    prim = W.crown_camera_rays(meshes, 160, 160)
    want, got = prim.copy(), prim.copy()
    o.intersect1(want)
    s.intersect1M(got)
    compare_closest(got, want, prim, o.triangle_t, label="crown primary")
    bounce = W.diffuse_bounce_rays(want, meshes)
    w2, g2 = bounce.copy(), bounce.copy()
    o.intersect1(w2)
    s.intersect1M(g2)
    compare_closest(g2, w2, bounce, o.triangle_t, label="crown bounce")
    sh = W.shadow_rays(w2[:4096], meshes, samples=4)
    ws, gs = sh.copy(), sh.copy()
    o.occluded1(ws)
    s.occluded1M(gs)
    compare_occluded(gs["tfar"], ws["tfar"], sh["tfar"], label="crown shadow")
    # device-resident geometry (rtcSetSharedGeometryBufferHostDevice) and library-owned buffers give the same tree
    for kw in (dict(device_resident=True), dict(shared=False)):
        s2 = api.make_scene(dev, meshes, **kw)
        g3 = bounce.copy()
        s2.intersect1M(g3)
        assert g3.tobytes() == g2.tobytes()
        s2.release()
    s.release()


# ----------------------------------------------------------------------------------------- API semantics
def test_single_ray_and_packet_entry_points(api, dev, restate):
    """rtcIntersect1/4/8/16 + rtcOccluded1/4/8/16 on host pointers; InactiveRaysTest: lanes with valid != -1 untouched."""
    m = W.cube_and_plane()
    s = api.make_scene(dev, m)
    o = oracle_scene(restate, m)
    rays = W.cube_camera_rays(8, 8)
    want = rays.copy()
    o.intersect1(want)
    one = rays[27:28].copy()
    s.intersect1(one)
    assert one.tobytes() == _trace_batch(s, rays)[27:28].tobytes()
    r1 = rays_of(rays[27:28])
    s.occluded1(r1)
    assert np.isneginf(r1["tfar"][0]) == (want["geomID"][27] != INVALID_ID)
    L = api.load()
    for K in (4, 8, 16):
        fields = list(RAYHIT_DTYPE.names[:21])
        pk = np.zeros((21, K), np.uint32)
        sel = rays[10:10 + K]
        for fi, f in enumerate(fields):
            pk[fi] = sel[f].view(np.uint32)
        valid = np.full(K, -1, np.int32)
        valid[1] = 0                                          # inactive lane
        before = pk.copy()
        buf = _aligned(pk)
        getattr(L, "rtcIntersect%d" % K)(valid.ctypes.data, s.h, buf.ctypes.data, None)
        dev.check()
        ref = _trace_batch(s, rays)[10:10 + K]
        for fi, f in enumerate(fields):
            col = buf[fi]
            assert col[1] == before[fi][1], f"inactive lane modified ({f})"
            act = np.arange(K) != 1
            hit = ref["geomID"][act] != INVALID_ID
            if f in ("tfar", "Ng_x", "Ng_y", "Ng_z", "u", "v", "primID", "geomID"):
                assert (col[act][hit] == ref[f].view(np.uint32)[act][hit]).all(), f
        rk = np.zeros((12, K), np.uint32)
        for fi, f in enumerate(fields[:12]):
            rk[fi] = sel[f].view(np.uint32)
        rbuf = _aligned(rk)
        getattr(L, "rtcOccluded%d" % K)(valid.ctypes.data, s.h, rbuf.ctypes.data, None)
        dev.check()
        tf = rbuf[8].view(np.float32)
        assert tf[1] == sel["tfar"][1]
        act = np.arange(K) != 1
        assert (np.isneginf(tf[act]) == (ref["geomID"][act] != INVALID_ID)).all()
    s.release()


def _aligned(a, align=64):
    raw = np.zeros(a.nbytes + align, np.uint8)
    off = (-raw.ctypes.data) % align
    out = raw[off:off + a.nbytes].view(a.dtype).reshape(a.shape)
    out[...] = a
    return out


def _trace_batch(s, rays):
    r = rays.copy()
    s.intersect1M(r)
    return r


def test_errors_and_state_machine(api, dev):
    """error codes never cross as exceptions; first error wins and is cleared on read (rtcGetDeviceError)."""
    L = api.load()
    s = api.Scene(dev)
    rh = make_rayhits([[0, 0, -1]], [[0, 0, 1]])
    L.rtcIntersect1(s.h, rh.ctypes.data, None)               # not committed: missing_rtcCommit -> INVALID_OPERATION
    assert dev.get_error() == api.RTC_ERROR_INVALID_OPERATION
    assert dev.get_error() == api.RTC_ERROR_NONE             # cleared on read
    g = L.rtcNewGeometry(dev.h, 2)                            # RTC_GEOMETRY_TYPE_GRID: a feature outside the triangle / quad path
    assert not g and dev.get_error() == api.RTC_ERROR_INVALID_OPERATION
    g = L.rtcNewGeometry(dev.h, api.RTC_GEOMETRY_TYPE_TRIANGLE)
    v = np.zeros(16, np.float32)
    L.rtcSetSharedGeometryBuffer(g, api.RTC_BUFFER_TYPE_VERTEX, 0, api.RTC_FORMAT_UINT3, v.ctypes.data, 0, 12, 3)   # wrong format
    assert dev.get_error() == api.RTC_ERROR_INVALID_OPERATION
    L.rtcSetSharedGeometryBuffer(g, api.RTC_BUFFER_TYPE_VERTEX, 0, api.RTC_FORMAT_FLOAT3, v.ctypes.data + 2, 0, 12, 3)  # misaligned
    assert dev.get_error() == api.RTC_ERROR_INVALID_OPERATION
    L.rtcSetSharedGeometryBuffer(g, api.RTC_BUFFER_TYPE_VERTEX, 1, api.RTC_FORMAT_FLOAT3, v.ctypes.data, 0, 12, 3)  # slot
    assert dev.get_error() == api.RTC_ERROR_INVALID_ARGUMENT
    L.rtcSetGeometryIntersectFilterFunction(g, None)          # filter callbacks are accepted (host functions, run between launches: test_filter_callbacks_vs_reference)
    assert dev.get_error() == api.RTC_ERROR_NONE
    L.rtcReleaseGeometry(g)
    # two errors: the first one is kept
    L.rtcNewGeometry(dev.h, 2)
    L.rtcSetSharedGeometryBuffer(None, 0, 0, 0, None, 0, 0, 0)
    assert dev.get_error() == api.RTC_ERROR_INVALID_OPERATION
    # the NULL-handle call has no device: its error lands in the calling thread's device-less slot (device.cpp:273-279), first error wins, cleared on read
    assert L.rtcGetDeviceError(None) == api.RTC_ERROR_INVALID_ARGUMENT
    assert L.rtcGetDeviceError(None) == api.RTC_ERROR_NONE
    s.release()
    assert L.rtcGetErrorString(3) == b"Invalid operation"


def test_empty_scene_masks_enable_detach(api, dev, restate):
    L = api.load()
    s = api.Scene(dev)
    s.commit()                                               # EmptySceneTest
    rays = W.incoherent_rays(100, [0, 0, 0])
    got = rays.copy()
    s.intersect1M(got)
    assert got.tobytes() == rays.tobytes()
    r = rays_of(rays)
    s.occluded1M(r)
    assert (r["tfar"] == rays["tfar"]).all()
    lo, hi = s.bounds()
    assert np.isposinf(lo).all() and np.isneginf(hi).all()
    # two planes at z=1 (geom 0, mask 1) and z=2 (geom 1, mask 2)
    def plane(z):
        return (np.array([[-1, -1, z], [1, -1, z], [1, 1, z], [-1, 1, z]], np.float32), np.array([[0, 1, 2], [0, 2, 3]], np.uint32))
    g0 = s.add_triangle_mesh(*plane(1), mask=1)
    g1 = s.add_triangle_mesh(*plane(2), mask=2)
    assert (g0, g1) == (0, 1)
    s.commit()
    rays = make_rayhits([[0.1, 0.2, 0]] * 4, [[0, 0, 1]] * 4)
    rays["mask"] = [1, 2, 3, 4]
    got = rays.copy()
    s.intersect1M(got)
    assert list(got["geomID"]) == [0, 1, 0, INVALID_ID] and list(got["tfar"][:3]) == [1.0, 2.0, 1.0]   # RayMasksTest
    # disable geometry 0 (EnableDisableGeometryTest), then detach it: lowest free ID is reused
    L.rtcDisableGeometry(L.rtcGetGeometry(s.h, 0))
    s.commit()
    got = rays.copy()
    got["mask"] = 0xFFFFFFFF
    s.intersect1M(got)
    assert (got["geomID"] == 1).all()
    L.rtcDetachGeometry(s.h, 0)
    g2 = s.add_triangle_mesh(*plane(0.5))
    assert g2 == 0
    s.commit()
    got = rays.copy()
    got["mask"] = 0xFFFFFFFF
    s.intersect1M(got)
    assert (got["geomID"] == 0).all() and (got["tfar"] == 0.5).all()
    s.release()


def test_nan_inf_rays_and_strides(api, dev):
    """NaNTest / InfTest: invalid rays must neither hang nor crash; BufferStrideTest: strided vertex/index/ray records."""
    v, t = soup(5000, 9)
    vs = np.zeros((v.shape[0], 5), np.float32)
    vs[:, :3] = v
    ts = np.zeros((t.shape[0], 4), np.uint32)
    ts[:, :3] = t
    L = api.load()
    s = api.Scene(dev)
    g = L.rtcNewGeometry(dev.h, api.RTC_GEOMETRY_TYPE_TRIANGLE)
    L.rtcSetSharedGeometryBuffer(g, api.RTC_BUFFER_TYPE_VERTEX, 0, api.RTC_FORMAT_FLOAT3, vs.ctypes.data, 0, 20, v.shape[0])
    L.rtcSetSharedGeometryBuffer(g, api.RTC_BUFFER_TYPE_INDEX, 0, api.RTC_FORMAT_UINT3, ts.ctypes.data, 0, 16, t.shape[0])
    L.rtcCommitGeometry(g)
    L.rtcAttachGeometry(s.h, g)
    L.rtcReleaseGeometry(g)
    s.commit()
    s_ref = api.make_scene(dev, [(v, t)])
    rays = W.incoherent_rays(4096, [0.5, 0.5, 0.5], seed=2)
    a, b = rays.copy(), rays.copy()
    s.intersect1M(a)
    s_ref.intersect1M(b)
    assert a.tobytes() == b.tobytes()
    # strided ray records (byteStride 128) through rtcIntersect1M
    wide = np.zeros((rays.shape[0], 128), np.uint8)
    wide[:, :96] = rays.view(np.uint8).reshape(-1, 96)
    wide[:, 96:] = 0xAB
    L.rtcIntersect1M(s.h, wide.ctypes.data, rays.shape[0], 128, None)
    dev.check()
    assert wide[:, :96].tobytes() == b.tobytes() and (wide[:, 96:] == 0xAB).all()
    bad = rays.copy()
    bad["dir_x"][::3] = np.nan
    bad["org_y"][1::3] = np.inf
    bad["tfar"][2::7] = np.nan
    bad["dir_z"][5::11] = 0.0
    s.intersect1M(bad)                                        # must return
    r = rays_of(bad)
    s.occluded1M(r)
    L.rtcIntersect1M(s.h, None, 0, 96, None)                  # M == 0 is a no-op
    dev.check()
    s.release()
    s_ref.release()


def test_rebuild_is_deterministic(api, dev):
    meshes = W.synthetic_crown(num_phi=12)
    rays = W.incoherent_rays(30000, [2, 2, 1.5], seed=4)
    outs, infos = [], []
    for _ in range(3):
        s = api.make_scene(dev, meshes)
        r = rays.copy()
        s.intersect1M(r)
        outs.append(r.tobytes())
        i = s.info()
        infos.append((i["num_nodes"], i["num_leaves"], i["num_binary_nodes"], i["depth"]))
        s.release()
    assert outs[0] == outs[1] == outs[2] and infos[0] == infos[1] == infos[2]


def test_tree_layout_is_bit_identical_across_rebuilds(api, dev):
    """No atomic counter decides an index (implicit binary numbering, scanned wide numbering): the CNode and TriRec arrays of two
    commits of the same scene are the same bytes -- what makes 'every rank builds its own replica'  exact."""
    meshes = W.synthetic_crown(num_phi=40)                    # ~310k triangles: several top-phase levels, thousands of small sub-trees
    blobs = []
    for _ in range(3):
        s = api.make_scene(dev, meshes)
        nodes, tris = s.download_bvh()
        blobs.append((nodes.tobytes(), tris.tobytes(), s.info()["sah"]))
        s.release()
    assert blobs[0] == blobs[1] == blobs[2]


def test_queries_are_thread_safe_and_streams_independent(api, dev, restate):
    """rtcIntersect1M / rtcOccluded1M from several host threads on one committed scene (doc/src/api/rtcIntersect1.md: thread safe), and
    rtcIntersect1MDevice on four streams at once (each stream has its own traversal scratch): every call returns the single-threaded answer."""
    import threading
    meshes = W.synthetic_crown(num_phi=20)
    s = api.make_scene(dev, meshes)
    L = api.load()
    rays = [W.incoherent_rays(30000, [2, 2, 1.5], seed=20 + k) for k in range(6)]
    want = []
    for r in rays:
        w = r.copy()
        s.intersect1M(w)
        want.append(w)
    got = [r.copy() for r in rays]
    occ = [rays_of(r) for r in rays]
    errs = []

    def work(k):
        try:
            for _ in range(3):
                g = rays[k].copy()
                s.intersect1M(g)
                assert g.tobytes() == want[k].tobytes()
            s.intersect1M(got[k])
            s.occluded1M(occ[k])
        except Exception as e:            # noqa: BLE001
            errs.append(repr(e))
    th = [threading.Thread(target=work, args=(k,)) for k in range(6)]
    [t.start() for t in th]
    [t.join() for t in th]
    assert not errs, errs
    for k in range(6):
        assert got[k].tobytes() == want[k].tobytes()
        assert (np.isneginf(occ[k]["tfar"]) == (want[k]["geomID"] != INVALID_ID)).all()
    # device-pointer entry point on four streams, launches interleaved
    streams, bufs = [], []
    for k in range(4):
        st = C.c_void_p()
        L.mi355_stream_create(0, C.byref(st))
        streams.append(st)
        bufs.append(api.DeviceArray.from_numpy(rays[k]))
    for rep in range(3):
        for k in range(4):
            s.intersect1M_device(bufs[k].ptr, rays[k].shape[0], 96, streams[k])
    L.mi355_device_synchronize(0)
    for k in range(4):      # tracing a traced buffer again must not change it (tfar already holds the hit distance: same hit, inclusive at tfar)
        assert bufs[k].download(RAYHIT_DTYPE).tobytes() == want[k].tobytes()
        bufs[k].free()
        L.mi355_stream_destroy(streams[k])
    s.release()


@pytest.mark.parametrize("quality", [None, 0])
def test_degenerate_inputs_at_scale(api, dev, quality):
    """Inputs that defeat the heuristics must neither hang nor lose triangles: 300,000 coincident triangles (every SAH split invalid -> median
    splits all the way down; equal Morton codes), plus two far outliers that stretch the scene bounds by 10^6, plus zero-area triangles."""
    base = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0]], np.float32)
    n = 300000
    v = np.tile(base, (n, 1))
    t = np.arange(3 * n, dtype=np.uint32).reshape(-1, 3)
    far = np.array([[1e6, 1e6, 1e6], [1e6 + 1, 1e6, 1e6], [1e6, 1e6 + 1, 1e6], [-1e6, 0, 0], [-1e6, 1, 0], [-1e6, 0, 1], [5, 5, 5], [5, 5, 5], [5, 5, 5]], np.float32)
    m = [(v, t), (far, np.arange(9, dtype=np.uint32).reshape(-1, 3))]
    s = api.make_scene(dev, m, quality=quality)
    info = s.info()
    assert info["num_triangles"] == n + 3 and info["depth"] < 64
    rays = make_rayhits(np.array([[0.25, 0.25, -1], [1e6 + 0.25, 1e6 + 0.25, 1e6 - 1], [-1e6, 0.25, 0.25 - 1 + 1], [9, 9, 9]], np.float32),
                        np.array([[0, 0, 1], [0, 0, 1], [0, 0, 1], [0, 0, 1]], np.float32))
    rays["org_x"][2], rays["org_y"][2], rays["org_z"][2] = -1e6 - 1, 0.25, 0.25
    rays["dir_x"][2], rays["dir_y"][2], rays["dir_z"][2] = 1, 0, 0
    s.intersect1M(rays)
    assert rays["geomID"][0] == 0 and rays["tfar"][0] == 1.0          # one of the 300,000 (which one is an exact tie)
    assert rays["geomID"][1] == 1 and rays["primID"][1] == 0
    assert rays["geomID"][2] == 1 and rays["primID"][2] == 1
    assert rays["geomID"][3] == INVALID_ID
    s.release()


# ---------------------------------------------------------------------- RTC_GEOMETRY_TYPE_QUAD (first half)
def noisy_quad_grid(k, seed, z=0.5, amp=0.05):
    gy, gx = np.meshgrid(np.arange(k + 1, dtype=np.float32), np.arange(k + 1, dtype=np.float32), indexing="ij")
    rng = np.random.default_rng(seed)
    v = np.stack([gx / k, gy / k, z + amp * rng.standard_normal(gx.shape).astype(np.float32)], -1).reshape(-1, 3).astype(np.float32)
    i = (np.arange(k)[:, None] * (k + 1) + np.arange(k)[None, :]).ravel()
    return v, np.stack([i, i + 1, i + k + 2, i + k + 1], -1).astype(np.uint32)      # non-planar quads


@pytest.mark.parametrize("robust", [False, True])
def test_quads_mixed_scene_vs_golden_and_oracle(api, dev, restate, golden_dir, robust):
    """Quad meshes next to triangle meshes: primID = quad index, u/v/Ng of the quad (second half flipped like the reference's AVX quad
    intersectors), against the REAL reference's outputs (tests/golden/ref_quads.npz) and the restatement; invalid quads are dropped whole."""
    g = np.load(os.path.join(golden_dir, "ref_quads.npz"))
    tv, tt, qv, qq = g["tv"], g["tt"], g["qv"], g["qq"]
    flags = api.RTC_SCENE_FLAG_ROBUST if robust else 0
    s = api.Scene(dev, flags)
    o = restate.OracleScene(robust=robust)
    assert s.add_triangle_mesh(tv, tt) == 0 and o.add_mesh(tv, tt) == 0
    assert s.add_quad_mesh(qv, qq, mask=3) == 1 and o.add_quads(qv, qq, 3) == 1
    s.commit()
    o.commit()
    rays = g["rays"]
    want = g["hits_robust" if robust else "hits"]
    got = rays.copy()
    s.intersect1M(got)
    st = compare_closest(got, want, rays, o.triangle_t, label="quads golden robust=%s" % robust)
    assert (want["geomID"] == 1).sum() > 1000 and st["ties"] < 50
    r = rays_of(rays)
    s.occluded1M(r)
    compare_occluded(r["tfar"], g["occl_robust" if robust else "occl"], rays_of(rays)["tfar"], label="quads occluded")
    lo, hi = s.bounds()
    assert (lo == g["bounds_lo"]).all() and (hi == g["bounds_hi"]).all()
    info = s.info()
    assert info["num_triangles"] == tt.shape[0] + 2 * qq.shape[0]
    s.release()
    # a quad with an out-of-range index or a non-finite vertex disappears as a whole (QuadMesh::buildBounds)
    bad_v = np.concatenate([qv, np.array([[np.nan, 0, 0]], np.float32)])
    bad_q = np.concatenate([qq[:50], np.array([[0, 1, 2, 9999999], [0, 1, qv.shape[0], 3]], np.uint32)])
    s2 = api.Scene(dev, flags)
    s2.add_quad_mesh(bad_v, bad_q)
    s2.commit()
    assert s2.info()["num_triangles"] == 100
    s2.release()


# ---------------------------------------------------------------------- RTC_BUILD_QUALITY_LOW: Morton build
@pytest.mark.parametrize("n,seed", [(1, 1), (2, 7), (5, 2), (64, 3), (1000, 4), (30000, 6)])
def test_low_quality_build_tree_and_parity(api, dev, restate, n, seed):
    """rtcSetSceneBuildQuality(LOW) = Morton-code build: the tree must satisfy every structural invariant of the SAH tree (same node / leaf
    layout) and the hits must be the reference's (they do not depend on the tree)."""
    m = [soup(n, seed)]
    s = api.make_scene(dev, m, quality=api.RTC_BUILD_QUALITY_LOW)
    info = s.info()
    nodes, tris = s.download_bvh()
    bvh_check.validate(nodes, tris, info["root_ref"], m, max_leaf=info["max_leaf"])
    o = oracle_scene(restate, m)
    rays = W.incoherent_rays(4096, [0.5, 0.5, 0.5], seed=seed)
    want, got = rays.copy(), rays.copy()
    o.intersect1(want)
    s.intersect1M(got)
    compare_closest(got, want, rays, o.triangle_t, label="morton soup %d" % n)
    s.release()


def test_low_quality_crown_duplicates_and_determinism(api, dev, restate):
    """Morton build on a multi-geometry scene with thousands of coincident centroids (equal codes: ties broken by index), bit-identical
    across rebuilds, robust flag combined with it, and a sane SAH (the quality knob trades build time for SAH, not correctness)."""
    meshes = W.synthetic_crown(num_phi=24)
    base = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0]], np.float32) * 0.01 + np.float32(1.0)
    dup = (np.tile(base, (2000, 1)), np.arange(6000, dtype=np.uint32).reshape(-1, 3))       # 2000 identical triangles = 2000 equal Morton codes
    meshes = meshes + [dup]
    blobs = []
    for _ in range(2):
        s = api.make_scene(dev, meshes, quality=api.RTC_BUILD_QUALITY_LOW)
        nodes, tris = s.download_bvh()
        blobs.append((nodes.tobytes(), tris.tobytes()))
        info_low = s.info()
        if _ == 0:
            bvh_check.validate(nodes, tris, info_low["root_ref"], meshes, max_leaf=info_low["max_leaf"])
            o = oracle_scene(restate, meshes)
            rays = W.incoherent_rays(40000, [2, 2, 1.5], seed=11)
            want, got = rays.copy(), rays.copy()
            o.intersect1(want)
            s.intersect1M(got)
            compare_closest(got, want, rays, o.triangle_t, label="morton crown")
        s.release()
    assert blobs[0] == blobs[1]
    med = api.make_scene(dev, meshes)
    assert info_low["sah"] < 2.5 * med.info()["sah"], (info_low["sah"], med.info()["sah"])
    med.release()
    r = api.make_scene(dev, meshes, flags=api.RTC_SCENE_FLAG_ROBUST, quality=api.RTC_BUILD_QUALITY_LOW)
    orb = restate.OracleScene(robust=True)
    for v, t in meshes:
        orb.add_mesh(v, t)
    orb.commit()
    rays = W.incoherent_rays(20000, [2, 2, 1.5], seed=12)
    want, got = rays.copy(), rays.copy()
    orb.intersect1(want)
    r.intersect1M(got)
    compare_closest(got, want, rays, orb.triangle_t, label="morton + robust")
    r.release()


# ---------------------------------------------------------------------- RTC_SCENE_FLAG_ROBUST
WATERTIGHT_POS = np.array([148376.0, 1234.0, -223423.0], np.float32)     # verify.cpp:6575-6621


def test_robust_golden_watertight(api, dev, restate, golden_dir):
    """Robust scenes against the REAL reference's robust outputs (tests/golden/ref_watertight_robust.npz): WatertightTest's sphere
    200 km from the origin, rays from inside.  IDs bit-exact (tie rule), t/u/v/Ng within tolerance, every ray hits."""
    g = np.load(os.path.join(golden_dir, "ref_watertight_robust.npz"))
    sph = W.triangle_sphere(WATERTIGHT_POS, 2.0, 50)
    s = api.make_scene(dev, [sph], flags=api.RTC_SCENE_FLAG_ROBUST)
    o = restate.OracleScene(robust=True)
    o.add_mesh(*sph)
    o.commit()
    got = g["rays"].copy()
    s.intersect1M(got)
    assert (got["geomID"] == 0).all(), "a ray leaked through the closed sphere in robust mode"
    compare_closest(got, g["hits"], g["rays"], o.triangle_t, label="robust golden")
    r = rays_of(g["rays"])
    s.occluded1M(r)
    compare_occluded(r["tfar"], g["occluded_tfar"], rays_of(g["rays"])["tfar"], label="robust golden occluded")
    nodes, tris = s.download_bvh()
    # robust leaves keep the vertices themselves (TriangleMv), not v0/e1/e2
    v, t = sph
    key = {int(p): i for i, p in enumerate(tris["primID"])}
    for p in (0, 17, t.shape[0] - 1):
        rec = tris[key[p]]
        assert (rec["v0"] == v[t[p, 0]]).all() and (rec["e1"] == v[t[p, 1]]).all() and (rec["e2"] == v[t[p, 2]]).all()
    s.release()


@pytest.mark.parametrize("model", ["sphere", "plane"])
def test_robust_watertight_fail_rate(api, dev, model):
    """WatertightTest (verify.cpp:3611-3688) at full intensity: numPhi / grid 200, rays as the test builds them; the reference accepts a
    failure rate of 2e-5, closest hit and occlusion."""
    rng = np.random.default_rng(5)
    n = 400000
    if model == "sphere":
        mesh = W.triangle_sphere(WATERTIGHT_POS, 2.0, 200)
        org = (WATERTIGHT_POS[None, :] + (2.0 * rng.random((n, 3), dtype=np.float32) - 1.0)).astype(np.float32)
        dirs = (2.0 * rng.random((n, 3), dtype=np.float32) - 1.0).astype(np.float32)
    else:   # createTrianglePlane(p0 = (pos.x,-6,-6), dx = (0,0,12), dy = (0,12,0), 200 x 200)
        k = 200
        gy, gx = np.meshgrid(np.arange(k + 1, dtype=np.float32), np.arange(k + 1, dtype=np.float32), indexing="ij")
        verts = np.stack([np.full_like(gx, WATERTIGHT_POS[0]), -6.0 + 12.0 * gy / k, -6.0 + 12.0 * gx / k], -1).reshape(-1, 3).astype(np.float32)
        i = (np.arange(k)[:, None] * (k + 1) + np.arange(k)[None, :]).ravel()
        tris = np.concatenate([np.stack([i, i + 1, i + k + 1], -1), np.stack([i + 1, i + k + 2, i + k + 1], -1)]).astype(np.uint32)
        mesh = (verts, tris)
        org = np.tile(np.array([[WATERTIGHT_POS[0] - 3.0, 0.0, 0.0]], np.float32), (n, 1))
        dirs = (2.0 * rng.random((n, 3), dtype=np.float32) - 1.0).astype(np.float32)
        dirs[:, 0] = 1.0
    s = api.make_scene(dev, [mesh], flags=api.RTC_SCENE_FLAG_ROBUST)
    rh = make_rayhits(org, dirs)
    s.intersect1M(rh)
    fail = float((rh["geomID"] == INVALID_ID).mean())
    r = rays_of(make_rayhits(org, dirs))
    s.occluded1M(r)
    fail_o = float((~np.isneginf(r["tfar"])).mean())
    s.release()
    assert fail <= 2e-5 and fail_o <= 2e-5, (model, fail, fail_o)


def test_robust_parity_crown_vs_oracle(api, dev, restate):
    """Robust traversal on a multi-geometry scene against the robust restatement: incoherent closest hit + occlusion + packets."""
    meshes = W.synthetic_crown(num_phi=24)
    rays = W.incoherent_rays(60000, [2, 2, 1.5], seed=9)
    s = api.make_scene(dev, meshes, flags=api.RTC_SCENE_FLAG_ROBUST)
    o = restate.OracleScene(robust=True)
    for v, t in meshes:
        o.add_mesh(v, t)
    o.commit()
    want = rays.copy()
    o.intersect1(want)
    got = rays.copy()
    s.intersect1M(got)
    st = compare_closest(got, want, rays, o.triangle_t, label="robust crown")
    assert st["hits"] > 0.9 * rays.shape[0]
    wr, gr = rays_of(rays), rays_of(rays)
    o.occluded1(wr)
    s.occluded1M(gr)
    compare_occluded(gr["tfar"], wr["tfar"], rays_of(rays)["tfar"], label="robust crown occluded")
    # the fast scene of the same geometry must still give the fast answers (flag is per scene)
    f = api.make_scene(dev, meshes)
    of = oracle_scene(restate, meshes)
    wf, gf = rays.copy(), rays.copy()
    of.intersect1(wf)
    f.intersect1M(gf)
    compare_closest(gf, wf, rays, of.triangle_t, label="fast scene next to a robust one")
    f.release()
    s.release()


# ---------------------------------------------------------------------- BASELINE.json full sizes: properties
@pytest.fixture(scope="module")
def crown_full(api, dev):
    meshes = W.synthetic_crown()                              # 4,762,764 triangles
    s = api.make_scene(dev, meshes, device_resident=True)
    prim = W.crown_camera_rays(meshes, 1024, 1024)
    tr = prim.copy()
    s.intersect1M(tr)
    bounce = W.diffuse_bounce_rays(tr, meshes)
    yield meshes, s, bounce
    s.release()


def check_properties(s, meshes, rays, min_hit=0.0):
    """Size-independent properties of a closest-hit / any-hit pass that need no oracle (used at BASELINE.json's full sizes)."""
    got = rays.copy()
    s.intersect1M(got)
    hit = got["geomID"] != INVALID_ID
    assert hit.mean() >= min_hit
    # (1) the reported triangle really is hit at the reported t: recompute Moeller-Trumbore in float64 on the host
    idx = np.nonzero(hit)[0][:: max(1, hit.sum() // 200000)]
    gid, pid = got["geomID"][idx], got["primID"][idx]
    tri = np.zeros((idx.size, 3, 3), np.float64)
    for g in np.unique(gid):
        m = gid == g
        v, t = meshes[g]
        tri[m] = v[t[pid[m]]].astype(np.float64)
    O = np.stack([rays["org_x"][idx], rays["org_y"][idx], rays["org_z"][idx]], -1).astype(np.float64)
    D = np.stack([rays["dir_x"][idx], rays["dir_y"][idx], rays["dir_z"][idx]], -1).astype(np.float64)
    e1, e2 = tri[:, 1] - tri[:, 0], tri[:, 2] - tri[:, 0]
    Ng = np.cross(e1, e2)
    tt = ((tri[:, 0] - O) * Ng).sum(-1) / (D * Ng).sum(-1)
    assert np.abs(tt - got["tfar"][idx]).max() <= 1e-4 * np.abs(tt).max()
    NgGot = np.stack([got["Ng_x"][idx], got["Ng_y"][idx], got["Ng_z"][idx]], -1)
    assert np.abs(NgGot - Ng).max() <= 1e-4 * np.abs(Ng).max() + 1e-12
    P = O + tt[:, None] * D
    # u, v: the hit point rebuilt from them, v0 + u (v1 - v0) + v (v2 - v0), is the point on the ray (a spatial tolerance:
    # u along a 1e-3 long edge of a 40 unit long pipe triangle is ill-conditioned in fp32 for the reference as well)
    Q = tri[:, 0] + got["u"][idx].astype(np.float64)[:, None] * e1 + got["v"][idx].astype(np.float64)[:, None] * e2
    assert (np.sqrt(((Q - P) ** 2).sum(-1)) <= 1e-4 * (np.abs(tt) + np.sqrt((e1 * e1).sum(-1)) + np.sqrt((e2 * e2).sum(-1)))).all()
    assert (got["u"][idx] >= 0).all() and (got["v"][idx] >= 0).all() and (got["u"][idx] + got["v"][idx] <= 1 + 1e-5).all()
    # (2) idempotence: tracing the result again (tfar = hit distance, inclusive) changes nothing
    again = got.copy()
    s.intersect1M(again)
    same = (again["primID"] == got["primID"]) & (again["geomID"] == got["geomID"])
    assert same.mean() > 0.9999 and np.abs(again["tfar"][hit] - got["tfar"][hit]).max() <= RTOL * np.abs(got["tfar"][hit]).max()
    # (3) closest-hit / any-hit consistency: a ray is occluded iff it has a closest hit
    r = rays_of(rays)
    s.occluded1M(r)
    assert (np.isneginf(r["tfar"]) == hit).mean() > 0.99999
    # (4) shortening the ray to just before its hit removes the hit
    short = rays.copy()
    short["tfar"] = np.where(hit, got["tfar"] * np.float32(1 - 1e-3), rays["tfar"])
    s.intersect1M(short)
    closer = (short["geomID"] != INVALID_ID) & hit
    assert closer.mean() < 1e-4
    # (5) scaling the direction by 2 halves t and keeps IDs (linearity of the parametrisation)
    sc = rays.copy()
    for f in ("dir_x", "dir_y", "dir_z"):
        sc[f] *= np.float32(2)
    sc["tnear"] *= np.float32(0.5)
    sc["tfar"] *= np.float32(0.5)
    s.intersect1M(sc)
    both = hit & (sc["geomID"] != INVALID_ID)
    ids_same = (sc["primID"][both] == got["primID"][both]) & (sc["geomID"][both] == got["geomID"][both])
    assert ids_same.mean() > 0.9995
    assert np.abs(2 * sc["tfar"][both][ids_same] - got["tfar"][both][ids_same]).max() <= 4 * RTOL * got["tfar"][both].max()
    return got


def test_full_size_properties(api, dev, crown_full):
    """configs[2] at full size (2^20 incoherent rays, 4.76M triangles): properties that need no oracle."""
    meshes, s, rays = crown_full
    assert s.info()["num_triangles"] - s.info()["num_presplit"] == W.num_triangles(meshes)    # (a MEDIUM build may cut triangles of its first levels: num_presplit extra leaf records)
    check_properties(s, meshes, rays, min_hit=0.99)        # closed room: (almost) every bounce ray hits something


def test_cornell_full_size_primary(api, dev, restate):
    """configs[1]: Cornell box (34 triangles), 1024 x 1024 coherent primary rays, closest hit, against the oracle."""
    m = W.cornell_box()
    s = api.make_scene(dev, m)
    o = oracle_scene(restate, m)
    rays = W.cornell_camera_rays(1024, 1024)
    want, got = rays.copy(), rays.copy()
    o.intersect1(want)
    s.intersect1M(got)
    st = compare_closest(got, want, rays, o.triangle_t, max_tie_frac=0.01, label="cornell 1M primary")   # wall seams are exact ties
    assert st["rays"] == 1 << 20 and st["hits"] > 0.5 * st["rays"]
    s.release()


def test_shadow_shard_of_16M(api, dev, crown_full):
    """configs[3]: 16 Mi shadow rays sharded over 8 GPUs -> every rank traces a contiguous 2 Mi range with rtcOccluded1M.
    One rank's range here; occlusion must agree with a closest-hit query on the same segment (any-hit == closest-hit exists)."""
    from embree_amd import shard
    meshes, s, bounce = crown_full
    lo, hi = shard.shard_range(16 << 20, 3, 8)
    assert hi - lo == 2 << 20
    traced = bounce[: (hi - lo) // 16].copy()
    s.intersect1M(traced)
    sh = W.shadow_rays(traced, meshes, samples=16)             # 2 Mi RTCRay records: this rank's shard
    assert 0.99 * (hi - lo) <= sh.shape[0] <= hi - lo         # closed room: (almost) every bounce ray has a hit point to shade
    occ = sh.copy()
    s.occluded1M(occ)
    full = np.zeros(sh.shape[0], RAYHIT_DTYPE)
    for f in sh.dtype.names:
        full[f] = sh[f]
    full["geomID"] = INVALID_ID; full["primID"] = INVALID_ID; full["instID"] = INVALID_ID
    s.intersect1M(full)
    assert (np.isneginf(occ["tfar"]) == (full["geomID"] != INVALID_ID)).all()
    keep = ~np.isneginf(occ["tfar"])
    assert (occ["tfar"][keep] == sh["tfar"][keep]).all()       # unoccluded rays untouched


def test_powerplant_parity_small(api, dev, restate):
    """configs[4] at a size the oracle finishes in seconds: long thin pipe triangles + boxes, tree check + closest/any parity."""
    m = W.synthetic_powerplant(target_tris=60000)
    s = api.make_scene(dev, m)
    info = s.info()
    nodes, tris = s.download_bvh()
    bvh_check.validate(nodes, tris, info["root_ref"], m, max_leaf=info["max_leaf"])
    o = oracle_scene(restate, m)
    lo, hi = W.scene_bounds(m)
    rays = W.incoherent_rays(50000, (lo + hi) / 2, seed=9)
    want, got = rays.copy(), rays.copy()
    o.intersect1(want)
    s.intersect1M(got)
    compare_closest(got, want, rays, o.triangle_t, label="powerplant 60k")
    wr, gr = rays_of(rays), rays_of(rays)
    o.occluded1(wr)
    s.occluded1M(gr)
    compare_occluded(gr["tfar"], wr["tfar"], rays_of(rays)["tfar"], label="powerplant 60k")
    s.release()


def test_powerplant_full_size(api, dev):
    """configs[4] at full size: 12.7M-triangle GPU SAH build + 2^20 incoherent rays, properties that need no oracle."""
    m = W.synthetic_powerplant()
    s = api.make_scene(dev, m, device_resident=True)
    info = s.info()
    assert info["num_triangles"] - info["num_presplit"] == W.num_triangles(m) == 12699996
    assert info["bytes_nodes"] == 80 * info["num_nodes"] and info["depth"] < 64
    lo, hi = W.scene_bounds(m)
    blo, bhi = s.bounds()
    assert (blo == lo).all() and (bhi == hi).all()
    rays = W.incoherent_rays(1 << 20, (lo + hi) / 2, seed=11)
    check_properties(s, m, rays, min_hit=0.3)
    s.release()


def test_full_size_vs_real_reference(api, dev, crown_full):
    """configs[2] at full size against the REAL reference (oracle/_ref travels with the repo snapshot)."""
    from oracle import refembree
    if not refembree.available():
        pytest.skip("oracle/_ref not present on this box")
    meshes, s, rays = crown_full
    R = refembree.RefScene("threads=%d" % min(16, refembree.hw_threads()))
    for v, t in meshes:
        R.add_mesh(v, t)
    R.commit()
    want, got = rays.copy(), rays.copy()
    R.intersect1(want, refembree.hw_threads())
    s.intersect1M(got)
    from oracle import restate
    o = restate.OracleScene()                                 # only for triangle_t (no tree needed for that)
    for v, t in meshes:
        o.add_mesh(v, t)
    st = compare_closest(got, want, rays, o.triangle_t, label="crown full vs reference")
    assert st["ties"] <= 16, st                               # measured: 2 of 2^20
    assert st["hits"] > 0.99 * st["rays"]
    sh = W.shadow_rays(want[: 1 << 16], meshes, samples=16)   # 2^20 shadow rays (config 4 per-GPU shard size is 2^21)
    ws, gs = sh.copy(), sh.copy()
    R.occluded1(ws, refembree.hw_threads())
    s.occluded1M(gs)
    compare_occluded(gs["tfar"], ws["tfar"], sh["tfar"], max_flip_frac=1e-5, label="crown shadow vs reference")
    R.close()
