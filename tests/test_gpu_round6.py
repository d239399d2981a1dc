"""Round 6 GPU tests (pytest -m gpu): the boundary items of VERDICT r05 / ADVICE r05 -- overlapping index elements through a COMMIT, the property table -- and the
gates of the round's kernel work (small-batch launches give the bytes of large-batch launches; the commit's trees are bit-identical across rebuilds).  All through the
C ABI; the checker is the REAL reference (oracle/_ref)."""
import ctypes as C
import hashlib
import os
import subprocess
import sys

import numpy as np
import pytest

from embree_amd import workloads as W
from embree_amd.rtypes import rays_of, INVALID_ID, RAYHIT_DTYPE, RAY_DTYPE, make_rayhits
from tests.helpers import compare_closest, compare_occluded

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def api():
    from embree_amd import api as a
    return a


@pytest.fixture(scope="module")
def dev(api):
    d = api.Device("gpu=0")
    yield d
    d.release()


@pytest.fixture(scope="module")
def ref():
    from oracle import refembree
    assert refembree.available(), "oracle/_ref is missing on the GPU box: make -f oracle/ref.mk must have run in the build container before the snapshot was taken"
    return refembree


# ------------------------------------------------------------------------------------------- overlapping index elements through rtcCommitScene
@pytest.mark.parametrize("flags", [0, 4])                       # fast, RTC_SCENE_FLAG_ROBUST
def test_quad_indices_with_12_byte_stride_commit_and_trace_vs_reference(api, dev, ref, flags):
    """A UINT4 quad index view with a 12-byte stride -- consecutive quads share one word -- is what the reference's BufferStrideTest binds
    (tutorials/verify/verify.cpp:995-1008), and the reference reads quad i at offset + i * stride whatever the stride (kernels/common/buffer.h BufferView::operator[]).
    Round 5 accepted the view in rtcSetSharedGeometryBuffer and refused it in rtcCommitScene; verify never commits that scene, so nothing showed.  Here the scene IS
    committed and traced: a strip of non-planar quads whose quad i is the words [3 i, 3 i + 4) of ONE flat array, against the real reference given the same quads written out
    with the usual 16-byte stride."""
    rng = np.random.default_rng(12)
    nq = 300                                                      # a strip of nq quads stacked in y; quad i = (w[3i], w[3i+1], w[3i+2], w[3i+3]) of ONE word array, every word a vertex of its own:
    words = np.arange(3 * nq + 1, dtype=np.uint32)                # ... vertex 3i = (0, y_i), 3i+1 = (1, y_i), 3i+2 = (1, y_i + h), 3i+3 = (0, y_i + h) = the first corner of quad i + 1
    h = 1.0 / nq
    v = np.zeros((3 * nq + 1, 3), np.float32)
    i = np.arange(nq)
    v[3 * i] = np.stack([np.zeros(nq), i * h, np.zeros(nq)], -1)
    v[3 * i + 1] = np.stack([np.ones(nq), i * h, np.zeros(nq)], -1)
    v[3 * i + 2] = np.stack([np.ones(nq), (i + 1) * h, np.zeros(nq)], -1)
    v[3 * nq] = (0.0, 1.0, 0.0)
    v[:, 2] = 0.05 * rng.random(3 * nq + 1, dtype=np.float32)     # non-planar quads, cracks along the seams (every quad has corners of its own)
    quads = np.stack([words[0:3 * nq:3], words[1:3 * nq:3], words[2:3 * nq:3], words[3:3 * nq + 1:3]], -1).astype(np.uint32)   # the same quads, element by element
    assert (quads[1:, 0] == quads[:-1, 3]).all()                  # (the shared word)
    s = api.Scene(dev, flags)
    assert s.add_quad_mesh(v, quads, index_words=words, index_stride=12) == 0
    s.commit()
    dev.check()                                                    # no error from the commit
    assert s.info()["num_triangles"] == 2 * nq
    R = ref.RefScene("threads=4", flags=flags)
    R.add_quads(v, quads)
    R.commit()
    assert R.error() == 0
    n = 20000
    org = np.stack([rng.random(n, dtype=np.float32), rng.random(n, dtype=np.float32), np.full(n, 1.0, np.float32)], -1)
    tgt = np.stack([rng.random(n, dtype=np.float32), rng.random(n, dtype=np.float32), np.zeros(n, np.float32)], -1)
    rays = make_rayhits(org, tgt - org)
    want, got = rays.copy(), rays.copy()
    R.intersect1(want, threads=4)
    s.intersect1M(got)

    def quad_t(rr, geom, prim):                                   # t of a named quad (either half), fp64: the tie classifier of tests/helpers.py
        out = np.zeros(rr.shape[0], np.float32)
        for j in range(rr.shape[0]):
            q = quads[int(prim[j])]
            o = np.array([rr["org_x"][j], rr["org_y"][j], rr["org_z"][j]], np.float64)
            d = np.array([rr["dir_x"][j], rr["dir_y"][j], rr["dir_z"][j]], np.float64)
            best = np.inf
            for a, b, c in ((q[0], q[1], q[3]), (q[2], q[3], q[1])):
                A, B, Cc = v[a].astype(np.float64), v[b].astype(np.float64), v[c].astype(np.float64)
                nrm = np.cross(B - A, Cc - A)
                den = np.dot(nrm, d)
                if den != 0.0:
                    t = np.dot(nrm, A - o) / den
                    if abs(t - rr["tfar"][j]) < abs(best - rr["tfar"][j]):
                        best = t
            out[j] = best
        return out
    st = compare_closest(got, want, rays, quad_t, max_tie_frac=0.01, label="12-byte-stride quads vs reference (flags %d)" % flags)
    assert st["hits"] > 0.3 * n
    wr, gr = rays_of(rays), rays_of(rays)
    R.occluded1(wr, threads=4)
    s.occluded1M(gr)
    compare_occluded(gr["tfar"], wr["tfar"], rays_of(rays)["tfar"], label="12-byte-stride quads, occlusion")
    R.close()
    s.release()


def test_property_table_tells_what_is_built(api, dev):
    """rtcGetDeviceProperty against what the library implements (reference table: kernels/common/device.cpp:480-600).  FILTER_FUNCTION_SUPPORTED answered
    0 for three rounds although geometry and argument filters run on every entry point, which hid the reference's intersection_filter test group."""
    L = api.load()
    get = lambda p: L.rtcGetDeviceProperty(dev.h, p)
    assert get(66) == 1                                            # RTC_DEVICE_PROPERTY_FILTER_FUNCTION_SUPPORTED
    assert get(129) == 1                                           # RTC_DEVICE_PROPERTY_JOIN_COMMIT_SUPPORTED (rtcJoinCommitScene commits under the scene's lock)
    assert get(96) == 1 and get(97) == 1 and get(64) == 1          # triangles, quads, ray masks
    assert get(32) == 1 and get(33) == 1 and get(34) == 1          # rtcIntersect4/8/16 are entry points of their own
    for p in (62, 63, 65, 67, 68, 98, 99, 100, 101, 128, 130, 140, 141):   # off in the reference's default build, or not built here
        assert get(p) == 0, p
    assert get(142) == 1                                           # RTC_DEVICE_PROPERTY_HIP_DEVICE (extension)
    dev.check()


# ------------------------------------------------------------------------------------------- the launch shape must not show in the answers
@pytest.mark.parametrize("any_hit", [False, True])
def test_small_batches_give_the_bytes_of_the_full_batch(api, dev, any_hit):
    """A batch that fits the lane slots of the resident grid takes the STATIC launch shape (trace.hip: R rays per wave, no cursor, no atomics, the lanes without a ray help
    from the first iteration on); larger ones the persistent grid with the cursor hand-out.  Closest hit = the minimum over all accepted candidates of (t bits, triangle),
    occlusion = any accepted candidate: neither depends on which lane or wave traced a ray, so the first n rays of a batch, launched alone, must come back as the first n
    records of the whole batch's launch -- byte for byte, for every n on either side of every shape boundary (16 / 32 rays per wave, 2^16 rays, ragged last waves, one ray)."""
    L = api.load()
    meshes = W.synthetic_crown(num_phi=48)
    s = api.make_scene(dev, meshes)
    prim = W.crown_camera_rays(meshes, 512, 512)
    s.intersect1M(prim)
    rays = W.diffuse_bounce_rays(prim, meshes, seed=3)            # 2^18 incoherent rays: more than the static shape takes
    src = rays_of(rays) if any_hit else rays
    dt, rec = (RAY_DTYPE, 48) if any_hit else (RAYHIT_DTYPE, 96)
    query = s.occluded1M_device if any_hit else s.intersect1M_device
    d = api.DeviceArray.from_numpy(src)
    query(d.ptr, src.shape[0])
    L.mi355_device_synchronize(0)
    full = d.download(dt)
    assert (np.isneginf(full["tfar"]).mean() > 0.3) if any_hit else ((full["geomID"] != INVALID_ID).mean() > 0.9)
    for n in (1, 15, 16, 17, 1000, 4096, 32768, 32769, 40001, 65536, 65537, 131072, 200000):
        L.mi355_memcpy_h2d(d.ptr, src.ctypes.data, n * rec)
        query(d.ptr, n)
        L.mi355_device_synchronize(0)
        assert s.trace_status() == 0
        assert d.download(dt, n).tobytes() == full[:n].tobytes(), "the first %d rays launched alone differ from their records in the full batch" % n
    d.free()
    s.release()


def test_counting_kernel_reports_distinct_nodes_and_triangles(api, dev):
    """mi355_trace_stats out[18] / out[19]: the DISTINCT nodes / triangle records a launch fetches (one bit per record, set by the counting kernel) -- what bench.py's
    roofline.compulsory_bytes is made of.  Bounds that must hold: 0 < distinct <= visits, distinct <= what the tree has; a batch traced twice touches the same set; one
    ray touches exactly as many distinct nodes as it visits."""
    meshes = W.synthetic_crown(num_phi=32)
    s = api.make_scene(dev, meshes)
    info = s.info()
    prim = W.crown_camera_rays(meshes, 128, 128)
    d = api.DeviceArray.from_numpy(prim)
    a = s.trace_stats(d.ptr, prim.shape[0], 96)
    assert 0 < a["unique_nodes"] <= min(a["nodes"], info["num_nodes"]) and 0 < a["unique_tris"] <= min(a["tris"], info["num_triangles"])
    api.load().mi355_memcpy_h2d(d.ptr, prim.ctypes.data, prim.nbytes)
    b = s.trace_stats(d.ptr, prim.shape[0], 96)
    assert (a["unique_nodes"], a["unique_tris"], a["nodes"], a["tris"]) == (b["unique_nodes"], b["unique_tris"], b["nodes"], b["tris"])
    api.load().mi355_memcpy_h2d(d.ptr, prim.ctypes.data, 96)
    one = s.trace_stats(d.ptr, 1, 96)
    assert one["unique_nodes"] == one["nodes"] and one["unique_tris"] == one["tris"] and one["nodes"] >= 1
    d.free()
    s.release()


def test_commit_is_bit_identical_across_rebuilds_and_qualities_keep_their_hashes(api, dev):
    """The round's builder changes (the area statistics of the outlier cut taken by primref_gen, four items per thread in primref_gen / tri_records, DPP block scans) move no
    byte of any tree: MEDIUM and HIGH commits of a scene WITH outliers (a room around noisy spheres) repeated four times give identical node and triangle arrays, and the SAH
    the builder reports is the one of round 5 (305.88 for the bench's scene is checked by bench.py's line; here: a smaller crown)."""
    meshes = W.synthetic_crown(num_phi=40)
    for quality in (None, api.RTC_BUILD_QUALITY_HIGH):
        s = api.make_scene(dev, meshes, quality=quality)
        n0, t0 = s.download_bvh()
        sah0 = s.info()["sah"]
        assert s.info()["num_presplit"] > 0 or quality is None
        for _ in range(3):
            s.touch()
            s.commit()
            n1, t1 = s.download_bvh()
            assert n1.tobytes() == n0.tobytes() and t1.tobytes() == t0.tobytes() and s.info()["sah"] == sah0
        s.release()


def test_outlier_cut_together_with_invalid_triangles_vs_reference(api, dev, ref):
    """Round 6, last session: the front end of a MEDIUM commit no longer reads the references three more times -- outlier_mark counts the valid references of every 256-tile for
    the compaction (outlier_emit takes the cut ones off, compact_count covers only the tile N falls into and the reserve) and measures the centroid box of what stays
    (outlier_clip adds the pieces').  The case that needs all of it at once: a scene WITH outliers (the room of the crown stand-in) AND invalid triangles -- NaN vertices in
    the first tile, in the middle, and in the very tile the room's triangles sit in.  Answers against the real reference (which skips the same triangles), the reference count,
    and bit-identical rebuilds; the same scene without the cut (top_splits=0) must give the same hits."""
    meshes = [(v.copy(), t.copy()) for v, t in W.synthetic_crown(num_phi=40)]
    ntri = W.num_triangles(meshes)
    rng = np.random.default_rng(77)
    bad = 0
    for gi in (0, len(meshes) // 2, len(meshes) - 1):              # the last geometry is the room (the outliers)
        v, t = meshes[gi]
        extra = np.array([[np.nan, 0, 0], [3e18, 1, 1]], np.float32)
        base = v.shape[0]
        v2 = np.concatenate([v, extra])
        t2 = t.copy()
        pick = [0, t.shape[0] // 2] if t.shape[0] > 16 else [0]
        for k, j in enumerate(pick):
            t2[j, k % 3] = base + (k & 1); bad += 1                 # one corner becomes NaN / huge: the triangle is skipped (like the reference's isvalid)
        meshes[gi] = (v2, t2)
    plain = api.Device("gpu=0,top_splits=0")
    s = api.make_scene(dev, meshes)
    p = api.make_scene(plain, meshes)
    info, pinfo = s.info(), p.info()
    assert pinfo["num_presplit"] == 0 and pinfo["num_triangles"] == ntri - bad
    assert info["num_presplit"] > 0 and info["num_triangles"] - info["num_presplit"] == ntri - bad, (info["num_triangles"], info["num_presplit"], ntri, bad)
    R = ref.RefScene("threads=4")
    for v, t in meshes:
        R.add_mesh(v, t)
    R.commit()
    assert R.error() == 0
    from tests.test_gpu_round4 import tri_t_of
    tri_t = tri_t_of(meshes)
    prim = W.crown_camera_rays(meshes, 192, 192)
    want, got, gotp = prim.copy(), prim.copy(), prim.copy()
    R.intersect1(want)
    s.intersect1M(got)
    p.intersect1M(gotp)
    compare_closest(got, want, prim, tri_t, label="outliers + invalid, primary")
    compare_closest(got, gotp, prim, tri_t, label="cut vs uncut tree, primary")   # the cut changes the tree, not the answers (an exact-t tie may name the other triangle)
    bounce = W.diffuse_bounce_rays(want, meshes)
    want, got, gotp = bounce.copy(), bounce.copy(), bounce.copy()
    R.intersect1(want)
    s.intersect1M(got)
    p.intersect1M(gotp)
    compare_closest(got, want, bounce, tri_t, label="outliers + invalid, bounce")
    compare_closest(got, gotp, bounce, tri_t, label="cut vs uncut tree, bounce")
    n0, t0 = s.download_bvh()
    for _ in range(3):                                             # (the second commit of the kind also takes the learned node-buffer path: tri_records copies the nodes out)
        s.touch()
        s.commit()
        n1, t1 = s.download_bvh()
        assert n1.tobytes() == n0.tobytes() and t1.tobytes() == t0.tobytes() and s.info()["num_nodes"] == info["num_nodes"]
    s.release(); p.release()


@pytest.mark.parametrize("instanced", [False, True])
def test_host_arrays_cross_the_link_packed_and_leave_misses_alone(api, instanced):
    """Round 6, last session: a host-array query sends 48 bytes per ray up and only the fields the query writes down (packed_link, the default; rtcore_api.cpp
    staged_query_packed) -- the caller's records are written by the CPU, and only those of the rays that HIT: a miss must leave the hit fields exactly as the caller passed
    them (the reference's rtcIntersect1 never touches the hit of a ray that misses; kernels/geometry/intersector_epilog.h:235-300 writes on a hit only), whatever they hold.
    Records with garbage in their hit fields, rays that hit, miss, are inactive (tnear > tfar) or NaN: byte-identical to the whole-record path (packed_link=0), chunked
    (a batch above host_pipeline_min) and unchunked, closest hit and occlusion, with and without instances (the 48-byte form of the results)."""
    from embree_amd.rtypes import make_rayhits
    rng = np.random.default_rng(31)
    n = 300000                                                     # (> host_pipeline_min: three chunks, the last one ragged)
    org = np.stack([rng.uniform(-2.0, 14.0, n), rng.uniform(-2.0, 2.0, n), np.full(n, -5.0)], -1).astype(np.float32)
    dirs = np.tile(np.array([0, 0, 1], np.float32), (n, 1)); dirs[:, :2] += rng.normal(0.0, 0.05, (n, 2)).astype(np.float32)
    rays = make_rayhits(org, dirs)
    rays["tnear"][::97] = 50.0; rays["tfar"][::97] = 10.0           # inactive
    rays["dir_x"][5::211] = np.nan                                  # NaN rays: no hit, nothing written
    for f, v in (("Ng_x", 7.0), ("Ng_y", -3.0), ("Ng_z", 1e30), ("u", 9.0), ("v", 11.0)): rays[f] = np.float32(v)
    rays["primID"] = 12345; rays["geomID"] = 777; rays["instID"] = 55          # what a careless caller leaves in the hit part
    out = {}
    for cfg in ("gpu=0,packed_link=0", "gpu=0"):
        dev = api.Device(cfg)
        obj = api.make_scene(dev, [W.triangle_sphere(np.zeros(3, np.float32), 1.0, 24)])
        if instanced:
            top = api.Scene(dev)
            for k in range(5):
                top.add_instance(obj, np.array([1, 0, 0, 0, 1, 0, 0, 0, 1, 3.0 * k, 0, 0], np.float32))
            top.commit()
        else:
            top = obj
        a = rays.copy(); top.intersect1M(a)
        b = rays[:5000].copy(); top.intersect1M(b)                 # (one chunk)
        c = rays_of(rays); top.occluded1M(c)
        out[cfg] = (a, b, c)
        if instanced: top.release()
        obj.release()
    w, p = out["gpu=0,packed_link=0"], out["gpu=0"]
    hit = w[0]["geomID"] != 777
    assert 0.02 < hit.mean() < 0.95 and (w[0]["geomID"][hit] == 0).all()                       # (the whole-record path leaves 777 where a ray missed: the kernel wrote nothing)
    if instanced: assert len(set(w[0]["instID"][hit].tolist())) == 5
    assert (w[0]["Ng_z"][~hit] == np.float32(1e30)).all() and (w[0]["primID"][~hit] == 12345).all()
    for k in range(3):
        assert p[k].tobytes() == w[k].tobytes(), "packed link differs from the whole-record path (%s)" % ("closest, chunked", "closest, one chunk", "occluded")[k]
    assert (w[2]["tfar"] == -np.inf).sum() > 0.02 * n


def test_user_data_change_reaches_the_device_filter_function(api, ref):
    """with device_filter_functions=1 the user pointer of a geometry that enabled the argument filter travels to the GPU in the rule table at commit.
    rtcSetGeometryUserData did not mark anything modified, so rtcCommitScene returned early and the function kept seeing the OLD pointer.  Now a changed pointer counts as a
    changed rule: after set + commit the function (tests/dev_filter.hip adds args->geometryUserPtr into a device counter) sees the new one."""
    from tests.test_gpu_round3 import _rule_scene_meshes, _rule_rays
    devfilter = os.path.join(ROOT, "tests", "golden", "_bin", "libdevfilter.so")
    assert os.path.exists(devfilter), "tests/golden/_bin/libdevfilter.so is missing: __graft_entry__.build() step 6"
    L = api.load()
    lib = C.CDLL(devfilter)
    lib.devfilter_address.restype = C.c_uint64
    fn = lib.devfilter_address()
    assert fn != 0
    meshes, rays = _rule_scene_meshes(), _rule_rays()
    fdev = api.Device("gpu=0,device_filter_functions=1")
    s = api.make_scene(fdev, meshes)
    hg = L.rtcGetGeometry(s.h, 0)
    L.rtcSetGeometryEnableFilterFunctionFromArguments(hg, True)
    counters = api.DeviceArray.from_numpy(np.zeros(3, np.uint64))

    def run(user):
        L.rtcSetGeometryUserData(hg, C.c_void_p(user))
        s.commit()
        L.mi355_memcpy_h2d(counters.ptr, np.zeros(3, np.uint64).ctypes.data, 24)
        d = api.DeviceArray.from_numpy(rays)
        qa = api.QueryArguments(None, 0)
        qa.filter, qa.context = C.c_void_p(fn), C.c_void_p(counters.ptr)
        s.intersect1M_device(d.ptr, rays.shape[0], args=qa)
        L.mi355_device_synchronize(0)
        d.free()
        c = counters.download(np.uint64)
        return int(c[0]), int(c[2])
    calls_a, sum_a = run(0x1000)
    calls_b, sum_b = run(0x2000)
    # (the NUMBER of calls is not a constant of the query: a candidate reaches the function only if it is nearer than what its ray has so far, and the order in which a ray's
    # candidates arrive depends on which wave helped -- 1388 against 1389 calls in one run of round 6; what must hold is that EVERY call saw the pointer of its commit)
    assert calls_a > 0 and abs(calls_a - calls_b) <= max(4, calls_a // 100)
    assert sum_a == calls_a * 0x1000 and sum_b == calls_b * 0x2000, "the device filter function still sees the old user pointer: %x / %d calls" % (sum_b, calls_b)
    counters.free()
    s.release()
    fdev.release()


# ------------------------------------------------------------------------------------------- device filter functions in instanced scenes, at -O3
@pytest.mark.parametrize("flags", [0, 4])                       # fast, RTC_SCENE_FLAG_ROBUST
def test_device_filter_function_in_an_instanced_scene_vs_reference_callback(api, ref, flags):
    """Round 5 refused a __device__ filter function in scenes with instances (hipErrorNotSupported) and compiled the calling kernels at -O1.  Round 6: the calling kernels are
    built at -O3 without the record prefetch (what breaks them above -O1: profiles/r06_device_filter.md) and exist for instanced scenes as well; the function sees the
    candidate's primID / geomID inside the instanced scene and the instance's id in hit.instID[0] (kernels/geometry/filter_sycl.h:12-120, instance_stack.h:19-50).  Checker:
    the REAL reference running the same rule (reject (primID + 2 geomID) % 5 == 1) as the argument filter callback of an enforcing query over the same two-level scene."""
    from tests.test_gpu_round3 import _rule_scene_meshes, _rule_rays, _tri_t64
    devfilter = os.path.join(ROOT, "tests", "golden", "_bin", "libdevfilter.so")
    assert os.path.exists(devfilter), "tests/golden/_bin/libdevfilter.so is missing: __graft_entry__.build() step 6"
    L = api.load()
    lib = C.CDLL(devfilter)
    lib.devfilter_address.restype = C.c_uint64
    fn = lib.devfilter_address()
    assert fn != 0
    meshes, rays = _rule_scene_meshes(), _rule_rays()
    xf = [[1, 0, 0, 0, 1, 0, 0, 0, 1, 0.0, 0, 0], [0.8, 0, 0, 0, 0.8, 0, 0, 0, 0.8, 0.45, 0.1, -0.05]]
    fdev = api.Device("gpu=0,device_filter_functions=1")
    obj = api.make_scene(fdev, meshes[:2], flags=flags)
    top = api.Scene(fdev, flags)
    for x in xf:
        top.add_instance(obj, x)
    top.add_triangle_mesh(*meshes[2])
    top.commit()
    R = ref.RefScene(flags=flags)
    Robj = R.new_object(flags=flags)
    for v, t in meshes[:2]:
        Robj.add_mesh(v, t)
    Robj.commit()
    for x in xf:
        R.add_instance(Robj, x)
    R.add_mesh(*meshes[2])
    R.commit()
    assert R.error() == 0
    plain = rays.copy()
    top.intersect1M(plain)
    counters = api.DeviceArray.from_numpy(np.zeros(3, np.uint64))
    enforce = api.RTC_RAY_QUERY_FLAG_INVOKE_ARGUMENT_FILTER

    def device_query(any_hit):
        src = rays_of(rays) if any_hit else rays
        d = api.DeviceArray.from_numpy(src)
        qa = api.QueryArguments(None, enforce)
        qa.filter, qa.context = C.c_void_p(fn), C.c_void_p(counters.ptr)
        (top.occluded1M_device if any_hit else top.intersect1M_device)(d.ptr, src.shape[0], args=qa)
        L.mi355_device_synchronize(0)
        assert top.trace_status() == 0
        out = d.download(RAY_DTYPE if any_hit else RAYHIT_DTYPE)
        d.free()
        return out
    want = rays.copy()
    R.intersect1_args(want, arg_rule=True, flags=enforce)
    got = device_query(False)
    same = (got["geomID"] == want["geomID"]) & (got["primID"] == want["primID"]) & (got["instID"] == want["instID"])
    hit = want["geomID"] != INVALID_ID
    assert ((got["geomID"] != INVALID_ID) == hit).mean() > 0.998 and same.mean() > 0.995, "IDs differ on %d of %d rays" % (int((~same).sum()), rays.shape[0])   # (exact-t ties between the two instances' copies are legal)
    m = same & hit
    assert (np.abs(got["tfar"][m] - want["tfar"][m]) <= 1e-4 * np.abs(want["tfar"][m])).all()
    assert int(((got["primID"] != plain["primID"]) | (got["geomID"] != plain["geomID"]) | (got["instID"] != plain["instID"])).sum()) > 500, "the function rejected next to nothing"
    calls = counters.download(np.uint64)
    assert calls[0] > 0 and 0 < calls[1] <= calls[0]
    wr = rays_of(rays)
    R.occluded1_args(wr, arg_rule=True, flags=enforce)
    gr = device_query(True)
    compare_occluded(gr["tfar"], wr["tfar"], rays_of(rays)["tfar"], max_flip_frac=2e-3, label="device filter function in an instanced scene, occlusion")
    R.close(); Robj.close()
    counters.free(); top.release(); obj.release(); fdev.release()


# ------------------------------------------------------------------------------------------- RTC_RAY_QUERY_FLAG_COHERENT is a hint with a memory -- say so, and let it be switched off
def test_coherent_flag_on_an_alternating_stream_with_and_without_its_memory(api):
    """RTC_RAY_QUERY_FLAG_COHERENT sends a batch to the wave-packet kernel; large batches sample every 32nd packet first and hand the batch to the per-lane kernel when the
    sample's packets fall apart.  Three such verdicts in a row are REMEMBERED per (tree, stream): the next 15 flagged queries skip the sample -- right for a renderer that sets
    the flag on everything, wrong-footed by one that alternates coherent primary batches and incoherent bounce batches on ONE stream (the memory never settles: every bounce
    batch pays its sample, or -- after three bounce batches in a row -- a primary batch is sent to the per-lane kernel).  rtcNewDevice("coherent_memory=0") makes every query
    decide on its own sample.  The ANSWERS never depend on any of this: byte-identical to the unflagged query for every interleaving, with and without the memory.  The
    timings are printed (INTEGRATION.md names the knob); the one bound asserted is that no interleaving costs a flagged stream more than twice the unflagged one."""
    import time
    L = api.load()
    meshes = W.synthetic_crown(num_phi=64)
    out = {}
    for cfg in ("gpu=0", "gpu=0,coherent_memory=0"):
        d_ = api.Device(cfg)
        s = api.make_scene(d_, meshes)
        prim = W.crown_camera_rays(meshes, 512, 512)
        hit = prim.copy(); s.intersect1M(hit)
        bounce = W.diffuse_bounce_rays(hit, meshes, seed=5)
        dp, db = api.DeviceArray.from_numpy(prim), api.DeviceArray.from_numpy(bounce)
        wp, wb = api.DeviceArray(prim.nbytes), api.DeviceArray(bounce.nbytes)
        st = C.c_void_p(); L.mi355_stream_create(0, C.byref(st))
        want_p = want_b = None
        for flagged in (False, True):
            qa = api.QueryArguments(None, api.RTC_RAY_QUERY_FLAG_COHERENT if flagged else 0)
            for order in ("pbpbpbpbpbpb", "bbbbpbbbbpbbbbp", "pppbpppbpppb"):
                best = None
                for rep in range(3):                                  # (wall clock on a shared host: one run of round 6 had a lone 6.25 ms among 2.2 ms -- the best of three is the stream's cost)
                    L.mi355_synchronize(st)
                    t0 = time.perf_counter()
                    for ch in order:
                        src, dst, n = (dp, wp, prim) if ch == "p" else (db, wb, bounce)
                        L.mi355_memcpy_d2d_async(dst.ptr, src.ptr, n.nbytes, st)
                        s.intersect1M_device(dst.ptr, n.shape[0], stream=st, args=qa)
                    L.mi355_synchronize(st)
                    dt = (time.perf_counter() - t0) * 1e3
                    best = dt if best is None else min(best, dt)
                out[(cfg, flagged, order)] = best
                assert s.trace_status(st) == 0
                rp, rb = wp.download(RAYHIT_DTYPE), wb.download(RAYHIT_DTYPE)
                if want_p is None:
                    want_p, want_b = rp, rb
                assert rp.tobytes() == want_p.tobytes() and rb.tobytes() == want_b.tobytes(), "flag %s, order %s, %s: the answers moved" % (flagged, order, cfg)
        for a_ in (dp, db, wp, wb):
            a_.free()
        s.release(); d_.release()
    for k, v in sorted(out.items()):
        print("coherent stream cfg=%s flag=%s order=%s: %.2f ms" % (k[0], k[1], k[2], v))
    for (cfg, flagged, order), v in out.items():
        if flagged:
            assert v <= 2.0 * out[(cfg, False, order)] + 0.5, "flagged stream %s (%s): %.2f ms against %.2f ms unflagged" % (order, cfg, v, out[(cfg, False, order)])


# ------------------------------------------------------------------------------------------- the sort of the Morton build is this repository's own
@pytest.mark.parametrize("case", ["random", "few_values", "sorted", "reversed", "one_bit", "all_equal"])
@pytest.mark.parametrize("n", [1, 2, 63, 4095, 4096, 4097, 12289, 1000003])
def test_radix_sort_of_the_morton_build_vs_stable_argsort(api, case, n):
    """RTC_BUILD_QUALITY_LOW sorted its Morton codes with hipcub::DeviceRadixSort until round 6; the reference has its own radix sort
    (kernels/builders/bvh_builder_morton.h:439).  build_sort.inl: 7 passes of 9 bits, one kernel per pass with a look-back at the tiles in front.  The sort on its own
    (mi355_sort_keys63) against numpy's stable argsort: keys in order, equal keys in index order, every index exactly once -- sizes around the tile size of 4096, digit
    patterns that put all keys of a tile into one bucket, into two, into all."""
    L = api.load()
    rng = np.random.default_rng(n * 7 + len(case))
    if case == "random":
        keys = rng.integers(0, 1 << 63, n, dtype=np.uint64)
    elif case == "few_values":
        keys = rng.choice(rng.integers(0, 1 << 63, 5, dtype=np.uint64), n)
    elif case == "sorted":
        keys = np.sort(rng.integers(0, 1 << 63, n, dtype=np.uint64))
    elif case == "reversed":
        keys = np.sort(rng.integers(0, 1 << 63, n, dtype=np.uint64))[::-1].copy()
    elif case == "one_bit":
        keys = (rng.integers(0, 2, n, dtype=np.uint64) << np.uint64(int(rng.integers(0, 63))))
    else:
        keys = np.full(n, 0x1234567890ABCDEF & ((1 << 63) - 1), np.uint64)
    dk = api.DeviceArray.from_numpy(keys)
    ok, oi = api.DeviceArray(n * 8), api.DeviceArray(n * 4)
    ms = C.c_float()
    assert L.mi355_sort_keys63(0, dk.ptr, ok.ptr, oi.ptr, n, C.byref(ms)) == 0, L.mi355_last_error().decode()
    got_k, got_i = ok.download(np.uint64), oi.download(np.uint32)
    want_i = np.argsort(keys, kind="stable").astype(np.uint32)
    assert np.array_equal(got_i, want_i), "order differs from a stable sort at place %d" % int(np.nonzero(got_i != want_i)[0][0])
    assert np.array_equal(got_k, keys[want_i])
    assert np.array_equal(dk.download(np.uint64), keys), "the source array was written"
    if n >= 1000000:
        print("radix sort of %d keys (%s): %.3f ms" % (n, case, ms.value))
    for a_ in (dk, ok, oi):
        a_.free()


@pytest.mark.parametrize("case", ["random", "few_values"])
def test_radix_sort_at_sixteen_million_keys(api, case):
    """4097 tiles: every workgroup of a pass looks back over tiles that are still running (512 are resident at a time); the order must still be the stable one."""
    L = api.load()
    n = (1 << 24) + 5
    rng = np.random.default_rng(11)
    keys = rng.integers(0, 1 << 63, n, dtype=np.uint64) if case == "random" else rng.choice(rng.integers(0, 1 << 63, 300, dtype=np.uint64), n)
    dk = api.DeviceArray.from_numpy(keys)
    ok, oi = api.DeviceArray(n * 8), api.DeviceArray(n * 4)
    ms = C.c_float()
    assert L.mi355_sort_keys63(0, dk.ptr, ok.ptr, oi.ptr, n, C.byref(ms)) == 0, L.mi355_last_error().decode()
    got_i = oi.download(np.uint32)
    want_i = np.argsort(keys, kind="stable").astype(np.uint32)
    assert np.array_equal(got_i, want_i)
    assert np.array_equal(ok.download(np.uint64), keys[want_i])
    print("radix sort of %d keys (%s): %.3f ms" % (n, case, ms.value))
    for a_ in (dk, ok, oi):
        a_.free()


def test_low_quality_build_has_no_library_kernel():
    """the Morton build's translation unit no longer includes hipcub / rocprim: every kernel of every commit is this repository's"""
    src = os.path.join(ROOT, "embree_amd", "csrc")
    for f in os.listdir(src):
        code = "\n".join(line.split("//")[0] for line in open(os.path.join(src, f)).read().split("\n"))     # (comments may name what was replaced)
        assert "#include <hipcub" not in code and "#include <rocprim" not in code and "hipcub::" not in code and "rocprim::" not in code, f
