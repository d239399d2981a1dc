"""GPU tests added in round 3 (pytest -m gpu), all through the C ABI:

  * one RTCDevice over several GPUs (rtcNewDevice("gpus=N")): replicas bit-identical, the sharded host-array and device-array queries give the
    single-GPU answer bit for bit (on a 1-GPU box the replicas share the GPU: gpu_oversubscribe=1 -- same code, same threads, same peer copies);
  * rtcCommitScene after detach + a NEW geometry that reuses the freed one's address and counters (the advisor's stale-tree scenario);
  * filter callbacks in a ROBUST scene see every rejected candidate exactly once;
  * configs[3] as a whole job at N = 1 through bench.py's own code (pack + gather) against the real reference on a 2^22-ray prefix;
  * RTC_RAY_QUERY_FLAG_COHERENT: the wave-packet kernel gives the incoherent kernel's answers;
  * device-side filter rules against the same rule run as a host callback in the real reference.
"""
import ctypes as C
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from embree_amd import workloads as W
from embree_amd.rtypes import rays_of, make_rayhits, INVALID_ID, RAYHIT_DTYPE, RAY_DTYPE
from tests import bvh_check
from tests.helpers import compare_closest, compare_occluded

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


@pytest.fixture(scope="module")
def ref():
    from oracle import refembree
    if not refembree.available():
        pytest.skip("oracle/_ref not present on this box (make -f oracle/ref.mk in the build container)")
    return refembree


def tri_t_of(meshes):
    from oracle import restate
    o = restate.OracleScene()
    for v, t in meshes:
        o.add_mesh(v, t)
    return o.triangle_t


# ------------------------------------------------------------------------------------------- one RTCDevice over N GPUs
@pytest.mark.parametrize("gpus", [1, 2, 3, 8])
def test_one_device_over_several_gpus(api, dev, gpus):
    """rtcNewDevice("gpus=N") (the reference's shape: a GPU device behind the same RTCDevice, kernels/common/scene.cpp:866-872; the ray count of its own
    benchmark, tutorials/verify/verify.cpp:5933, is what gets sharded).  gpus=1 drives the very same code with one replica."""
    L = api.load()
    ngpu = L.mi355_device_count()
    cfg = "gpu=0,gpus=%d,shard_min=1024%s" % (gpus, ",gpu_oversubscribe=1" if gpus > ngpu else "")
    md = api.Device(cfg)
    assert md.gpu_count() == gpus
    meshes = W.synthetic_crown(num_phi=24)
    single = api.make_scene(dev, meshes)
    ms = api.make_scene(md, meshes)
    n0, t0 = single.download_bvh()
    for k in range(gpus):                                      # every replica is the single-GPU tree, bit for bit
        b = ms.replica_bvh(k)
        assert b
        i = api.BvhInfo()
        L.mi355_bvh_get_info(b, C.byref(i))
        nodes = np.zeros(i.num_nodes, api.NODE_DTYPE)
        tris = np.zeros(i.num_triangles, api.TRI_DTYPE)
        assert L.mi355_bvh_download(b, nodes.ctypes.data, nodes.nbytes, tris.ctypes.data, tris.nbytes) == 0
        assert nodes.tobytes() == n0.tobytes() and tris.tobytes() == t0.tobytes(), "replica %d differs from the single-GPU tree" % k
    assert ms.replica_bvh(gpus) is None
    prim = W.crown_camera_rays(meshes, 256, 256)
    a = prim.copy()
    single.intersect1M(a)
    rays = W.diffuse_bounce_rays(a, meshes)
    for M in (rays.shape[0], 40001, 3000):                     # sharded, sharded with ragged shards, below shard_min * N (one replica)
        want, got = rays[:M].copy(), rays[:M].copy()
        single.intersect1M(want)
        ms.intersect1M(got)
        assert got.tobytes() == want.tobytes(), "sharded rtcIntersect1M differs from the single-GPU answer (M = %d)" % M
        wr, gr = rays_of(rays[:M]), rays_of(rays[:M])
        single.occluded1M(wr)
        ms.occluded1M(gr)
        assert gr.tobytes() == wr.tobytes()
    # device-array form: the array lives on the first GPU, shards travel peer to peer
    want = rays.copy()
    single.intersect1M(want)
    d = api.DeviceArray.from_numpy(rays, 0)
    st = C.c_void_p()
    L.mi355_stream_create(0, C.byref(st))
    ms.intersect1M_device(d.ptr, rays.shape[0], stream=st)
    L.mi355_synchronize(st)
    assert d.download(RAYHIT_DTYPE).tobytes() == want.tobytes(), "sharded rtcIntersect1MDevice differs from the single-GPU answer"
    r48 = rays_of(rays)
    wr = r48.copy()
    single.occluded1M(wr)
    d2 = api.DeviceArray.from_numpy(r48, 0)
    ms.occluded1M_device(d2.ptr, r48.shape[0], stream=st)
    L.mi355_synchronize(st)
    assert d2.download(RAY_DTYPE).tobytes() == wr.tobytes()
    # a refit and an instanced scene on every replica
    if gpus > 1:
        obj = api.make_scene(md, [meshes[0]])
        top = api.Scene(md)
        top.add_instance(obj, [1, 0, 0, 0, 1, 0, 0, 0, 1, 0.25, 0, 0])
        top.add_triangle_mesh(*meshes[1])
        top.commit()
        obj1 = api.make_scene(dev, [meshes[0]])
        top1 = api.Scene(dev)
        top1.add_instance(obj1, [1, 0, 0, 0, 1, 0, 0, 0, 1, 0.25, 0, 0])
        top1.add_triangle_mesh(*meshes[1])
        top1.commit()
        w, g = rays.copy(), rays.copy()
        top1.intersect1M(w)
        top.intersect1M(g)
        assert g.tobytes() == w.tobytes(), "instanced scene on replicas differs"
        for s_ in (top, obj, top1, obj1):
            s_.release()
    L.mi355_stream_destroy(st)
    d.free(); d2.free()
    ms.release(); single.release(); md.release()


def test_more_gpus_than_the_node_has_is_an_error(api):
    L = api.load()
    n = L.mi355_device_count()
    with pytest.raises(api.RTCErrorException) as e:
        api.Device("gpus=%d" % (n + 1))
    assert e.value.code == api.RTC_ERROR_INVALID_ARGUMENT
    L.rtcGetDeviceError(None)


# ------------------------------------------------------------------------------------------- advisor: stale tree after detach + new geometry
def test_commit_after_detach_and_a_new_geometry_at_the_same_address(api, dev):
    """rtcDetachGeometry frees the old geometry (the scene held the last reference); rtcNewGeometry very likely returns the same address; the same buffer-set
    sequence gives the same counters and rtcAttachGeometry the same id.  rtcCommitScene must build the NEW geometry's tree (Scene::commit compares
    process-unique serials, not addresses)."""
    L = api.load()
    v0 = np.array([[0, 0, 1], [1, 0, 1], [0, 1, 1]], np.float32)
    t = np.array([[0, 1, 2]], np.uint32)
    ray = make_rayhits([[0.2, 0.2, 0]], [[0, 0, 1]])
    for trial in range(8):
        s = api.Scene(dev)
        s.add_triangle_mesh(v0, t, shared=False)
        s.commit()
        r = ray.copy(); s.intersect1M(r)
        assert r["geomID"][0] == 0 and abs(r["tfar"][0] - 1.0) < 1e-6
        L.rtcDetachGeometry(s.h, 0)
        dev.check()
        gid = s.add_triangle_mesh(v0 + np.float32([0, 0, 2.0 + trial]), t, shared=False)      # same sequence of calls, other vertices
        assert gid == 0
        s.commit()
        r = ray.copy(); s.intersect1M(r)
        assert r["geomID"][0] == 0 and abs(r["tfar"][0] - (3.0 + trial)) < 1e-5, "the scene still answers with the detached geometry's tree (t = %g)" % r["tfar"][0]
        s.release()


# ------------------------------------------------------------------------------------------- advisor: robust scenes, a rejected candidate is offered once
@pytest.mark.parametrize("flags", [0, 4])
def test_filter_sees_every_rejected_candidate_once(api, dev, flags):
    """A stack of 6 parallel triangles, a filter that rejects everything and counts: the callback must see each triangle exactly once per ray, fast
    (strict at tnear) and robust (Pluecker, inclusive at tnear) alike -- a transparency accumulation would otherwise count layers twice."""
    vs, ts = [], []
    for k in range(6):
        vs += [[-1, -1, 1 + k], [3, -1, 1 + k], [-1, 3, 1 + k]]
        ts.append([3 * k, 3 * k + 1, 3 * k + 2])
    s = api.make_scene(dev, [(np.array(vs, np.float32), np.array(ts, np.uint32))], flags=flags)
    seen = []

    def reject_all(a):
        a = a.contents
        for i in range(a.N):
            if a.valid[i] != -1:
                continue
            seen.append(int(C.cast(a.hit, C.POINTER(C.c_uint32))[5 * a.N + i]))
            a.valid[i] = 0
    f = api.FILTER_FN(reject_all)
    s.set_filters(0, intersect=f, occluded=f)
    r = make_rayhits([[0.1, 0.1, 0]], [[0, 0, 1]])
    s.intersect1M(r)
    assert r["geomID"][0] == INVALID_ID and sorted(seen) == [0, 1, 2, 3, 4, 5], seen
    seen.clear()
    o = rays_of(make_rayhits([[0.1, 0.1, 0]], [[0, 0, 1]]))
    s.occluded1M(o)
    assert not np.isneginf(o["tfar"][0]) and sorted(seen) == [0, 1, 2, 3, 4, 5], seen
    s.release()


# ------------------------------------------------------------------------------------------- configs[3]: the whole job through bench.py's code
def test_shadow16m_whole_job_vs_reference_all_rays(api, ref, tmp_path):
    """configs[3] at N = 1 exactly as the driver would run it: bench.py --workload shadow16m (16 Mi shadow rays through rtcOccluded1MDevice, results packed on the GPU
    and gathered with RCCL -- one rank: the all-gather is a copy -- on the communication stream, inside the timed step), the gathered words dumped, and ALL
    16,777,216 of them checked against the REAL reference's rtcOccluded1 (round 3: a 2^22-ray prefix)."""
    dump = str(tmp_path / "shadow.npz")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "shadow16m", "--gather", "rccl", "--steps", "3", "--warmup", "1", "--no-cpu", "--dump", dump,
                        "--dump-rays", str(1 << 24)], capture_output=True, text=True, timeout=1500, env=env)
    assert p.returncode == 0, p.stderr[-3000:]
    line = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 1 and line["config"]["rays_per_gpu"] == 16 * (1 << 20) and line["scaling"] == "strong"
    assert "error" not in line.get("gather", {}) and line["gather"]["inside_timed_region"] and line["rccl_ranks"] == 1, line.get("gather")
    z = np.load(dump)
    rays, words = z["rays"], z["gathered"]
    assert rays.shape[0] == 1 << 24 and words.shape[0] == 1 << 24
    meshes = W.synthetic_crown(num_phi=158)
    R = ref.RefScene("threads=%d" % min(16, ref.hw_threads()))
    for v, t in meshes:
        R.add_mesh(v, t)
    R.commit()
    want = rays.copy()
    R.occluded1(want, ref.hw_threads())
    st = compare_occluded(words.view(np.float32), want["tfar"], rays["tfar"], max_flip_frac=1e-5, label="shadow16m, all rays, vs reference")
    print("shadow16m whole job: %.1f Mrays/s; all %d rays vs reference: %s" % (line["value"], rays.shape[0], st))
    R.close()


# ------------------------------------------------------------------------------------------- RTC_RAY_QUERY_FLAG_COHERENT: the wave-packet kernel
@pytest.mark.parametrize("flags", [0, 4])
def test_coherent_flag_gives_the_incoherent_answers(api, dev, flags):
    """RTC_RAY_QUERY_FLAG_COHERENT selects the wave-packet kernel (the reference: intersectCoherent, kernels/bvh/bvh_intersector_hybrid.cpp:374-533; a hint,
    never a change of the answer).  Both kernels report the minimum of (t, triangle index) over all accepted candidates, so the results must be the same BYTES:
    coherent primary rays, incoherent rays (a broken promise must still be answered correctly), occlusion, masks, ragged counts, quads, fast and robust."""
    qa = api.QueryArguments(None, api.RTC_RAY_QUERY_FLAG_COHERENT)
    cases = []
    m = W.cornell_box()
    cases.append(("cornell primary", m, None, W.cornell_camera_rays(300, 217)))                         # 65100 rays: the last packet is ragged
    m = W.synthetic_crown(num_phi=32)
    prim = W.crown_camera_rays(m, 256, 256)
    cases.append(("crown primary", m, None, prim))
    cases.append(("crown incoherent", m, None, W.incoherent_rays(50000, [2, 2, 1.5], seed=9)))
    # batches of >= 1024 packets are traced as a sample (every 32nd packet) and the rest: the rest goes straight to the per-lane kernel when the sample's
    # packets fell apart (incoherent rays), through the packet kernel otherwise (primary rays)
    cases.append(("crown primary, sampled", m, None, W.crown_camera_rays(m, 640, 480)))
    cases.append(("crown incoherent, sampled", m, None, W.incoherent_rays(300001, [2, 2, 1.5], seed=11)))
    m2 = W.synthetic_crown(num_phi=12)
    masks = [1 << (i % 3) for i in range(len(m2))]
    r = W.incoherent_rays(30001, [2, 2, 1.5], seed=3)
    r["mask"] = np.where(np.arange(r.shape[0]) % 2 == 0, 3, 4).astype(np.uint32)
    cases.append(("masks", m2, masks, r))
    for name, meshes, masks, rays in cases:
        s = api.make_scene(dev, meshes, masks, flags=flags)
        a, b = rays.copy(), rays.copy()
        s.intersect1M(a)
        s.intersect1M(b, qa)
        assert (a["geomID"] != INVALID_ID).any()
        assert a.tobytes() == b.tobytes(), "%s: the packet kernel's closest hits differ on %d rays" % (name, int((a["primID"] != b["primID"]).sum()))
        oa, ob = rays_of(rays), rays_of(rays)
        s.occluded1M(oa)
        s.occluded1M(ob, qa)
        assert oa.tobytes() == ob.tobytes(), "%s: occlusion differs" % name
        # device-pointer form and the 8-wide packet entry point with the flag
        d = api.DeviceArray.from_numpy(rays)
        s.intersect1M_device(d.ptr, rays.shape[0], args=qa)
        api.load().mi355_device_synchronize(0)
        assert d.download(RAYHIT_DTYPE).tobytes() == a.tobytes()
        d.free()
        assert s.trace_status() == 0
        s.release()
    # quads
    rng = np.random.default_rng(5)
    gx, gy = np.meshgrid(np.arange(20), np.arange(20))
    v = np.stack([gx.ravel() * 0.1, gy.ravel() * 0.1, rng.random(400) * 0.05], -1).astype(np.float32)
    q = np.array([[j * 20 + i, j * 20 + i + 1, (j + 1) * 20 + i + 1, (j + 1) * 20 + i] for j in range(19) for i in range(19)], np.uint32)
    s = api.Scene(dev, flags)
    s.add_quad_mesh(v, q)
    s.commit()
    org = np.stack([rng.random(20000) * 1.9, rng.random(20000) * 1.9, np.full(20000, 1.0)], -1).astype(np.float32)
    rays = make_rayhits(org, np.tile(np.float32([0.01, -0.02, -1]), (20000, 1)))
    a, b = rays.copy(), rays.copy()
    s.intersect1M(a)
    s.intersect1M(b, qa)
    assert (a["geomID"] == 0).mean() > 0.9 and a.tobytes() == b.tobytes()
    s.release()


# ------------------------------------------------------------------------------------------- device-side filter rules
def _rule_scene_meshes():
    from tests.test_gpu_reference_suite import _sticks
    return [_sticks(400, 3), W.triangle_sphere(np.array([0.5, 0.5, 0.5], np.float32), 0.3, 40, noise=0.1, seed=5),
            W.triangle_sphere(np.array([0.5, 0.5, 0.5], np.float32), 0.55, 30, noise=0.05, seed=6)]


def _rule_rays(n=12000, seed=77):
    rng = np.random.default_rng(seed)
    org = rng.random((n, 3), dtype=np.float32) * 1.8 - 0.4
    tgt = rng.random((n, 3), dtype=np.float32)
    return make_rayhits(org, (tgt - org) * np.float32(2.0))


def _tri_t64(meshes):
    def tri_t(rr, geom, prim):
        out = np.zeros(rr.shape[0], np.float32)
        for k in range(rr.shape[0]):
            v, t = meshes[int(geom[k])]
            a, b, c = v[t[int(prim[k])]].astype(np.float64)
            o = np.array([rr["org_x"][k], rr["org_y"][k], rr["org_z"][k]], np.float64)
            d = np.array([rr["dir_x"][k], rr["dir_y"][k], rr["dir_z"][k]], np.float64)
            nrm = np.cross(b - a, c - a)
            out[k] = np.dot(nrm, a - o) / np.dot(nrm, d)
        return out
    return tri_t


@pytest.mark.parametrize("flags", [0, 4])
def test_device_filter_rules_vs_reference_callbacks(api, dev, ref, flags):
    """rtcSetGeometryFilterRule: the reference's geometry rule of oracle/ref_driver.cpp ("reject primID % 3 == 0 or u > 0.7") installed as a RULE that runs
    inside the traversal kernels, against the real reference running it as a filter CALLBACK inside its traversal (kernels/geometry/filter.h:14-80,
    intersector_epilog.h:235-368): closest accepted hit and occlusion must be the reference's, through the host-array AND the device-pointer entry points
    (which cannot run callbacks), with the packet kernel, fast and robust; and the same rule as a host callback here gives the same bytes."""
    meshes = _rule_scene_meshes()
    rays = _rule_rays()
    r = ref.RefScene(flags=flags)
    for v, t in meshes:
        r.add_mesh(v, t)
    r.commit()
    s = api.make_scene(dev, meshes, flags=flags)
    plain = rays.copy()
    s.intersect1M(plain)
    rule = api.FilterRule(kinds=api.RTC_FILTER_RULE_MODULO | api.RTC_FILTER_RULE_UV_CUTOFF, apply=api.RTC_FILTER_RULE_APPLY_INTERSECT | api.RTC_FILTER_RULE_APPLY_OCCLUDED,
                          modulus=3, remainder=0, primFactor=1, geomFactor=0, tmin=0, tmax=0, umax=0.7, vmax=np.inf, bits=None, numBits=0)
    for g in range(len(meshes)):
        s.set_filter_rule(g, rule)
    s.commit()
    r.set_filters(len(meshes), 1 | 2)
    want = rays.copy()
    r.intersect1_args(want)
    got = rays.copy()
    s.intersect1M(got)
    st = compare_closest(got, want, rays, _tri_t64(meshes), max_tie_frac=2e-3, label="device rule vs reference callback")
    changed = int(((got["primID"] != plain["primID"]) | (got["geomID"] != plain["geomID"])).sum())
    assert changed > 1000
    d = api.DeviceArray.from_numpy(rays)                       # the device-pointer entry point filters as well
    s.intersect1M_device(d.ptr, rays.shape[0])
    api.load().mi355_device_synchronize(0)
    assert d.download(RAYHIT_DTYPE).tobytes() == got.tobytes()
    d.free()
    coh = rays.copy()                                          # ... and so does the packet kernel
    s.intersect1M(coh, api.QueryArguments(None, api.RTC_RAY_QUERY_FLAG_COHERENT))
    assert coh.tobytes() == got.tobytes()
    wr, gr = rays_of(rays), rays_of(rays)
    r.occluded1_args(wr)
    s.occluded1M(gr)
    compare_occluded(gr["tfar"], wr["tfar"], rays_of(rays)["tfar"], max_flip_frac=2e-3, label="device rule, occlusion")
    # the same rule as a HOST callback on a second scene: same bytes
    def rule_geometry(a):
        a = a.contents
        for i in range(a.N):
            if a.valid[i] != -1:
                continue
            prim = C.cast(a.hit, C.POINTER(C.c_uint32))[5 * a.N + i]
            if prim % 3 == 0 or a.hit[3 * a.N + i] > 0.7:
                a.valid[i] = 0
    f = api.FILTER_FN(rule_geometry)
    s2 = api.make_scene(dev, meshes, flags=flags)
    for g in range(len(meshes)):
        s2.set_filters(g, intersect=f, occluded=f)
    host = rays.copy()
    s2.intersect1M(host)
    assert host.tobytes() == got.tobytes(), "device rule and host callback disagree on %d rays" % int((host["primID"] != got["primID"]).sum())
    # rule removed again: the plain answers come back, without a rebuild
    n0 = s.info()["num_nodes"]
    for g in range(len(meshes)):
        s.set_filter_rule(g, None)
    s.commit()
    back = rays.copy()
    s.intersect1M(back)
    assert back.tobytes() == plain.tobytes() and s.info()["num_nodes"] == n0
    print("device rule: %d hits, %d rays changed, %d ties" % (st["hits"], changed, st["ties"]))
    s.release(); s2.release(); r.close()


def test_device_filter_rule_kinds_and_instances(api, dev):
    """The other rule kinds (per-primitive bit array = alpha mask, distance window) against the same predicate as a host callback, and a rule on a geometry
    INSIDE an instanced scene (host callbacks cannot run there: rtcore_api.cpp filtered_query refuses instances)."""
    meshes = _rule_scene_meshes()
    rays = _rule_rays(8000, seed=5)
    rng = np.random.default_rng(1)
    nb = meshes[1][1].shape[0]
    bits = (rng.random(nb) < 0.4)
    words = np.zeros((nb + 31) // 32, np.uint32)
    for i in np.nonzero(bits)[0]:
        words[i >> 5] |= np.uint32(1 << (int(i) & 31))
    s = api.make_scene(dev, meshes)
    s.set_filter_rule(1, api.FilterRule(kinds=api.RTC_FILTER_RULE_PRIMITIVE_BITS, apply=3, bits=words.ctypes.data, numBits=nb))
    s.set_filter_rule(2, api.FilterRule(kinds=api.RTC_FILTER_RULE_DISTANCE_WINDOW, apply=3, tmin=0.2, tmax=0.45))
    s.commit()
    got = rays.copy()
    s.intersect1M(got)

    def cb(a):
        a = a.contents
        for i in range(a.N):
            if a.valid[i] != -1:
                continue
            hp = C.cast(a.hit, C.POINTER(C.c_uint32))
            prim, geom, t = hp[5 * a.N + i], hp[6 * a.N + i], a.ray[8 * a.N + i]
            if (geom == 1 and bits[prim]) or (geom == 2 and not (np.float32(0.2) <= np.float32(t) <= np.float32(0.45))):
                a.valid[i] = 0
    f = api.FILTER_FN(cb)
    s2 = api.make_scene(dev, meshes)
    for g in range(len(meshes)):
        s2.set_filters(g, intersect=f, occluded=f)
    host = rays.copy()
    s2.intersect1M(host)
    assert host.tobytes() == got.tobytes(), "bit-array / window rules differ from the host callback on %d rays" % int((host["primID"] != got["primID"]).sum())
    assert (got["geomID"] == 1).any() and not bits[got["primID"][got["geomID"] == 1]].any()
    og, oh = rays_of(rays), rays_of(rays)
    s.occluded1M(og); s2.occluded1M(oh)
    assert og.tobytes() == oh.tobytes()
    # ---- a rule inside an instanced scene: the instance of a filtered object answers like the filtered object itself, moved
    obj = api.make_scene(dev, [meshes[1]])
    obj.set_filter_rule(0, api.FilterRule(kinds=api.RTC_FILTER_RULE_PRIMITIVE_BITS, apply=3, bits=words.ctypes.data, numBits=nb))
    obj.commit()
    top = api.Scene(dev)
    shift = np.float32([0.25, -0.1, 0.05])
    top.add_instance(obj, [1, 0, 0, 0, 1, 0, 0, 0, 1, shift[0], shift[1], shift[2]])
    top.add_triangle_mesh(*meshes[0])
    top.commit()
    a = rays.copy()
    top.intersect1M(a)
    inst = a["instID"][:, 0] if a["instID"].ndim > 1 else a["instID"]
    hit_inst = (a["geomID"] != INVALID_ID) & (inst == 0)
    assert hit_inst.sum() > 100 and not bits[a["primID"][hit_inst]].any(), "a masked primitive of the instanced object was reported"
    moved = rays.copy()                                        # the same rays in the object's space against the object alone
    moved["org_x"] -= shift[0]; moved["org_y"] -= shift[1]; moved["org_z"] -= shift[2]
    b = moved.copy()
    obj.intersect1M(b)
    closer = hit_inst & (b["geomID"] != INVALID_ID)
    assert closer.sum() > 100 and (np.abs(a["tfar"][closer] - b["tfar"][closer]) <= 1e-4 * np.abs(b["tfar"][closer]) + 1e-6).mean() > 0.95
    for x in (top, obj, s, s2):
        x.release()


def test_deep_tree_beyond_the_enqueued_levels(api, dev, restate):
    """The one-round-trip commit enqueues 16 levels of the wide collapse; a tree that is deeper is finished level by level afterwards and its leaf records are
    written again.  Triangles whose size and position double from one to the next make every SAH split peel one triangle off: a tree as deep as it gets."""
    n = 460                                                   # sizes up to 2^57.5: inside the reference's validity limit of 1.844e18
    vs, ts = [], []
    for i in range(n):
        s_ = np.float32(2.0) ** np.float32(i * 0.125)
        x = np.float32(3.0) * s_
        y = np.float32(2 * (i % 8)) * s_                             # neighbours in x do not overlap: no coplanar ties
        vs += [[x, y, 0], [x + s_, y, 0], [x, y + s_, 0]]
        ts.append([3 * i, 3 * i + 1, 3 * i + 2])
    meshes = [(np.array(vs, np.float32), np.array(ts, np.uint32))]
    s = api.make_scene(dev, meshes)
    info = s.info()
    nodes, tris = s.download_bvh()
    bvh_check.validate(nodes, tris, info["root_ref"], meshes, max_leaf=info["max_leaf"])
    assert info["depth"] > 18, info["depth"]
    o = restate.OracleScene()
    o.add_mesh(*meshes[0]); o.commit()
    c = np.array(vs, np.float32).reshape(n, 3, 3).mean(1)
    rays = make_rayhits(c + np.float32([0, 0, 1]), np.tile(np.float32([0, 0, -1]), (n, 1)))
    want, got = rays.copy(), rays.copy()
    o.intersect1(want); s.intersect1M(got)
    st = compare_closest(got, want, rays, o.triangle_t, label="deep tree")
    assert st["hits"] > n // 2                               # (at coordinates of 1e17 some centre rays miss in fp32: the reference misses the same ones)
    s.release()


# ------------------------------------------------------------------------------------------- builder knobs
@pytest.mark.parametrize("cfg", ["min_leaf=1", "min_leaf=1,max_leaf=1", "min_leaf=3,max_leaf=3", "small_threshold=128", "small_threshold=4096", "small_threshold=64"])
def test_builder_knobs_keep_the_tree_valid(api, dev, cfg):
    """The small phase has paths the default configuration never takes: 48 instead of 32 LDS words per triangle (min_leaf = 1: sets of two triangles are
    split), a root list that fills up and is worked off before the sub-tree is finished (small_threshold = 4096: up to 127 roots of <= 64 triangles, 32
    parked at a time), sub-trees that are ALL micro roots (small_threshold = 64).  Every configuration must give a valid tree (each triangle once, every box
    contains what is below it) and the hits of the default tree (same t; an ID may differ where two triangles are hit at the same distance)."""
    meshes = W.synthetic_crown(num_phi=24)                     # ~110 k triangles
    rng = np.random.default_rng(5)
    lo = np.min([v.min(0) for v, _ in meshes], 0); hi = np.max([v.max(0) for v, _ in meshes], 0)
    org = (lo + (hi - lo) * rng.random((20000, 3))).astype(np.float32)
    d = rng.normal(size=(20000, 3)).astype(np.float32)
    rays = make_rayhits(org, d)
    base = api.make_scene(dev, meshes)
    want = rays.copy(); base.intersect1M(want)
    d2 = api.Device("gpu=0," + cfg)
    try:
        s = api.make_scene(d2, meshes)
        info = s.info()
        nodes, tris = s.download_bvh()
        bvh_check.validate(nodes, tris, info["root_ref"], meshes, max_leaf=info["max_leaf"], allow_splits=info["num_presplit"] > 0)
        got = rays.copy(); s.intersect1M(got)
        hit = want["geomID"] != INVALID_ID
        assert ((got["geomID"] != INVALID_ID) == hit).all()
        assert (got["tfar"][hit] == want["tfar"][hit]).all(), "a different tree over the same triangles changed a hit distance"
        same = (got["geomID"] == want["geomID"]) & (got["primID"] == want["primID"])
        assert (~same[hit]).mean() < 1e-3, "IDs differ on more rays than coincident triangles explain"
        again = api.make_scene(d2, meshes)                     # and the same tree again
        n2, t2 = again.download_bvh()
        assert nodes.tobytes() == n2.tobytes() and tris.tobytes() == t2.tobytes(), "two commits of the same scene differ"
        again.release(); s.release()
    finally:
        base.release(); d2.release()
