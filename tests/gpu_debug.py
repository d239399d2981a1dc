"""Step-by-step bring-up script for the GPU box (not a pytest file): prints what each stage does so that one
gpurun call localises a failure.  Usage:  timeout 600 python tests/gpu_debug.py [stage ...]"""
import os
import sys
import time
import traceback

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from embree_amd import api, workloads as W                      # noqa: E402
from embree_amd.rtypes import make_rayhits, rays_of, INVALID_ID, RAYHIT_DTYPE, RAY_DTYPE   # noqa: E402
from oracle import restate                                      # noqa: E402
from tests import bvh_check                                     # noqa: E402
from tests.helpers import compare_closest, compare_occluded     # noqa: E402


def log(*a):
    print(*a, flush=True)


def oracle_scene(meshes, masks=None):
    o = restate.OracleScene()
    for i, (v, t) in enumerate(meshes):
        o.add_mesh(v, t, 1 if masks is None else masks[i])
    o.commit()
    return o


def check_scene(dev, name, meshes, rays, masks=None, validate=True, **kw):
    log(f"--- {name}: {W.num_triangles(meshes)} tris, {rays.shape[0]} rays")
    t0 = time.time()
    s = api.make_scene(dev, meshes, masks, **kw)
    info = s.info()
    log(f"    commit {1e3 * (time.time() - t0):.1f} ms wall, build {info['build_ms']:.3f} ms GPU; {info}")
    if validate:
        nodes, tris = s.download_bvh()
        st = bvh_check.validate(nodes, tris, info["root_ref"], meshes, masks, max_leaf=info["max_leaf"])
        log(f"    BVH valid: {st}")
    o = oracle_scene(meshes, masks)
    want = rays.copy()
    o.intersect1(want)
    got = rays.copy()
    t0 = time.time()
    s.intersect1M(got)
    log(f"    rtcIntersect1M {1e3 * (time.time() - t0):.1f} ms wall")
    st = compare_closest(got, want, rays, o.triangle_t, max_tie_frac=0.02, label=name)   # symmetric cameras graze cube edges exactly
    log(f"    closest parity OK: {st}")
    r0 = rays_of(rays)
    wr, gr = r0.copy(), r0.copy()
    o.occluded1(wr)
    s.occluded1M(gr)
    st = compare_occluded(gr["tfar"], wr["tfar"], r0["tfar"], label=name)
    log(f"    occluded parity OK: {st}")
    d = api.DeviceArray.from_numpy(rays)
    stats = s.trace_stats(d.ptr, rays.shape[0], 96)
    log(f"    stats/ray: nodes {stats['nodes'] / rays.shape[0]:.2f} "
        f"tris {stats['tris'] / rays.shape[0]:.2f} spills {stats['spills']} maxdepth {stats['max_depth']}")
    got2 = d.download(RAYHIT_DTYPE)
    assert got2.tobytes() == got.tobytes(), "stats build of the kernel gives different results"
    d.free()
    s.release()
    return info


def stage_basic(dev):
    tv = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0]], np.float32)
    rays = make_rayhits([[0.33, 0.33, -1], [1.0, 1.0, -1], [0.1, 0.2, -3]], [[0, 0, 1], [0, 0, 1], [0, 0, 1]])
    check_scene(dev, "minimal (1 tri)", [(tv, np.array([[0, 1, 2]], np.uint32))], rays)
    check_scene(dev, "cube+plane cfg1", W.cube_and_plane(), W.cube_camera_rays())
    check_scene(dev, "cornell 64x64", W.cornell_box(), W.cornell_camera_rays(64, 64))


def stage_soup(dev):
    rng = np.random.default_rng(3)
    for n in (100, 1500, 40000):
        c = rng.random((n, 3), dtype=np.float32)
        v = (c[:, None, :] + (rng.random((n, 3, 3), dtype=np.float32) - 0.5) * 0.08).reshape(-1, 3)
        t = np.arange(3 * n, dtype=np.uint32).reshape(-1, 3)
        rays = W.incoherent_rays(20000, [0.5, 0.5, 0.5], seed=n)
        check_scene(dev, f"soup {n}", [(v, t)], rays)
    g = np.load(os.path.join(ROOT, "tests", "golden", "ref_soup_8k.npz"))
    check_scene(dev, "golden soup masks", [(g["v0"], g["t0"]), (g["v1"], g["t1"])], g["rays"], masks=[1, 2])


def stage_crown_small(dev):
    meshes = W.synthetic_crown(num_phi=24)
    prim = W.crown_camera_rays(meshes, 128, 128)
    check_scene(dev, "crown phi=24 primary", meshes, prim)
    o = oracle_scene(meshes)
    tr = prim.copy()
    o.intersect1(tr)
    bounce = W.diffuse_bounce_rays(tr, meshes)
    check_scene(dev, "crown phi=24 bounce", meshes, bounce, validate=False)
    check_scene(dev, "crown phi=24 bounce (device-resident geometry)", meshes, bounce, validate=False, device_resident=True)


def stage_perf(dev):
    meshes = W.synthetic_crown(num_phi=int(os.environ.get("PHI", "158")))
    log(f"--- perf: crown {W.num_triangles(meshes)} tris")
    t0 = time.time()
    s = api.make_scene(dev, meshes, device_resident=True)
    info = s.info()
    log(f"    commit wall {1e3 * (time.time() - t0):.1f} ms; GPU build {info['build_ms']:.2f} ms = "
        f"{info['num_triangles'] / info['build_ms'] / 1e3:.1f} Mprims/s; {info}")
    prim = W.crown_camera_rays(meshes, 1024, 1024)
    d = api.DeviceArray.from_numpy(prim)
    s.intersect1M_device(d.ptr, prim.shape[0])
    api.load().mi355_synchronize(None)
    tr = d.download(RAYHIT_DTYPE)
    log(f"    primary hit fraction {(tr['geomID'] != INVALID_ID).mean():.3f}")
    bounce = W.diffuse_bounce_rays(tr, meshes)
    for name, rays in (("primary", prim), ("bounce", bounce)):
        for rep in range(3):
            d.upload(rays)
            api.load().mi355_synchronize(None)
            t0 = time.time()
            s.intersect1M_device(d.ptr, rays.shape[0])
            api.load().mi355_synchronize(None)
            dt = time.time() - t0
            log(f"    {name} closest: {1e3 * dt:.3f} ms -> {rays.shape[0] / dt / 1e6:.1f} Mrays/s (host clock)")
        d.upload(rays)
        st = s.trace_stats(d.ptr, rays.shape[0], 96)
        log(f"    {name} stats/ray: nodes {st['nodes'] / rays.shape[0]:.2f} "
            f"tris {st['tris'] / rays.shape[0]:.2f} spills {st['spills']} maxdepth {st['max_depth']}")
    d.free()
    s.release()


STAGES = dict(basic=stage_basic, soup=stage_soup, crown=stage_crown_small, perf=stage_perf)

if __name__ == "__main__":
    names = sys.argv[1:] or ["basic", "soup", "crown", "perf"]
    dev = api.Device("verbose=1")
    log("device:", dev.name())
    failed = 0
    for n in names:
        try:
            t0 = time.time()
            STAGES[n](dev)
            log(f"=== stage {n} OK in {time.time() - t0:.1f}s")
        except Exception:
            failed += 1
            log(f"=== stage {n} FAILED")
            traceback.print_exc()
            sys.stdout.flush()
    sys.exit(1 if failed else 0)
