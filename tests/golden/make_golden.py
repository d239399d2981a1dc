"""Generates the committed fixtures under tests/golden/ from the reference.

Run HERE (the container that mounts /root/reference and has oracle/_ref built by
`make -f oracle/ref.mk`); the fixtures travel to the GPU box, this script's inputs do not.

  python tests/golden/make_golden.py

Outputs
  cornell_box.npz        verts/tris of tutorials/models/cornell_box.obj ('v' and 'f' records only,
                         negative indices resolved, quads fan-triangulated (0,1,2),(0,2,3)) -> 34 tris
  ref_cube_1k.npz        config 1: the real reference's rtcIntersect1/rtcOccluded1 results for the 32x32
                         camera rays on cube+plane (tutorials/triangle_geometry)
  ref_cornell_4k.npz     config 2 at 64x64: real-reference results on the Cornell box
  ref_soup_8k.npz        8192 incoherent rays on a seeded 3,000-triangle soup (two geometries,
                         masks 1 and 2, ray masks alternating) incl. occluded results
  ref_trianglehit.npz    TriangleHitTest (tutorials/verify/verify.cpp:2462-2547) inputs + real-reference outputs
  ref_quads.npz          a noisy 40x40 quad grid (RTC_GEOMETRY_TYPE_QUAD, mask 3) + a triangle sphere, 26,000 rays, fast and robust scenes
  ref_instances.npz      RTC_GEOMETRY_TYPE_INSTANCE: a ground plane and a sphere as the scene's own geometry + 24 instances (random rotation, non-uniform
                         scale, shear, translation; geometry masks 1/2/3) of two object scenes (a noisy sphere; a cube + a 4x4 quad patch), 40,000 rays
                         with masks, fast and robust scenes: rtcIntersect1 (incl. instID / instPrimID) and rtcOccluded1 results
  ref_watertight_robust.npz  WatertightTest (verify.cpp:3611-3688) at its position (148376, 1234, -223423): triangle sphere of
                         radius 2 (numPhi 50), 8192 rays from inside, real reference with RTC_SCENE_FLAG_ROBUST
                         (BVH8Triangle4v + Pluecker + conservative node test); rtcIntersect1 and rtcOccluded1 results
"""
import os
import sys
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from embree_amd import workloads as W            # noqa: E402
from embree_amd.rtypes import make_rayhits, rays_of  # noqa: E402
from oracle import refembree                      # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"


def parse_obj(path):
    verts, tris = [], []
    for line in open(path):
        p = line.split()
        if not p:
            continue
        if p[0] == "v":
            verts.append([float(x) for x in p[1:4]])
        elif p[0] == "f":
            idx = []
            for tok in p[1:]:
                i = int(tok.split("/")[0])
                idx.append(i - 1 if i > 0 else len(verts) + i)
            for k in range(1, len(idx) - 1):
                tris.append([idx[0], idx[k], idx[k + 1]])
    return np.array(verts, np.float32), np.array(tris, np.uint32)


def trace_ref(meshes, rayhits, masks=None, occl=True, flags=0):
    s = refembree.RefScene("threads=1", flags=flags)
    for i, (v, t) in enumerate(meshes):
        s.add_mesh(v, t, 1 if masks is None else masks[i])
    s.commit()
    rh = rayhits.copy()
    s.intersect1(rh)
    out = dict(rays=rayhits, hits=rh)
    if occl:
        r = rays_of(rayhits)
        s.occluded1(r)
        out["occluded_tfar"] = r["tfar"].copy()
    lo, hi = s.bounds()
    out["bounds_lo"], out["bounds_hi"] = lo, hi
    s.close()
    return out


def soup(n, seed):
    rng = np.random.default_rng(seed)
    c = rng.random((n, 3), dtype=np.float32)
    v = (c[:, None, :] + (rng.random((n, 3, 3), dtype=np.float32) - 0.5) * 0.08).reshape(-1, 3)
    t = np.arange(3 * n, dtype=np.uint32).reshape(-1, 3)
    return v.astype(np.float32), t


def instances_fixture():
    rng = np.random.default_rng(2024)
    ground = (np.array([[-8, -3, -8], [8, -3, -8], [8, -3, 8], [-8, -3, 8]], np.float32), np.array([[0, 1, 2], [0, 2, 3]], np.uint32))
    own_sphere = W.triangle_sphere([0.0, 0.0, 0.0], 0.8, 10)
    obj_a = W.triangle_sphere([0.1, -0.2, 0.05], 1.0, 16, noise=0.2, seed=9)
    cube = W.cube_and_plane()[0]
    k = 4
    gy, gx = np.meshgrid(np.arange(k + 1, dtype=np.float32), np.arange(k + 1, dtype=np.float32), indexing="ij")
    qv = np.stack([gx / k * 2 - 1, 1.3 + 0.1 * rng.standard_normal(gx.shape).astype(np.float32), gy / k * 2 - 1], -1).reshape(-1, 3).astype(np.float32)
    ii = (np.arange(k)[:, None] * (k + 1) + np.arange(k)[None, :]).ravel()
    qq = np.stack([ii, ii + 1, ii + k + 2, ii + k + 1], -1).astype(np.uint32)
    n_inst = 24
    xf = np.zeros((n_inst, 12), np.float32)
    for i in range(n_inst):
        q, _ = np.linalg.qr(rng.standard_normal((3, 3)))
        sc = np.diag(rng.uniform(0.3, 1.2, 3))
        sh = np.eye(3); sh[0, 1] = rng.uniform(-0.3, 0.3)
        m = q @ sh @ sc
        xf[i, :9] = m.T.reshape(9).astype(np.float32)             # column major: vx, vy, vz
        xf[i, 9:] = rng.uniform(-4.5, 4.5, 3).astype(np.float32) * np.array([1, 0.5, 1], np.float32)
    imask = np.array([1 + (i % 3) for i in range(n_inst)], np.uint32)
    which = np.array([i % 2 for i in range(n_inst)], np.uint32)        # 0: object A, 1: object B
    rays = np.concatenate([W.incoherent_rays(24000, [0.0, 0.5, 0.0], seed=21), W.incoherent_rays(8000, [3.0, 1.0, -2.0], seed=22)])
    org = rng.uniform(-9, 9, (8000, 3)).astype(np.float32); org[:, 1] = np.abs(org[:, 1]) * 0.5 + 2.0
    tgt = (xf[rng.integers(0, n_inst, 8000), 9:] + rng.uniform(-0.6, 0.6, (8000, 3))).astype(np.float32)       # aimed at the instances
    rays = np.concatenate([rays, make_rayhits(org, tgt - org)])
    rays["id"] = np.arange(rays.shape[0], dtype=np.uint32)
    rays["mask"] = np.where(np.arange(rays.shape[0]) % 4 == 0, 1, np.where(np.arange(rays.shape[0]) % 4 == 1, 2, 0xFFFFFFFF)).astype(np.uint32)
    out = dict(ground_v=ground[0], ground_t=ground[1], sphere_v=own_sphere[0], sphere_t=own_sphere[1], a_v=obj_a[0], a_t=obj_a[1],
               cube_v=cube[0], cube_t=cube[1], qv=qv, qq=qq, xfm=xf, inst_mask=imask, inst_obj=which, rays=rays)
    for fl, suffix in ((0, ""), (4, "_robust")):
        top = refembree.RefScene("threads=1", flags=fl)
        oa, ob = top.new_object(fl), top.new_object(fl)
        if fl:
            refembree._load().refd_set_flags(oa._h, fl, 1); refembree._load().refd_set_flags(ob._h, fl, 1)
        oa.add_mesh(*obj_a); oa.commit()
        ob.add_mesh(*cube); ob.add_quads(qv, qq); ob.commit()
        assert top.add_mesh(*ground) == 0 and top.add_mesh(*own_sphere) == 1
        for i in range(n_inst):
            assert top.add_instance(ob if which[i] else oa, xf[i], int(imask[i])) == 2 + i
        top.commit()
        rh = rays.copy(); top.intersect1(rh)
        rr = rays_of(rays); top.occluded1(rr)
        out["hits" + suffix], out["occl" + suffix] = rh, rr["tfar"].copy()
        out["bounds_lo"], out["bounds_hi"] = top.bounds()
        assert top.error() == 0
        inst_hits = (rh["instID"] != 0xFFFFFFFF).sum()
        print("instances%s: %d hits, %d through instances, %d occluded" % (suffix, (rh["geomID"] != 0xFFFFFFFF).sum(), inst_hits, np.isneginf(rr["tfar"]).sum()))
        assert inst_hits > 6000, inst_hits
        top.close(); oa.close(); ob.close()
    return out


def main():
    v, t = parse_obj(os.path.join(REF, "tutorials/models/cornell_box.obj"))
    assert t.shape[0] == 34, t.shape
    np.savez_compressed(os.path.join(OUT, "cornell_box.npz"), verts=v, tris=t)

    np.savez_compressed(os.path.join(OUT, "ref_cube_1k.npz"), **trace_ref(W.cube_and_plane(), W.cube_camera_rays()))
    np.savez_compressed(os.path.join(OUT, "ref_cornell_4k.npz"), **trace_ref(W.cornell_box(), W.cornell_camera_rays(64, 64)))

    a, b = soup(2000, 11), soup(1000, 12)
    rays = W.incoherent_rays(8192, [0.5, 0.5, 0.5], seed=5)
    rays["mask"] = np.where(np.arange(8192) % 3 == 0, 1, np.where(np.arange(8192) % 3 == 1, 2, 3)).astype(np.uint32)
    d = trace_ref([a, b], rays, masks=[1, 2])
    d.update(v0=a[0], t0=a[1], v1=b[0], t1=b[1])
    np.savez_compressed(os.path.join(OUT, "ref_soup_8k.npz"), **d)

    # TriangleHitTest: one triangle (0,0,0),(1,0,0),(0,1,0); rays from (0,0,-1) to (u,v,0)
    tv = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0]], np.float32)
    tt = np.array([[0, 1, 2]], np.uint32)
    uu, vv = np.meshgrid(np.arange(16, dtype=np.float32), np.arange(16, dtype=np.float32))
    u = (0.01 + 0.9 * uu.ravel() / 16).astype(np.float32)
    w = (0.01 + 0.9 * vv.ravel() / 16).astype(np.float32) * (1 - u)
    tgt = np.stack([u, w, np.zeros_like(u)], -1)
    org = np.tile(np.array([[0, 0, -1]], np.float32), (256, 1))
    d = trace_ref([(tv, tt)], make_rayhits(org, tgt - org))
    d.update(u0=u, v0=w)
    np.savez_compressed(os.path.join(OUT, "ref_trianglehit.npz"), **d)
    # WatertightTest: rays from inside a sphere far away from the origin; every ray must hit (robust scenes only)
    pos = np.array([148376.0, 1234.0, -223423.0], np.float32)
    sph = W.triangle_sphere(pos, 2.0, 50)
    rng = np.random.default_rng(77)
    org = (pos[None, :] + (2.0 * rng.random((8192, 3), dtype=np.float32) - 1.0)).astype(np.float32)
    dirs = (2.0 * rng.random((8192, 3), dtype=np.float32) - 1.0).astype(np.float32)
    d = trace_ref([sph], make_rayhits(org, dirs), flags=4)          # RTC_SCENE_FLAG_ROBUST
    assert (d["hits"]["geomID"] == 0).all() and np.isneginf(d["occluded_tfar"]).all(), "the reference itself is not watertight here"
    np.savez_compressed(os.path.join(OUT, "ref_watertight_robust.npz"), **d)
    # quads next to triangles (RTC_GEOMETRY_TYPE_QUAD, non-planar quads, geometry mask 3), fast and robust
    k = 40
    gy, gx = np.meshgrid(np.arange(k + 1, dtype=np.float32), np.arange(k + 1, dtype=np.float32), indexing="ij")
    rq = np.random.default_rng(1)
    qv = np.stack([gx / k, gy / k, 0.5 + 0.05 * rq.standard_normal(gx.shape).astype(np.float32)], -1).reshape(-1, 3).astype(np.float32)
    ii = (np.arange(k)[:, None] * (k + 1) + np.arange(k)[None, :]).ravel()
    qq = np.stack([ii, ii + 1, ii + k + 2, ii + k + 1], -1).astype(np.uint32)
    tv, tt = W.triangle_sphere([0.5, 0.5, 0.2], 0.3, 12)
    rays = np.concatenate([W.incoherent_rays(20000, [0.5, 0.5, 1.2], seed=1), W.incoherent_rays(6000, [0.5, 0.5, 0.25], seed=2)])
    out = dict(tv=tv, tt=tt, qv=qv, qq=qq, rays=rays)
    for fl, suffix in ((0, ""), (4, "_robust")):
        sc = refembree.RefScene("threads=1", flags=fl)
        sc.add_mesh(tv, tt)
        sc.add_quads(qv, qq, 3)
        sc.commit()
        rh = rays.copy()
        sc.intersect1(rh)
        rr = rays_of(rays)
        sc.occluded1(rr)
        out["hits" + suffix], out["occl" + suffix] = rh, rr["tfar"].copy()
        out["bounds_lo"], out["bounds_hi"] = sc.bounds()
        sc.close()
    np.savez_compressed(os.path.join(OUT, "ref_quads.npz"), **out)
    np.savez_compressed(os.path.join(OUT, "ref_instances.npz"), **instances_fixture())
    print("golden fixtures written to", OUT)


if __name__ == "__main__":
    main()
