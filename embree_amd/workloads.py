"""Deterministic workload definitions (scenes + ray sets) for the BASELINE.json configs.

Workloads are *inputs*, shared by the parity tests, bench.py and smoke(); nothing in
here traces a ray.  Every generator is seeded and vectorised with numpy so that the
GPU box regenerates bit-identical inputs (no file travels except the 34-triangle
Cornell box fixture).

Reference workload definitions these follow (paths relative to the reference root):
  * cube + ground plane      tutorials/triangle_geometry/triangle_geometry_device.cpp:16-95
  * pinhole camera           tutorials/common/tutorial/camera.h (ISPCCamera, xfm.l.vz = -0.5*W*vx + 0.5*H*vy + 0.5*H/tan(fov/2)*vz)
  * Cornell box              tutorials/models/cornell_box.obj / cornell_box.ecs:3 (tests/golden/cornell_box.npz)
  * tessellated sphere       tutorials/common/scenegraph/geometry_creation.cpp:121-190 (createTriangleSphere)
  * RandomSampler            tutorials/common/math/random_sampler.h:15-115 (Murmur3 seed + LCG)
  * cosine hemisphere sample tutorials/common/math/sampling.h:52-78
  * incoherent ray benchmark tutorials/verify/verify.cpp:5923-6063, rtcore_helpers.h:200-223
crown.ecs / powerplant.ecs are not shipped with the reference; the
synthetic stand-ins below are sized to the same triangle counts.
"""
import os
import numpy as np

from .rtypes import make_rayhits, rays_of, INVALID_ID

_HERE = os.path.dirname(os.path.abspath(__file__))
_GOLDEN = os.path.join(os.path.dirname(_HERE), "tests", "golden")


# ----------------------------------------------------------------------------- RNG
def _u32(x):
    return np.asarray(x).astype(np.uint32)


def _murmur_mix(h, k):
    k = _u32(k) * np.uint32(0xCC9E2D51)
    k = (k << np.uint32(15)) | (k >> np.uint32(17))
    k = k * np.uint32(0x1B873593)
    h = h ^ k
    h = ((h << np.uint32(13)) | (h >> np.uint32(19))) * np.uint32(5) + np.uint32(0xE6546B64)
    return h


def _murmur_fin(h):
    h = h ^ (h >> np.uint32(16))
    h = h * np.uint32(0x85EBCA6B)
    h = h ^ (h >> np.uint32(13))
    h = h * np.uint32(0xC2B2AE35)
    h = h ^ (h >> np.uint32(16))
    return h


class RandomSampler:
    """Vectorised RandomSampler: one independent stream per element of `ids`."""

    def __init__(self, ids, sample=None):
        with np.errstate(over="ignore"):
            h = _murmur_mix(np.zeros(np.shape(ids), np.uint32), ids)
            if sample is not None:
                h = _murmur_mix(h, np.full(np.shape(ids), sample, np.uint32))
            self.s = _murmur_fin(h)

    def get_uint(self):
        with np.errstate(over="ignore"):
            self.s = self.s * np.uint32(1664525) + np.uint32(1013904223)
        return self.s

    def get_float(self):
        return (self.get_uint() >> np.uint32(1)).astype(np.float32) * np.float32(4.656612873077392578125e-10)

    def get_3d(self):
        return np.stack([self.get_float(), self.get_float(), self.get_float()], axis=-1)


def _hash_noise(n, seed):
    """n floats in [0,1) from the element index, seed-dependent (used to displace vertices)."""
    return RandomSampler(np.arange(n, dtype=np.uint32), seed).get_float()


# -------------------------------------------------------------------------- scenes
def cube_and_plane():
    """Config 1 scene: unit cube (8 verts / 12 tris, geomID 0) + ground plane (4 / 2, geomID 1)."""
    cv = np.array([[-1, -1, -1], [-1, -1, 1], [-1, 1, -1], [-1, 1, 1],
                   [1, -1, -1], [1, -1, 1], [1, 1, -1], [1, 1, 1]], np.float32)
    ct = np.array([[0, 1, 2], [1, 3, 2], [4, 6, 5], [5, 6, 7], [0, 4, 1], [1, 4, 5],
                   [2, 3, 6], [3, 7, 6], [0, 2, 4], [2, 6, 4], [1, 5, 3], [3, 5, 7]], np.uint32)
    pv = np.array([[-10, -2, -10], [-10, -2, 10], [10, -2, -10], [10, -2, 10]], np.float32)
    pt = np.array([[0, 1, 2], [1, 3, 2]], np.uint32)
    return [(cv, ct), (pv, pt)]


def cornell_box():
    """Config 2 scene: the reference's Cornell box, 17 quads fan-triangulated to 34 triangles,
    as ONE mesh (same choice in every run).  Fixture made by tests/golden/make_golden.py."""
    d = np.load(os.path.join(_GOLDEN, "cornell_box.npz"))
    return [(d["verts"].astype(np.float32), d["tris"].astype(np.uint32))]


def triangle_sphere(center, radius, num_phi, noise=0.0, seed=0):
    """Lat-long sphere with the reference's createTriangleSphere topology:
    numTheta = 2*numPhi, (numPhi+1)*numTheta vertices, 2*numTheta*(numPhi-1) triangles.
    `noise` scales each vertex radius by (1 + noise*(hash-0.5)*2)."""
    num_theta = 2 * num_phi
    phi = np.arange(num_phi + 1, dtype=np.float32)[:, None] * np.float32(np.pi / num_phi)
    theta = np.arange(num_theta, dtype=np.float32)[None, :] * np.float32(2.0 * np.pi / num_theta)
    r = np.full((num_phi + 1, num_theta), radius, np.float32)
    if noise:
        h = _hash_noise((num_phi + 1) * num_theta, seed).reshape(num_phi + 1, num_theta)
        # poles share one position per ring in the reference; keep rings watertight at the poles
        h[0, :] = h[0, 0]
        h[-1, :] = h[-1, 0]
        r = r * (1.0 + noise * (2.0 * h - 1.0)).astype(np.float32)
    sp, cp, st, ct = np.sin(phi), np.cos(phi), np.sin(theta), np.cos(theta)
    x = center[0] + r * sp * st
    y = center[1] + r * cp * np.ones_like(st)
    z = center[2] + r * sp * ct
    verts = np.stack([x, y, z], -1).reshape(-1, 3).astype(np.float32)
    th = np.arange(1, num_theta + 1, dtype=np.int64)
    tris = []
    # phi == 1 ring (reference: Triangle(p10, p00, p11) with p00 = numTheta-1)
    p00 = np.full_like(th, num_theta - 1)
    p10 = 1 * num_theta + th - 1
    p11 = 1 * num_theta + th % num_theta
    tris.append(np.stack([p10, p00, p11], -1))
    for ph in range(2, num_phi):
        q00 = (ph - 1) * num_theta + th - 1
        q01 = (ph - 1) * num_theta + th % num_theta
        q10 = ph * num_theta + th - 1
        q11 = ph * num_theta + th % num_theta
        tris.append(np.stack([q10, q00, q11], -1))
        tris.append(np.stack([q01, q11, q00], -1))
    b00 = (num_phi - 1) * num_theta + th - 1
    b01 = (num_phi - 1) * num_theta + th % num_theta
    b10 = np.full_like(th, num_phi * num_theta)
    tris.append(np.stack([b10, b00, b01], -1))
    return verts, np.concatenate(tris).astype(np.uint32)


def _box_room(lo, hi):
    """5 inward-facing walls + floor as 2 triangles each (a Cornell-style enclosure)."""
    lo = np.asarray(lo, np.float32)
    hi = np.asarray(hi, np.float32)
    c = np.array([[lo[0], lo[1], lo[2]], [hi[0], lo[1], lo[2]], [hi[0], hi[1], lo[2]], [lo[0], hi[1], lo[2]],
                  [lo[0], lo[1], hi[2]], [hi[0], lo[1], hi[2]], [hi[0], hi[1], hi[2]], [lo[0], hi[1], hi[2]]], np.float32)
    q = [(0, 1, 5, 4), (3, 7, 6, 2), (0, 4, 7, 3), (1, 2, 6, 5), (4, 5, 6, 7), (0, 3, 2, 1)]
    t = []
    for a, b, cc, d in q:
        t += [(a, b, cc), (a, cc, d)]
    return c, np.array(t, np.uint32)


def synthetic_crown(num_phi=158, spheres=(4, 4, 3), seed=0xC0FFEE, noise=0.15):
    """crown.ecs stand-in (config 3/4): 48 noisy spheres (99,224 triangles each at num_phi=158) on a
    jittered 4x4x3 lattice inside a closed 12-triangle room -> 4,762,764 triangles, one mesh
    per sphere + one for the room (49 geometries).  num_phi scales the triangle count
    quadratically (tests use small values)."""
    nx, ny, nz = spheres
    n = nx * ny * nz
    rs = RandomSampler(np.arange(n, dtype=np.uint32), seed)
    jitter = rs.get_3d()
    rad = 0.30 + 0.12 * rs.get_float()
    meshes = []
    k = 0
    for ix in range(nx):
        for iy in range(ny):
            for iz in range(nz):
                c = np.array([ix + 0.5, iy + 0.5, iz + 0.5], np.float32) + (jitter[k] - 0.5) * 0.3
                meshes.append(triangle_sphere(c, float(rad[k]), num_phi, noise=noise, seed=seed + 17 * k + 1))
                k += 1
    meshes.append(_box_room([-0.25, -0.25, -0.25], [nx + 0.25, ny + 0.25, nz + 0.25]))
    return meshes


def synthetic_powerplant(target_tris=12_700_000, seed=0xB011E2):
    """powerplant.ecs stand-in (config 5): axis-aligned boxes (12 tris) and 600-triangle cylinders
    ("pipes") on a plant-like lattice with extreme size variance (long thin triangles).
    Returns a single mesh; the exact triangle count is len(tris)."""
    seg = 150                                # 150 segments * 4 triangles (2 side + 2 caps) = 600 / pipe
    n_pipes = int(target_tris * 0.75) // (seg * 4)
    n_boxes = (target_tris - n_pipes * seg * 4) // 12
    rs = RandomSampler(np.arange(n_pipes, dtype=np.uint32), seed)
    base = rs.get_3d() * np.array([200.0, 30.0, 200.0], np.float32)
    base[:, 0] = np.round(base[:, 0] / 2.0) * 2.0            # pipes run on a 2-unit lattice
    base[:, 2] = np.round(base[:, 2] / 2.0) * 2.0
    axis = (rs.get_uint() % np.uint32(3)).astype(np.int64)
    length = (0.5 + 40.0 * rs.get_float() ** 3).astype(np.float32)   # mostly short, a few very long
    radius = (0.02 + 0.3 * rs.get_float() ** 2).astype(np.float32)
    ang = np.arange(seg, dtype=np.float32) * np.float32(2 * np.pi / seg)
    ca, sa = np.cos(ang), np.sin(ang)
    # vertices: ring0 (seg), ring1 (seg), cap centres (2)
    ring = np.zeros((n_pipes, 2 * seg + 2, 3), np.float32)
    u = np.stack([(axis + 1) % 3, (axis + 2) % 3], -1)
    idx = np.arange(n_pipes)
    for r_i in range(2):
        sl = slice(r_i * seg, (r_i + 1) * seg)
        ring[:, sl, :] = base[:, None, :]
        ring[idx[:, None], np.arange(seg)[None, :] + r_i * seg, u[:, 0:1]] += radius[:, None] * ca[None, :]
        ring[idx[:, None], np.arange(seg)[None, :] + r_i * seg, u[:, 1:2]] += radius[:, None] * sa[None, :]
        ring[idx[:, None], np.arange(seg)[None, :] + r_i * seg, axis[:, None]] += r_i * length[:, None]
    ring[:, 2 * seg, :] = base
    ring[:, 2 * seg + 1, :] = base
    ring[idx, 2 * seg + 1, axis] += length
    j = np.arange(seg, dtype=np.int64)
    jn = (j + 1) % seg
    local = np.concatenate([np.stack([j, jn, j + seg], -1), np.stack([jn, jn + seg, j + seg], -1),
                            np.stack([np.full(seg, 2 * seg), jn, j], -1),
                            np.stack([np.full(seg, 2 * seg + 1), j + seg, jn + seg], -1)])
    vp = ring.reshape(-1, 3)
    tp = (local[None, :, :] + (np.arange(n_pipes, dtype=np.int64) * (2 * seg + 2))[:, None, None]).reshape(-1, 3)
    # boxes
    rb = RandomSampler(np.arange(n_boxes, dtype=np.uint32), seed + 1)
    bc = rb.get_3d() * np.array([200.0, 30.0, 200.0], np.float32)
    bs = (0.05 + 3.0 * rb.get_3d() ** 4).astype(np.float32)
    corners = np.array([[-1, -1, -1], [-1, -1, 1], [-1, 1, -1], [-1, 1, 1],
                        [1, -1, -1], [1, -1, 1], [1, 1, -1], [1, 1, 1]], np.float32)
    vb = (bc[:, None, :] + corners[None, :, :] * bs[:, None, :]).reshape(-1, 3)
    ct = cube_and_plane()[0][1].astype(np.int64)
    tb = (ct[None, :, :] + (np.arange(n_boxes, dtype=np.int64) * 8)[:, None, None]).reshape(-1, 3) + vp.shape[0]
    verts = np.concatenate([vp, vb]).astype(np.float32)
    tris = np.concatenate([tp, tb]).astype(np.uint32)
    return [(verts, tris)]


def scene_bounds(meshes):
    lo = np.min([m[0].min(0) for m in meshes], 0)
    hi = np.max([m[0].max(0) for m in meshes], 0)
    return lo.astype(np.float32), hi.astype(np.float32)


def num_triangles(meshes):
    return int(sum(m[1].shape[0] for m in meshes))


# ---------------------------------------------------------------------------- rays
def _normalize(v):
    return (v / np.sqrt((v * v).sum(-1, keepdims=True))).astype(np.float32)


def camera_rays(frm, to, up, fov_deg, width, height):
    """Pinhole primary rays in pixel order (y major), as the tutorials' renderPixel does:
    dir = normalize(x*vx + y*vy + vz), tnear 0, tfar inf, mask -1."""
    frm, to, up = (np.asarray(a, np.float32) for a in (frm, to, up))
    vz = _normalize(to - frm)
    vx = _normalize(np.cross(up, vz))         # camera2world lookat (camera.h)
    vy = _normalize(np.cross(vz, vx))
    fovscale = np.float32(1.0 / np.tan(np.deg2rad(0.5 * fov_deg)))
    l_vz = -0.5 * width * vx + 0.5 * height * vy + 0.5 * height * fovscale * vz
    xs, ys = np.meshgrid(np.arange(width, dtype=np.float32), np.arange(height, dtype=np.float32))
    d = xs[..., None] * vx[None, None, :] - ys[..., None] * vy[None, None, :] + l_vz[None, None, :]
    d = _normalize(d.reshape(-1, 3))
    org = np.broadcast_to(frm, d.shape).copy()
    return make_rayhits(org, d)


def cube_camera_rays(width=32, height=32):
    """Config 1: 1024 rays from (1.5,1.5,-1.5) to the origin (triangle_geometry.cpp:23-24)."""
    return camera_rays([1.5, 1.5, -1.5], [0, 0, 0], [0, 1, 0], 90.0, width, height)


def cornell_camera_rays(width=1024, height=1024):
    """Config 2: -vp 278 273 -800 -vi 278 273 0 -vu 0 1 0 -fov 37 (cornell_box.ecs:3)."""
    return camera_rays([278, 273, -800], [278, 273, 0], [0, 1, 0], 37.0, width, height)


def crown_camera_rays(meshes, width=1024, height=1024):
    lo, hi = scene_bounds(meshes)
    c = 0.5 * (lo + hi)
    frm = np.array([c[0], c[1], lo[2] + 0.05], np.float32)
    return camera_rays(frm, c, [0, 1, 0], 75.0, width, height)


def incoherent_rays(n, center, seed=0):
    """fastMakeRay-style rays (rtcore_helpers.h:200-223): org = centre, dir = 2*rand3 - 1."""
    rs = RandomSampler(np.arange(n, dtype=np.uint32), seed)
    d = (2.0 * rs.get_3d() - 1.0).astype(np.float32)
    org = np.broadcast_to(np.asarray(center, np.float32), d.shape).copy()
    return make_rayhits(org, d)


def _frame(n):
    """Orthonormal frame around unit normals (frame(N), tutorials/common/math/linearspace3.h)."""
    dx0 = np.cross(np.array([1, 0, 0], np.float32), n)
    dx1 = np.cross(np.array([0, 1, 0], np.float32), n)
    use0 = ((dx0 * dx0).sum(-1) > (dx1 * dx1).sum(-1))[:, None]
    dx = _normalize(np.where(use0, dx0, dx1))
    dy = _normalize(np.cross(n, dx))
    return dx, dy


def diffuse_bounce_rays(primary_traced, meshes, seed=1):
    """Config 3 ray set.  `primary_traced` = primary RTCRayHit array AFTER tracing (any correct
    tracer: oracle in tests, the HIP path in bench.py).  Hit -> cosine-weighted bounce from
    P + eps*N; miss -> incoherent ray from the scene centre.  Ray order = pixel order."""
    p = primary_traced
    n = p.shape[0]
    lo, hi = scene_bounds(meshes)
    s = np.float32(np.sqrt(((hi - lo) ** 2).sum()))
    hit = p["geomID"] != INVALID_ID
    org = np.stack([p["org_x"], p["org_y"], p["org_z"]], -1)
    d = np.stack([p["dir_x"], p["dir_y"], p["dir_z"]], -1)
    ng = np.stack([p["Ng_x"], p["Ng_y"], p["Ng_z"]], -1)
    ng = np.where(hit[:, None], ng, np.array([0, 1, 0], np.float32))
    nn = _normalize(ng)
    nn = np.where(((nn * d).sum(-1) > 0)[:, None], -nn, nn)              # faceforward
    t = np.where(hit, p["tfar"], np.float32(0)).astype(np.float32)
    P = org + t[:, None] * d
    rs = RandomSampler(np.arange(n, dtype=np.uint32), seed)
    u1, u2 = rs.get_float(), rs.get_float()
    phi = np.float32(2 * np.pi) * u1                                      # cosineSampleHemisphere
    cos_t = np.sqrt(u2)
    sin_t = np.sqrt(np.maximum(np.float32(0), 1 - u2))
    dx, dy = _frame(nn)
    bd = (np.cos(phi) * sin_t)[:, None] * dx + (np.sin(phi) * sin_t)[:, None] * dy + cos_t[:, None] * nn
    borg = P + np.float32(1e-3) * s * np.float32(1e-3) * nn
    inc = incoherent_rays(n, 0.5 * (lo + hi), seed + 7)
    io = np.stack([inc["org_x"], inc["org_y"], inc["org_z"]], -1)
    idr = np.stack([inc["dir_x"], inc["dir_y"], inc["dir_z"]], -1)
    o = np.where(hit[:, None], borg, io).astype(np.float32)
    dd = np.where(hit[:, None], bd, idr).astype(np.float32)
    return make_rayhits(o, dd, tnear=np.float32(1e-4) * s * np.float32(1e-3))


def shadow_rays(bounce_traced, meshes, samples=16, seed=3, first=0):
    """Config 4 ray set: `samples` rays per hit point of the traced bounce rays toward points on a
    1x1 area light under the ceiling; tfar = dist*(1-1e-4).  Returns RTCRay records.
    `first` = global index of the first shadow ray (a rank that generates only its shard of the 16 Mi rays passes the start of
    its range, so that the shards of N ranks are exactly the rays one rank would generate)."""
    b = bounce_traced
    lo, hi = scene_bounds(meshes)
    s = np.float32(np.sqrt(((hi - lo) ** 2).sum()))
    c = 0.5 * (lo + hi)
    hit = b["geomID"] != INVALID_ID
    org = np.stack([b["org_x"], b["org_y"], b["org_z"]], -1)
    d = np.stack([b["dir_x"], b["dir_y"], b["dir_z"]], -1)
    t = np.where(hit, b["tfar"], np.float32(1.0)).astype(np.float32)
    P = (org + t[:, None] * d).astype(np.float32)
    n = P.shape[0] * samples
    rs = RandomSampler(np.arange(first, first + n, dtype=np.uint32), seed)
    lx, lz = rs.get_float() - 0.5, rs.get_float() - 0.5
    L = np.stack([c[0] + lx, np.full(n, hi[1] - 0.26 * 1.0, np.float32), c[2] + lz], -1).astype(np.float32)
    Pr = np.repeat(P, samples, axis=0)
    v = L - Pr
    dist = np.sqrt((v * v).sum(-1)).astype(np.float32)
    dirn = (v / np.maximum(dist, np.float32(1e-20))[:, None]).astype(np.float32)
    rh = make_rayhits(Pr, dirn, tnear=np.float32(1e-4) * s * np.float32(1e-3), tfar=dist * np.float32(1 - 1e-4))
    rh["id"] = np.arange(first, first + n, dtype=np.uint32)          # RTCRay.id = the ray's index in the whole job, not in the shard
    return rays_of(rh)
