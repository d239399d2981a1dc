"""ctypes front-end of oracle/librestate.so (oracle/restate.c) -- TEST INFRASTRUCTURE.

The plain-C restatement of the reference's build + traversal + triangle test.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import it.
"""
import ctypes
import os
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "librestate.so")
_lib = None


def available():
    return os.path.exists(_LIB)


def _load():
    global _lib
    if _lib is None:
        L = ctypes.CDLL(_LIB)
        L.ora_new.restype = ctypes.c_void_p
        L.ora_free.argtypes = [ctypes.c_void_p]
        L.ora_add_mesh.restype = ctypes.c_uint
        L.ora_add_mesh.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint, ctypes.c_void_p,
                                   ctypes.c_uint, ctypes.c_uint]
        L.ora_add_quads.restype = ctypes.c_uint
        L.ora_add_quads.argtypes = L.ora_add_mesh.argtypes
        L.ora_commit.argtypes = [ctypes.c_void_p]
        L.ora_set_robust.argtypes = [ctypes.c_void_p, ctypes.c_int]
        L.ora_bounds.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        L.ora_counts.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        L.ora_visit_stats.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
        L.ora_intersect1.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint]
        L.ora_occluded1.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint]
        L.ora_triangle_t.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                     ctypes.c_void_p, ctypes.c_uint]
        L.ora_inst_new.restype = ctypes.c_void_p
        L.ora_inst_free.argtypes = [ctypes.c_void_p]
        L.ora_inst_add.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint, ctypes.c_uint]
        L.ora_inst_bounds.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        L.ora_inst_intersect1.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint]
        L.ora_inst_occluded1.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint]
        _lib = L
    return _lib


class OracleScene:
    """One scene of the reference = one BVH per geometry type, queried in turn (Scene::commit kernels/common/scene.cpp:777-779: triangle accel, then
    quad accel; AccelN::intersect kernels/common/acceln.cpp:44-50).  geomIDs are shared: a mesh occupies its id in both trees, empty in the other."""

    def __init__(self, robust=False):
        """robust=True restates RTC_SCENE_FLAG_ROBUST: Triangle4v / Quad4v leaves, Pluecker test, conservative node test."""
        L = _load()
        self._h, self._hq = L.ora_new(), L.ora_new()
        self._has_quads = False
        self._hi, self._objs, self._next_id = None, [], 0          # instance accel (created with the first instance), instanced scenes kept alive
        if robust:
            L.ora_set_robust(self._h, 1)
            L.ora_set_robust(self._hq, 1)

    def add_mesh(self, verts, tris, mask=1):
        v = np.ascontiguousarray(verts, np.float32).reshape(-1, 3)
        t = np.ascontiguousarray(tris, np.uint32).reshape(-1, 3)
        L = _load()
        L.ora_add_quads(self._hq, None, 0, None, 0, mask)
        self._next_id += 1
        return L.ora_add_mesh(self._h, v.ctypes.data, v.shape[0], t.ctypes.data, t.shape[0], mask)

    def add_quads(self, verts, quads, mask=1):
        v = np.ascontiguousarray(verts, np.float32).reshape(-1, 3)
        q = np.ascontiguousarray(quads, np.uint32).reshape(-1, 4)
        L = _load()
        self._has_quads = True
        L.ora_add_mesh(self._h, None, 0, None, 0, mask)
        self._next_id += 1
        return L.ora_add_quads(self._hq, v.ctypes.data, v.shape[0], q.ctypes.data, q.shape[0], mask)

    def add_instance(self, obj, local2world, mask=1):
        """RTC_GEOMETRY_TYPE_INSTANCE of the committed OracleScene `obj` (its triangle and quad accels; instances inside `obj` are dropped like the
        reference drops a second level, instance_stack.h:36-47); local2world = 12 floats column major (vx, vy, vz, p).  Takes the next geometry id."""
        L = _load()
        if self._hi is None:
            self._hi = L.ora_inst_new()
        x = np.ascontiguousarray(local2world, np.float32).reshape(12)
        gid = self._next_id
        self._next_id += 1
        L.ora_add_mesh(self._h, None, 0, None, 0, mask)           # the id stays empty in the other accels
        L.ora_add_quads(self._hq, None, 0, None, 0, mask)
        L.ora_inst_add(self._hi, obj._h, obj._hq if obj._has_quads else None, x.ctypes.data, mask, gid)
        self._objs.append(obj)
        return gid

    def commit(self):
        _load().ora_commit(self._h)
        _load().ora_commit(self._hq)

    def bounds(self):
        b, c = np.zeros(6, np.float32), np.zeros(6, np.float32)
        _load().ora_bounds(self._h, b.ctypes.data)
        _load().ora_bounds(self._hq, c.ctypes.data)
        lo, hi = np.minimum(b[:3], c[:3]), np.maximum(b[3:], c[3:])
        if self._hi is not None:
            _load().ora_inst_bounds(self._hi, c.ctypes.data)
            lo, hi = np.minimum(lo, c[:3]), np.maximum(hi, c[3:])
        return lo, hi

    def counts(self):
        c = np.zeros(4, np.uint64)
        _load().ora_counts(self._h, c.ctypes.data)
        return dict(prims=int(c[0]), nodes=int(c[1]), blocks=int(c[2]), root_inner=bool(c[3]))

    def visit_stats(self, reset=True):
        c = np.zeros(3, np.uint64)
        _load().ora_visit_stats(self._h, c.ctypes.data, int(reset))
        return dict(nodes=int(c[0]), leaves=int(c[1]), blocks=int(c[2]))

    def intersect1(self, rayhits):
        assert rayhits.flags["C_CONTIGUOUS"] and rayhits.dtype.itemsize == 96
        _load().ora_intersect1(self._h, rayhits.ctypes.data, rayhits.shape[0])
        if self._has_quads:
            _load().ora_intersect1(self._hq, rayhits.ctypes.data, rayhits.shape[0])
        if self._hi is not None:                                   # the instance accel comes last (Scene::commit, kernels/common/scene.cpp:777-790)
            _load().ora_inst_intersect1(self._hi, rayhits.ctypes.data, rayhits.shape[0])

    def occluded1(self, rays):
        assert rays.flags["C_CONTIGUOUS"] and rays.dtype.itemsize == 48
        _load().ora_occluded1(self._h, rays.ctypes.data, rays.shape[0])
        if self._has_quads:
            _load().ora_occluded1(self._hq, rays.ctypes.data, rays.shape[0])
        if self._hi is not None:
            _load().ora_inst_occluded1(self._hi, rays.ctypes.data, rays.shape[0])

    def triangle_t(self, rayhits_in, geomID, primID):
        """t of ray i against the single primitive (geomID[i], primID[i]) (a quad: the nearer of its two triangles); NaN if not hit.
        `rayhits_in` must carry the ORIGINAL tnear/tfar (not a traced result)."""
        g = np.ascontiguousarray(geomID, np.uint32)
        p = np.ascontiguousarray(primID, np.uint32)
        out = np.zeros(rayhits_in.shape[0], np.float32)
        _load().ora_triangle_t(self._h, rayhits_in.ctypes.data, g.ctypes.data, p.ctypes.data, out.ctypes.data, rayhits_in.shape[0])
        if self._has_quads:
            out2 = np.zeros_like(out)
            _load().ora_triangle_t(self._hq, rayhits_in.ctypes.data, g.ctypes.data, p.ctypes.data, out2.ctypes.data, rayhits_in.shape[0])
            out = np.where(np.isnan(out), out2, out)
        return out

    def close(self):
        for h in (getattr(self, "_h", None), getattr(self, "_hq", None)):
            if h:
                _load().ora_free(h)
        self._h = self._hq = None
        if getattr(self, "_hi", None):
            _load().ora_inst_free(self._hi)
            self._hi = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
