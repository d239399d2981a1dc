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
        L.ora_commit.argtypes = [ctypes.c_void_p]
        L.ora_set_robust.argtypes = [ctypes.c_void_p, ctypes.c_int]
        L.ora_bounds.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        L.ora_counts.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        L.ora_visit_stats.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
        L.ora_intersect1.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint]
        L.ora_occluded1.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint]
        L.ora_triangle_t.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                     ctypes.c_void_p, ctypes.c_uint]
        _lib = L
    return _lib


class OracleScene:
    def __init__(self, robust=False):
        """robust=True restates RTC_SCENE_FLAG_ROBUST: Triangle4v leaves, Pluecker test, conservative node test."""
        self._h = _load().ora_new()
        if robust:
            _load().ora_set_robust(self._h, 1)

    def add_mesh(self, verts, tris, mask=1):
        v = np.ascontiguousarray(verts, np.float32).reshape(-1, 3)
        t = np.ascontiguousarray(tris, np.uint32).reshape(-1, 3)
        return _load().ora_add_mesh(self._h, v.ctypes.data, v.shape[0], t.ctypes.data, t.shape[0], mask)

    def commit(self):
        _load().ora_commit(self._h)

    def bounds(self):
        b = np.zeros(6, np.float32)
        _load().ora_bounds(self._h, b.ctypes.data)
        return b[:3].copy(), b[3:].copy()

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

    def occluded1(self, rays):
        assert rays.flags["C_CONTIGUOUS"] and rays.dtype.itemsize == 48
        _load().ora_occluded1(self._h, rays.ctypes.data, rays.shape[0])

    def triangle_t(self, rayhits_in, geomID, primID):
        """t of ray i against the single triangle (geomID[i], primID[i]); NaN if not hit.
        `rayhits_in` must carry the ORIGINAL tnear/tfar (not a traced result)."""
        g = np.ascontiguousarray(geomID, np.uint32)
        p = np.ascontiguousarray(primID, np.uint32)
        out = np.zeros(rayhits_in.shape[0], np.float32)
        _load().ora_triangle_t(self._h, rayhits_in.ctypes.data, g.ctypes.data, p.ctypes.data, out.ctypes.data,
                               rayhits_in.shape[0])
        return out

    def close(self):
        if self._h:
            _load().ora_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
