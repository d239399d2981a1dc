"""ctypes front-end of oracle/_ref/libref_driver.so -- TEST INFRASTRUCTURE.

Drives the REAL reference (Embree 4.4.1 built by oracle/ref.mk from the
sources under /root/reference).  Only tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg may import this module; the product never does.
"""
import ctypes
import os
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# two single-ISA builds of the reference (oracle/ref.mk): AVX2 (the default, what every parity test uses) and AVX-512 (ISA=avx512: the CPU baseline of bench.py
# reports both).  Different file names: two libraries of one name cannot live in one process.
_LIBS = {"avx2": os.path.join(_HERE, "_ref", "libref_driver.so"), "avx512": os.path.join(_HERE, "_ref", "avx512", "libref_driver_avx512.so")}
_LIB = _LIBS["avx2"]


def available(isa="avx2"):
    if not os.path.exists(_LIBS[isa]):
        return False
    if isa == "avx512":                                       # the library must also be able to RUN here
        try:
            flags = open("/proc/cpuinfo").read()
            return all(f in flags for f in ("avx512f", "avx512dq", "avx512bw", "avx512vl", "avx512cd"))
        except OSError:
            return False
    return True


_loaded = {}


def _load(isa="avx2"):
    if isa not in _loaded:
        L = ctypes.CDLL(_LIBS[isa])
        L.refd_new.restype = ctypes.c_void_p
        L.refd_new.argtypes = [ctypes.c_char_p]
        L.refd_set_flags.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
        L.refd_add_mesh.restype = ctypes.c_uint
        L.refd_add_mesh.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint,
                                    ctypes.c_void_p, ctypes.c_uint, ctypes.c_uint]
        L.refd_add_quads.restype = ctypes.c_uint
        L.refd_add_quads.argtypes = L.refd_add_mesh.argtypes
        L.refd_new_object.restype = ctypes.c_void_p
        L.refd_new_object.argtypes = [ctypes.c_void_p]
        L.refd_add_instance.restype = ctypes.c_uint
        L.refd_add_instance.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint]
        L.refd_commit.restype = ctypes.c_double
        L.refd_commit.argtypes = [ctypes.c_void_p]
        L.refd_error.restype = ctypes.c_int
        L.refd_error.argtypes = [ctypes.c_void_p]
        L.refd_bounds.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        for name in ("refd_intersect1", "refd_occluded1", "refd_intersect4", "refd_intersect8"):
            f = getattr(L, name)
            f.restype = ctypes.c_double
            f.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint, ctypes.c_int]
        L.refd_packet.restype = ctypes.c_double
        L.refd_packet.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_uint, ctypes.c_void_p, ctypes.c_int]
        L.refd_free.argtypes = [ctypes.c_void_p]
        L.refd_set_filters.argtypes = [ctypes.c_void_p, ctypes.c_uint, ctypes.c_uint]
        for f in (L.refd_intersect1_args, L.refd_occluded1_args):
            f.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint, ctypes.c_int, ctypes.c_int, ctypes.c_uint]
            f.restype = ctypes.c_double
        L.refd_hw_threads.restype = ctypes.c_uint
        L.refd_run_tiled.restype = ctypes.c_double
        L.refd_run_tiled.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint, ctypes.c_uint, ctypes.c_int, ctypes.c_int]
        L.refd_run_tiled_mode.restype = ctypes.c_double
        L.refd_run_tiled_mode.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint, ctypes.c_uint, ctypes.c_int, ctypes.c_int, ctypes.c_uint]
        L.refd_native16.restype = ctypes.c_int
        L.refd_native16.argtypes = [ctypes.c_void_p]
        _loaded[isa] = L
    return _loaded[isa]


def hw_threads():
    return int(_load().refd_hw_threads())


class RefScene:
    """One device + one scene of the real reference."""

    def __init__(self, cfg="", flags=0, quality=1, parent=None, isa="avx2"):
        self.isa = parent.isa if parent is not None else isa
        L = self._L = _load(self.isa)
        self._h = L.refd_new_object(parent._h) if parent is not None else L.refd_new(cfg.encode())
        if not self._h:
            raise RuntimeError("reference rtcNewDevice failed")
        if flags or quality != 1:
            L.refd_set_flags(self._h, flags, quality)
        self.commit_seconds = None

    def add_mesh(self, verts, tris, mask=1):
        v = np.ascontiguousarray(verts, np.float32).reshape(-1, 3)
        t = np.ascontiguousarray(tris, np.uint32).reshape(-1, 3)
        return self._L.refd_add_mesh(self._h, v.ctypes.data, v.shape[0], t.ctypes.data, t.shape[0], mask)

    def add_quads(self, verts, quads, mask=1):
        v = np.ascontiguousarray(verts, np.float32).reshape(-1, 3)
        q = np.ascontiguousarray(quads, np.uint32).reshape(-1, 4)
        return self._L.refd_add_quads(self._h, v.ctypes.data, v.shape[0], q.ctypes.data, q.shape[0], mask)

    def new_object(self, flags=0):
        """a second scene on this scene's device, to be instanced with add_instance (commit it first)"""
        return RefScene(flags=flags, parent=self)

    def add_instance(self, obj, local2world, mask=1):
        """RTC_GEOMETRY_TYPE_INSTANCE of `obj`; local2world = 12 floats, column major (vx, vy, vz, p)"""
        x = np.ascontiguousarray(local2world, np.float32).reshape(12)
        return self._L.refd_add_instance(self._h, obj._h, x.ctypes.data, mask)

    def commit(self):
        self.commit_seconds = self._L.refd_commit(self._h)
        return self.commit_seconds

    def bounds(self):
        b = np.zeros(8, np.float32)
        self._L.refd_bounds(self._h, b.ctypes.data)
        return b[0:3].copy(), b[4:7].copy()

    def _run(self, fn, arr, threads):
        assert arr.flags["C_CONTIGUOUS"]
        return getattr(self._L, fn)(self._h, arr.ctypes.data, arr.shape[0], threads)

    def intersect1(self, rayhits, threads=1):
        return self._run("refd_intersect1", rayhits, threads)

    def intersect4(self, rayhits, threads=1):
        return self._run("refd_intersect4", rayhits, threads)

    def intersect8(self, rayhits, threads=1):
        return self._run("refd_intersect8", rayhits, threads)

    def packet(self, K, rays, valid=None, any_hit=False, threads=1):
        """rtcIntersectK / rtcOccludedK (K = 4, 8, 16) of the real library over an AoS array (RTCRayHit, or RTCRay when any_hit);
        valid: optional int32 array, one flag per ray (0 = inactive lane)."""
        assert rays.flags["C_CONTIGUOUS"] and rays.dtype.itemsize == (48 if any_hit else 96)
        v = None
        if valid is not None:
            v = np.ascontiguousarray(valid, np.int32)
            assert v.shape[0] == rays.shape[0]
        dt = self._L.refd_packet(self._h, K, 1 if any_hit else 0, rays.ctypes.data, rays.shape[0], v.ctypes.data if v is not None else None, threads)
        assert dt >= 0.0, "unsupported packet size"
        return dt

    def run_tiled(self, rays, tiles, threads, any_hit=False, mode=0):
        """rtcIntersect1 / rtcOccluded1 over `tiles` copies of `rays` in a buffer the worker pool itself fills (first touch where it is traced); seconds of the traced pass"""
        assert rays.flags["C_CONTIGUOUS"] and rays.dtype.itemsize == (48 if any_hit else 96)
        dt = self._L.refd_run_tiled_mode(self._h, rays.ctypes.data, rays.shape[0], tiles, 1 if any_hit else 0, threads, mode)
        assert dt >= 0.0
        return dt

    def occluded1(self, rays, threads=1):
        return self._run("refd_occluded1", rays, threads)

    def set_filters(self, ngeom, mode):
        """the fixed filter rules of ref_driver.cpp on geometries 0..ngeom-1: bit 0 intersect filter, bit 1 occluded filter, bit 2 accept the argument filter"""
        self._L.refd_set_filters(self._h, ngeom, mode)

    def intersect1_args(self, rayhits, arg_rule=False, flags=0, threads=1):
        return self._L.refd_intersect1_args(self._h, rayhits.ctypes.data, rayhits.shape[0], threads, 1 if arg_rule else 0, flags)

    def occluded1_args(self, rays, arg_rule=False, flags=0, threads=1):
        return self._L.refd_occluded1_args(self._h, rays.ctypes.data, rays.shape[0], threads, 1 if arg_rule else 0, flags)

    def native16(self):
        """RTC_DEVICE_PROPERTY_NATIVE_RAY16_SUPPORTED of the library behind this scene: 1 for the AVX-512 build"""
        return int(self._L.refd_native16(self._h))

    def error(self):
        return self._L.refd_error(self._h)

    def close(self):
        if self._h:
            self._L.refd_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def build_stats(meshes, quality=1, flags=0, threads=4, cfg=""):
    """Builds the scene with the real reference at verbose=2 and parses what BVHN::postBuild / BVHNStatistics print (kernels/bvh/bvh.cpp:139-179,
    bvh_statistics.cpp:19-40): {'sah', 'sah_nodes', 'sah_leaves', 'nodes', 'depth', 'primitives', 'builder'}.  `sah` is the reference's own metric:
    (sum of inner-node half areas + sum of leaf half areas x Triangle4 blocks) / root half area.  quality: 0 LOW, 1 MEDIUM, 2 HIGH (RTCBuildQuality)."""
    import re
    import sys
    import tempfile
    sys.stdout.flush()
    tmp = tempfile.TemporaryFile()
    saved = os.dup(1)
    os.dup2(tmp.fileno(), 1)
    try:
        s = RefScene(("threads=%d,verbose=2," % threads) + cfg, flags=flags, quality=quality)
        for v, t in meshes:
            s.add_mesh(v, t)
        s.commit()
        s.close()
    finally:
        sys.stdout.flush()
        os.dup2(saved, 1)
        os.close(saved)
    tmp.seek(0)
    txt = tmp.read().decode(errors="replace")
    out = {}
    m = re.search(r"building (\S+) using (\S+)", txt)
    if m:
        out["builder"] = m.group(2)
    m = re.search(r"primitives = (\d+), vertices = \d+, depth = (\d+)", txt)
    if m:
        out["primitives"], out["depth"] = int(m.group(1)), int(m.group(2))
    for key, pat in (("sah", r"total\s+: sah =\s*([0-9.eE+-]+)"), ("sah_nodes", r"getAABBNodes\s+: sah =\s*([0-9.eE+-]+)"), ("sah_leaves", r"leaves\s+: sah =\s*([0-9.eE+-]+)")):
        m = re.search(pat, txt)
        if m:
            out[key] = float(m.group(1))
    m = re.search(r"total\s+: sah = .*?#nodes =\s*(\d+)", txt)
    if m:
        out["nodes"] = int(m.group(1))
    return out
