"""Thin ctypes binding of libembree4_mi355.so -- the same C ABI an Embree application links.

The Python layer only forwards to the C entry points of include/embree4/rtcore.h (names,
argument meaning and error behaviour are the reference's) and of include/embree_amd_hip.h.
There is no Python/CPU implementation of any part of the path: if the HIP library cannot be
loaded or no GPU is present, calls fail loudly.
"""
import ctypes as C
import os

import numpy as np

from .rtypes import RAY_DTYPE, RAYHIT_DTYPE

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libembree4_mi355.so")

RTC_ERROR_NONE, RTC_ERROR_UNKNOWN, RTC_ERROR_INVALID_ARGUMENT, RTC_ERROR_INVALID_OPERATION = 0, 1, 2, 3
RTC_ERROR_OUT_OF_MEMORY, RTC_ERROR_UNSUPPORTED_CPU, RTC_ERROR_CANCELLED = 4, 5, 6
RTC_GEOMETRY_TYPE_TRIANGLE, RTC_GEOMETRY_TYPE_QUAD = 0, 1
RTC_BUFFER_TYPE_INDEX, RTC_BUFFER_TYPE_VERTEX, RTC_BUFFER_TYPE_VERTEX_ATTRIBUTE = 0, 1, 2
RTC_FORMAT_UINT3, RTC_FORMAT_UINT4, RTC_FORMAT_FLOAT3 = 0x5003, 0x5004, 0x9003
RTC_SCENE_FLAG_ROBUST = 4
RTC_BUILD_QUALITY_LOW, RTC_BUILD_QUALITY_MEDIUM, RTC_BUILD_QUALITY_HIGH, RTC_BUILD_QUALITY_REFIT = 0, 1, 2, 3
RTC_SCENE_FLAG_DYNAMIC = 1
RTC_GEOMETRY_TYPE_INSTANCE = 121
RTC_FORMAT_FLOAT3X4_ROW_MAJOR, RTC_FORMAT_FLOAT3X4_COLUMN_MAJOR, RTC_FORMAT_FLOAT4X4_COLUMN_MAJOR = 0x9134, 0x9234, 0x9244

# every symbol include/embree4/rtcore.h and include/embree_amd_hip.h declare (checked by tests/test_abi.py)
RTC_SYMBOLS = """rtcNewDevice rtcRetainDevice rtcReleaseDevice rtcGetDeviceProperty rtcSetDeviceProperty rtcGetErrorString
rtcGetDeviceError rtcGetDeviceLastErrorMessage rtcSetDeviceErrorFunction rtcSetDeviceMemoryMonitorFunction
rtcNewBuffer rtcNewSharedBuffer rtcGetBufferData rtcRetainBuffer rtcReleaseBuffer
rtcNewGeometry rtcRetainGeometry rtcReleaseGeometry rtcCommitGeometry rtcEnableGeometry rtcDisableGeometry
rtcSetGeometryTimeStepCount rtcSetGeometryVertexAttributeCount rtcSetGeometryMask rtcSetGeometryBuildQuality
rtcSetGeometryInstancedScene rtcSetGeometryTransform rtcGetGeometryTransform rtcInterpolate rtcInterpolateN
rtcSetGeometryBuffer rtcSetSharedGeometryBuffer rtcSetSharedGeometryBufferHostDevice rtcSetNewGeometryBuffer
rtcGetGeometryBufferData rtcUpdateGeometryBuffer rtcSetGeometryUserData rtcGetGeometryUserData
rtcSetGeometryIntersectFilterFunction rtcSetGeometryOccludedFilterFunction rtcSetGeometryEnableFilterFunctionFromArguments rtcSetGeometryFilterRule
rtcNewScene rtcGetSceneDevice rtcRetainScene rtcReleaseScene rtcGetSceneTraversable rtcAttachGeometry
rtcAttachGeometryByID rtcDetachGeometry rtcGetGeometry rtcGetGeometryThreadSafe rtcCommitScene rtcJoinCommitScene
rtcSetSceneProgressMonitorFunction rtcSetSceneBuildQuality rtcSetSceneFlags rtcGetSceneFlags rtcGetSceneBounds
rtcIntersect1 rtcIntersect4 rtcIntersect8 rtcIntersect16 rtcOccluded1 rtcOccluded4 rtcOccluded8 rtcOccluded16 rtcIntersect1MDeviceSharded rtcOccluded1MDeviceSharded
rtcTraversableIntersect1 rtcTraversableIntersect4 rtcTraversableIntersect8 rtcTraversableIntersect16
rtcTraversableOccluded1 rtcTraversableOccluded4 rtcTraversableOccluded8 rtcTraversableOccluded16
rtcIntersect1M rtcOccluded1M rtcIntersect1MDevice rtcOccluded1MDevice""".split()
MI355_SYMBOLS = """mi355_default_build_params mi355_last_error mi355_device_count mi355_device_name mi355_bvh_build
mi355_bvh_destroy mi355_bvh_build_instanced mi355_bvh_refit mi355_bvh_refit_instanced mi355_release_build_scratch mi355_bvh_get_info mi355_bvh_set_filter_rules mi355_bvh_download mi355_trace_prepare mi355_trace_closest mi355_trace_any
mi355_trace_query mi355_trace_closest_packet mi355_trace_any_packet mi355_trace_stats mi355_trace_timed mi355_trace_status mi355_malloc mi355_malloc_retry mi355_free mi355_memcpy_h2d
mi355_memcpy_d2h mi355_synchronize mi355_device_synchronize mi355_memcpy_d2d_async mi355_stream_create
mi355_stream_destroy mi355_event_create mi355_event_record mi355_event_elapsed_ms mi355_event_destroy
mi355_comm_unique_id mi355_comm_init mi355_comm_destroy mi355_comm_allgather mi355_comm_gather mi355_pack_hits mi355_pack_hits_inst mi355_pack_occluded mi355_unpack_rays mi355_stream_query mi355_stream_wait_event mi355_measure_bandwidth mi355_measure_host_link mi355_trace_query_filtered mi355_sort_keys63""".split()


class FilterArguments(C.Structure):            # RTCFilterFunctionNArguments
    _fields_ = [("valid", C.POINTER(C.c_int)), ("geometryUserPtr", C.c_void_p), ("context", C.c_void_p), ("ray", C.POINTER(C.c_float)),
                ("hit", C.POINTER(C.c_float)), ("N", C.c_uint)]


FILTER_FN = C.CFUNCTYPE(None, C.POINTER(FilterArguments))
RTC_RAY_QUERY_FLAG_INVOKE_ARGUMENT_FILTER = 2
RTC_RAY_QUERY_FLAG_COHERENT = 1 << 16
RTC_DEVICE_PROPERTY_GPU_COUNT = 143


class QueryArguments(C.Structure):             # RTCIntersectArguments / RTCOccludedArguments (same layout)
    _fields_ = [("flags", C.c_int), ("feature_mask", C.c_int), ("context", C.c_void_p), ("filter", C.c_void_p), ("callback", C.c_void_p)]

    def __init__(self, filter_fn=None, flags=0):
        super().__init__()
        self.flags, self.feature_mask, self.context, self.callback = flags, -1, None, None
        self._fn = filter_fn
        self.filter = C.cast(filter_fn, C.c_void_p) if filter_fn else None


class FilterRule(C.Structure):                  # struct RTCFilterRule (include/embree4/rtcore.h)
    _fields_ = [("kinds", C.c_uint), ("apply", C.c_uint), ("modulus", C.c_uint), ("remainder", C.c_uint), ("primFactor", C.c_uint), ("geomFactor", C.c_uint),
                ("tmin", C.c_float), ("tmax", C.c_float), ("umax", C.c_float), ("vmax", C.c_float), ("bits", C.c_void_p), ("numBits", C.c_uint)]


RTC_FILTER_RULE_MODULO, RTC_FILTER_RULE_PRIMITIVE_BITS, RTC_FILTER_RULE_DISTANCE_WINDOW, RTC_FILTER_RULE_UV_CUTOFF = 1, 2, 4, 8
RTC_FILTER_RULE_APPLY_INTERSECT, RTC_FILTER_RULE_APPLY_OCCLUDED = 1, 2


class BuildParams(C.Structure):
    _fields_ = [("sah_block_shift", C.c_uint32), ("min_leaf", C.c_uint32), ("max_leaf", C.c_uint32),
                ("small_threshold", C.c_uint32), ("trav_cost", C.c_float), ("int_cost", C.c_float),
                ("robust", C.c_uint32), ("quality", C.c_uint32), ("split_factor", C.c_float), ("refit", C.c_uint32),
                ("presplits", C.c_uint32), ("top_splits", C.c_uint32), ("top_split_min", C.c_uint32), ("top_split_rel", C.c_float), ("top_split_cell", C.c_float)]


class BvhInfo(C.Structure):
    _fields_ = [("num_triangles", C.c_uint64), ("num_nodes", C.c_uint64), ("num_leaves", C.c_uint64),
                ("num_binary_nodes", C.c_uint64), ("bytes_nodes", C.c_uint64), ("bytes_triangles", C.c_uint64),
                ("bounds_lower", C.c_float * 3), ("bounds_upper", C.c_float * 3), ("sah", C.c_float),
                ("build_ms", C.c_float), ("root_ref", C.c_uint32), ("top_levels", C.c_uint32),
                ("max_leaf", C.c_uint32), ("depth", C.c_uint32),
                ("bytes_refit", C.c_uint64), ("num_refits", C.c_uint32), ("num_presplit", C.c_uint32),
                ("num_launches", C.c_uint32), ("num_host_syncs", C.c_uint32), ("build_attempts", C.c_uint32), ("reserved0", C.c_uint32)]

    def as_dict(self):
        d = {k: getattr(self, k) for k, _ in self._fields_ if not k.startswith("bounds")}
        d["bounds_lower"] = list(self.bounds_lower)
        d["bounds_upper"] = list(self.bounds_upper)
        return d


class RTCInterpolateArguments(C.Structure):
    _fields_ = [("geometry", C.c_void_p), ("primID", C.c_uint32), ("u", C.c_float), ("v", C.c_float), ("bufferType", C.c_int), ("bufferSlot", C.c_uint32),
                ("P", C.c_void_p), ("dPdu", C.c_void_p), ("dPdv", C.c_void_p), ("ddPdudu", C.c_void_p), ("ddPdvdv", C.c_void_p), ("ddPdudv", C.c_void_p),
                ("valueCount", C.c_uint32)]


class RTCInterpolateNArguments(C.Structure):
    _fields_ = [("geometry", C.c_void_p), ("valid", C.c_void_p), ("primIDs", C.c_void_p), ("u", C.c_void_p), ("v", C.c_void_p), ("N", C.c_uint32),
                ("bufferType", C.c_int), ("bufferSlot", C.c_uint32),
                ("P", C.c_void_p), ("dPdu", C.c_void_p), ("dPdv", C.c_void_p), ("ddPdudu", C.c_void_p), ("ddPdvdv", C.c_void_p), ("ddPdudv", C.c_void_p),
                ("valueCount", C.c_uint32)]


class RTCBounds(C.Structure):
    _fields_ = [("lower_x", C.c_float), ("lower_y", C.c_float), ("lower_z", C.c_float), ("align0", C.c_float),
                ("upper_x", C.c_float), ("upper_y", C.c_float), ("upper_z", C.c_float), ("align1", C.c_float)]


NODE_DTYPE = np.dtype([("org", "<f4", (3,)), ("exp", "u1", (3,)), ("imask", "u1"), ("childBase", "<u4"), ("triBase", "<u4"),
                       ("meta", "u1", (8,)), ("qlo", "u1", (3, 8)), ("qhi", "u1", (3, 8))])
TRI_DTYPE = np.dtype([("v0", "<f4", (3,)), ("e1", "<f4", (3,)), ("e2", "<f4", (3,)), ("primID", "<u4"),
                      ("geomID", "<u4"), ("mask", "<u4")])
assert NODE_DTYPE.itemsize == 80 and TRI_DTYPE.itemsize == 48

_lib = None


def load():
    """Load the HIP library (building it first if the sources are newer). Raises if that is impossible."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        from . import build as _build
        _build.build()
    # RTLD_NOW (+ -z now at link time): every hip* symbol binds to the ROCm runtime this library was linked
    # against at load time, before any other package could bring a second HIP runtime into the process
    L = C.CDLL(os.environ.get("MI355_LIB", LIB_PATH), mode=os.RTLD_NOW | os.RTLD_LOCAL)   # MI355_LIB: A/B builds of the kernels (tools/build_variant.sh)
    vp, u32, sz = C.c_void_p, C.c_uint32, C.c_size_t
    L.rtcNewDevice.restype = vp
    L.rtcNewDevice.argtypes = [C.c_char_p]
    L.rtcReleaseDevice.argtypes = [vp]
    L.rtcGetDeviceError.restype = C.c_int
    L.rtcGetDeviceError.argtypes = [vp]
    L.rtcGetDeviceLastErrorMessage.restype = C.c_char_p
    L.rtcGetDeviceLastErrorMessage.argtypes = [vp]
    L.rtcGetDeviceProperty.restype = C.c_ssize_t
    L.rtcGetDeviceProperty.argtypes = [vp, C.c_int]
    L.rtcGetErrorString.restype = C.c_char_p
    L.rtcGetErrorString.argtypes = [C.c_int]
    L.rtcNewGeometry.restype = vp
    L.rtcNewGeometry.argtypes = [vp, C.c_int]
    for f in ("rtcReleaseGeometry", "rtcCommitGeometry", "rtcEnableGeometry", "rtcDisableGeometry", "rtcRetainGeometry"):
        getattr(L, f).argtypes = [vp]
    L.rtcSetGeometryMask.argtypes = [vp, u32]
    L.rtcSetGeometryBuildQuality.argtypes = [vp, C.c_int]
    L.rtcSetGeometryInstancedScene.argtypes = [vp, vp]
    L.rtcInterpolate.argtypes = [C.POINTER(RTCInterpolateArguments)]
    L.rtcInterpolateN.argtypes = [C.POINTER(RTCInterpolateNArguments)]
    L.rtcSetGeometryTransform.argtypes = [vp, u32, C.c_int, vp]
    L.rtcGetGeometryTransform.argtypes = [vp, C.c_float, C.c_int, vp]
    L.rtcSetGeometryTimeStepCount.argtypes = [vp, u32]
    L.rtcSetGeometryVertexAttributeCount.argtypes = [vp, u32]
    L.rtcSetSharedGeometryBuffer.argtypes = [vp, C.c_int, u32, C.c_int, vp, sz, sz, sz]
    L.rtcSetSharedGeometryBufferHostDevice.argtypes = [vp, C.c_int, u32, C.c_int, vp, vp, sz, sz, sz]
    L.rtcSetNewGeometryBuffer.restype = vp
    L.rtcSetNewGeometryBuffer.argtypes = [vp, C.c_int, u32, C.c_int, sz, sz]
    L.rtcGetGeometryBufferData.restype = vp
    L.rtcGetGeometryBufferData.argtypes = [vp, C.c_int, u32]
    L.rtcUpdateGeometryBuffer.argtypes = [vp, C.c_int, u32]
    L.rtcSetGeometryIntersectFilterFunction.argtypes = [vp, vp]
    L.rtcNewScene.restype = vp
    L.rtcNewScene.argtypes = [vp]
    for f in ("rtcReleaseScene", "rtcCommitScene", "rtcJoinCommitScene", "rtcRetainScene"):
        getattr(L, f).argtypes = [vp]
    L.rtcAttachGeometry.restype = u32
    L.rtcAttachGeometry.argtypes = [vp, vp]
    L.rtcAttachGeometryByID.argtypes = [vp, vp, u32]
    L.rtcDetachGeometry.argtypes = [vp, u32]
    L.rtcGetGeometry.restype = vp
    L.rtcGetGeometry.argtypes = [vp, u32]
    L.rtcSetSceneFlags.argtypes = [vp, C.c_int]
    L.rtcGetSceneFlags.restype = C.c_int
    L.rtcGetSceneFlags.argtypes = [vp]
    L.rtcSetSceneBuildQuality.argtypes = [vp, C.c_int]
    L.rtcGetSceneBounds.argtypes = [vp, C.POINTER(RTCBounds)]
    L.rtcIntersect1.argtypes = [vp, vp, vp]
    L.rtcOccluded1.argtypes = [vp, vp, vp]
    for k in (4, 8, 16):
        getattr(L, "rtcIntersect%d" % k).argtypes = [vp, vp, vp, vp]
        getattr(L, "rtcOccluded%d" % k).argtypes = [vp, vp, vp, vp]
    for fn in (L.rtcSetGeometryIntersectFilterFunction, L.rtcSetGeometryOccludedFilterFunction):
        fn.argtypes = [vp, vp]
    L.rtcSetGeometryEnableFilterFunctionFromArguments.argtypes = [vp, C.c_bool]
    L.rtcSetGeometryUserData.argtypes = [vp, vp]
    L.rtcGetGeometryUserData.argtypes = [vp]
    L.rtcGetGeometryUserData.restype = vp
    L.rtcSetGeometryFilterRule.argtypes = [vp, C.POINTER(FilterRule)]
    L.rtcIntersect1M.argtypes = [vp, vp, u32, sz, vp]
    L.rtcOccluded1M.argtypes = [vp, vp, u32, sz, vp]
    L.rtcIntersect1MDevice.argtypes = [vp, vp, u32, sz, vp, vp]
    L.rtcIntersect1MDeviceSharded.argtypes = [vp, u32, vp, vp, sz, vp, vp]
    L.rtcOccluded1MDeviceSharded.argtypes = [vp, u32, vp, vp, sz, vp, vp]
    L.rtcOccluded1MDevice.argtypes = [vp, vp, u32, sz, vp, vp]
    L.rtcGetSceneBVH_mi355.restype = vp
    L.rtcGetSceneBVH_mi355.argtypes = [vp]
    L.mi355_last_error.restype = C.c_char_p
    L.mi355_device_count.restype = C.c_int
    L.mi355_device_name.argtypes = [C.c_int, C.c_char_p, sz]
    L.mi355_bvh_get_info.argtypes = [vp, C.POINTER(BvhInfo)]
    L.mi355_bvh_download.argtypes = [vp, vp, sz, vp, sz]
    L.mi355_bvh_refit.argtypes = [vp, vp, u32, vp]
    L.mi355_trace_prepare.argtypes = [vp, vp]
    L.mi355_trace_closest.argtypes = [vp, vp, u32, sz, vp]
    L.mi355_trace_any.argtypes = [vp, vp, u32, sz, vp]
    L.mi355_trace_query.argtypes = [vp, vp, u32, sz, C.c_int, u32, vp]
    L.rtcGetSceneReplicaBVH_mi355.restype = vp
    L.rtcGetSceneReplicaBVH_mi355.argtypes = [vp, u32]
    L.mi355_trace_timed.argtypes = [vp, vp, u32, sz, C.c_int, vp, vp, vp]
    L.mi355_trace_stats.argtypes = [vp, vp, u32, sz, C.c_int, C.POINTER(C.c_uint64)]
    L.mi355_trace_status.argtypes = [vp, vp, C.POINTER(u32)]
    L.mi355_trace_closest_packet.argtypes = [vp, vp, vp, u32, u32, sz, vp]
    L.mi355_trace_any_packet.argtypes = [vp, vp, vp, u32, u32, sz, vp]
    L.mi355_malloc.argtypes = [C.c_int, sz, C.POINTER(vp)]
    L.mi355_free.argtypes = [vp]
    L.mi355_memcpy_h2d.argtypes = [vp, vp, sz]
    L.mi355_memcpy_d2h.argtypes = [vp, vp, sz]
    L.mi355_synchronize.argtypes = [vp]
    L.mi355_default_build_params.argtypes = [C.POINTER(BuildParams)]
    L.mi355_device_synchronize.argtypes = [C.c_int]
    L.mi355_memcpy_d2d_async.argtypes = [vp, vp, sz, vp]
    L.mi355_stream_create.argtypes = [C.c_int, C.POINTER(vp)]
    L.mi355_stream_destroy.argtypes = [vp]
    L.mi355_event_create.argtypes = [C.POINTER(vp)]
    L.mi355_event_record.argtypes = [vp, vp]
    L.mi355_event_elapsed_ms.argtypes = [vp, vp, C.POINTER(C.c_float)]
    L.mi355_event_destroy.argtypes = [vp]
    L.mi355_comm_unique_id.argtypes = [vp]
    L.mi355_comm_init.argtypes = [C.c_int, vp, C.c_int, C.c_int, C.POINTER(vp)]
    L.mi355_comm_destroy.argtypes = [vp]
    L.mi355_comm_allgather.argtypes = [vp, vp, vp, sz, vp]
    L.mi355_comm_gather.argtypes = [vp, vp, vp, sz, C.c_int, vp]
    L.mi355_pack_hits.argtypes = [vp, u32, sz, vp, vp]
    L.mi355_pack_occluded.argtypes = [vp, u32, sz, vp, vp]
    L.mi355_pack_hits_inst.argtypes = [vp, u32, sz, vp, vp]
    L.mi355_unpack_rays.argtypes = [vp, u32, vp, sz, C.c_int, vp]
    L.mi355_stream_wait_event.argtypes = [vp, vp]
    L.mi355_stream_query.argtypes = [vp]
    L.mi355_measure_bandwidth.argtypes = [C.c_int, sz, C.c_int, C.POINTER(C.c_double)]
    L.mi355_measure_host_link.argtypes = [C.c_int, sz, C.c_int, C.POINTER(C.c_double)]
    L.mi355_sort_keys63.argtypes = [C.c_int, vp, vp, vp, u32, C.POINTER(C.c_float)]
    _lib = L
    return L


class RTCErrorException(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("RTCError %d: %s" % (code, msg))
        self.code = code


class Device:
    """rtcNewDevice(config).  `check()` raises the pending per-thread error, like polling rtcGetDeviceError."""

    def __init__(self, config=""):
        self.L = load()
        self.gpu = 0
        for tok in config.replace(" ", ",").split(","):
            if tok.startswith("gpu="):
                self.gpu = int(tok[4:])
        self.h = self.L.rtcNewDevice(config.encode())
        if not self.h:
            code = self.L.rtcGetDeviceError(None)
            raise RTCErrorException(code, (self.L.rtcGetDeviceLastErrorMessage(None) or b"").decode())

    def get_error(self):
        return self.L.rtcGetDeviceError(self.h)

    def last_message(self):
        return (self.L.rtcGetDeviceLastErrorMessage(self.h) or b"").decode()

    def check(self):
        e = self.get_error()
        if e != RTC_ERROR_NONE:
            raise RTCErrorException(e, self.last_message())

    def gpu_count(self):
        return int(self.L.rtcGetDeviceProperty(self.h, RTC_DEVICE_PROPERTY_GPU_COUNT))

    def name(self):
        buf = C.create_string_buffer(256)
        self.L.mi355_device_name(self.gpu, buf, 256)
        return buf.value.decode()

    def release(self):
        if self.h:
            self.L.rtcReleaseDevice(self.h)
            self.h = None


class DeviceArray:
    """Raw HIP device allocation (mi355_malloc) with numpy upload/download."""

    def __init__(self, nbytes, gpu=0):
        self.L = load()
        p = C.c_void_p()
        if self.L.mi355_malloc(gpu, nbytes, C.byref(p)) != 0:
            raise RuntimeError("mi355_malloc: " + self.L.mi355_last_error().decode())
        self.ptr, self.nbytes = p.value, nbytes

    @classmethod
    def from_numpy(cls, a, gpu=0):
        a = np.ascontiguousarray(a)
        d = cls(a.nbytes, gpu)
        d.upload(a)
        return d

    def upload(self, a):
        a = np.ascontiguousarray(a)
        assert a.nbytes <= self.nbytes
        if self.L.mi355_memcpy_h2d(self.ptr, a.ctypes.data, a.nbytes) != 0:
            raise RuntimeError("mi355_memcpy_h2d: " + self.L.mi355_last_error().decode())

    def download(self, dtype, count=None):
        dt = np.dtype(dtype)
        n = self.nbytes // dt.itemsize if count is None else count
        out = np.empty(n, dt)
        if self.L.mi355_memcpy_d2h(out.ctypes.data, self.ptr, out.nbytes) != 0:
            raise RuntimeError("mi355_memcpy_d2h: " + self.L.mi355_last_error().decode())
        return out

    def free(self):
        if self.ptr:
            self.L.mi355_free(self.ptr)
            self.ptr = None


class Scene:
    """rtcNewScene + helpers that follow the tutorials' idiom (attach, release, commit)."""

    def __init__(self, device, flags=0, quality=None):
        self.dev, self.L = device, device.L
        self.h = self.L.rtcNewScene(device.h)
        device.check()
        if flags:
            self.L.rtcSetSceneFlags(self.h, flags)
        if quality is not None:
            self.L.rtcSetSceneBuildQuality(self.h, quality)
        self._keep = []

    def add_triangle_mesh(self, verts, tris, mask=None, shared=True, device_resident=False):
        """rtcNewGeometry(TRIANGLE) + vertex/index buffers + commit + attach + release -> geomID."""
        L = self.L
        v = np.ascontiguousarray(verts, np.float32).reshape(-1, 3)
        t = np.ascontiguousarray(tris, np.uint32).reshape(-1, 3)
        g = L.rtcNewGeometry(self.dev.h, RTC_GEOMETRY_TYPE_TRIANGLE)
        self.dev.check()
        if device_resident:
            dv = DeviceArray.from_numpy(np.concatenate([v.ravel(), np.zeros(4, np.float32)]), self.dev.gpu)
            dt = DeviceArray.from_numpy(t, self.dev.gpu)
            self._keep += [dv, dt]
            L.rtcSetSharedGeometryBufferHostDevice(g, RTC_BUFFER_TYPE_VERTEX, 0, RTC_FORMAT_FLOAT3, None, dv.ptr, 0, 12, v.shape[0])
            L.rtcSetSharedGeometryBufferHostDevice(g, RTC_BUFFER_TYPE_INDEX, 0, RTC_FORMAT_UINT3, None, dt.ptr, 0, 12, t.shape[0])
        elif shared:
            vp = np.concatenate([v.ravel(), np.zeros(4, np.float32)])      # 16-byte readable padding (README: shared buffers)
            self._keep += [vp, t]
            L.rtcSetSharedGeometryBuffer(g, RTC_BUFFER_TYPE_VERTEX, 0, RTC_FORMAT_FLOAT3, vp.ctypes.data, 0, 12, v.shape[0])
            L.rtcSetSharedGeometryBuffer(g, RTC_BUFFER_TYPE_INDEX, 0, RTC_FORMAT_UINT3, t.ctypes.data, 0, 12, t.shape[0])
        else:
            pv = L.rtcSetNewGeometryBuffer(g, RTC_BUFFER_TYPE_VERTEX, 0, RTC_FORMAT_FLOAT3, 12, v.shape[0])
            pt = L.rtcSetNewGeometryBuffer(g, RTC_BUFFER_TYPE_INDEX, 0, RTC_FORMAT_UINT3, 12, t.shape[0])
            self.dev.check()
            if v.size:
                C.memmove(pv, v.ctypes.data, v.nbytes)
            if t.size:
                C.memmove(pt, t.ctypes.data, t.nbytes)
        if mask is not None:
            L.rtcSetGeometryMask(g, mask)
        L.rtcCommitGeometry(g)
        gid = L.rtcAttachGeometry(self.h, g)
        L.rtcReleaseGeometry(g)
        self.dev.check()
        return gid

    def add_quad_mesh(self, verts, quads, mask=None, index_words=None, index_stride=16):
        """rtcNewGeometry(QUAD): float3 vertices, uint4 indices (shared host buffers).  `index_words` + `index_stride`: the index view as the application lays it
        out -- a flat uint32 array in which quad i is the four words at byte i * index_stride (12: consecutive quads overlap in one word, BufferStrideTest,
        tutorials/verify/verify.cpp:995-1008); `quads` then only gives the count."""
        L = self.L
        v = np.ascontiguousarray(verts, np.float32).reshape(-1, 3)
        q = np.ascontiguousarray(quads, np.uint32).reshape(-1, 4)
        if index_words is not None:
            w = np.ascontiguousarray(index_words, np.uint32).ravel()
            g = L.rtcNewGeometry(self.dev.h, RTC_GEOMETRY_TYPE_QUAD)
            self.dev.check()
            vp = np.concatenate([v.ravel(), np.zeros(4, np.float32)])
            self._keep += [vp, w]
            L.rtcSetSharedGeometryBuffer(g, RTC_BUFFER_TYPE_VERTEX, 0, RTC_FORMAT_FLOAT3, vp.ctypes.data, 0, 12, v.shape[0])
            L.rtcSetSharedGeometryBuffer(g, RTC_BUFFER_TYPE_INDEX, 0, RTC_FORMAT_UINT4, w.ctypes.data, 0, index_stride, q.shape[0])
            if mask is not None:
                L.rtcSetGeometryMask(g, mask)
            L.rtcCommitGeometry(g)
            gid = L.rtcAttachGeometry(self.h, g)
            L.rtcReleaseGeometry(g)
            self.dev.check()
            return gid
        g = L.rtcNewGeometry(self.dev.h, RTC_GEOMETRY_TYPE_QUAD)
        self.dev.check()
        vp = np.concatenate([v.ravel(), np.zeros(4, np.float32)])
        self._keep += [vp, q]
        L.rtcSetSharedGeometryBuffer(g, RTC_BUFFER_TYPE_VERTEX, 0, RTC_FORMAT_FLOAT3, vp.ctypes.data, 0, 12, v.shape[0])
        L.rtcSetSharedGeometryBuffer(g, RTC_BUFFER_TYPE_INDEX, 0, RTC_FORMAT_UINT4, q.ctypes.data, 0, 16, q.shape[0])
        if mask is not None:
            L.rtcSetGeometryMask(g, mask)
        L.rtcCommitGeometry(g)
        gid = L.rtcAttachGeometry(self.h, g)
        L.rtcReleaseGeometry(g)
        self.dev.check()
        return gid

    def add_instance(self, obj, local2world, mask=None, fmt=RTC_FORMAT_FLOAT3X4_COLUMN_MAJOR):
        """rtcNewGeometry(INSTANCE) + rtcSetGeometryInstancedScene + rtcSetGeometryTransform + commit + attach + release -> geomID
        (tutorials/instanced_geometry/instanced_geometry_device.cpp:145-160).  `obj` = a committed Scene of the same device;
        local2world in `fmt` (default: 12 floats column major = vx, vy, vz, p)."""
        L = self.L
        g = L.rtcNewGeometry(self.dev.h, RTC_GEOMETRY_TYPE_INSTANCE)
        self.dev.check()
        x = np.ascontiguousarray(local2world, np.float32).ravel()
        L.rtcSetGeometryInstancedScene(g, obj.h)
        L.rtcSetGeometryTimeStepCount(g, 1)
        L.rtcSetGeometryTransform(g, 0, fmt, x.ctypes.data)
        if mask is not None:
            L.rtcSetGeometryMask(g, mask)
        L.rtcCommitGeometry(g)
        gid = L.rtcAttachGeometry(self.h, g)
        L.rtcReleaseGeometry(g)
        self.dev.check()
        return gid

    def set_instance_transform(self, gid, local2world, fmt=RTC_FORMAT_FLOAT3X4_COLUMN_MAJOR):
        """rtcSetGeometryTransform + rtcCommitGeometry on an attached instance (the caller commits the scene): a MOVE -- the next rtcCommitScene refits the top tree"""
        g = self.L.rtcGetGeometry(self.h, gid)
        x = np.ascontiguousarray(local2world, np.float32).ravel()
        self.L.rtcSetGeometryTransform(g, 0, fmt, x.ctypes.data)
        self.L.rtcCommitGeometry(g)

    def set_geometry_build_quality(self, gid, quality):
        """rtcSetGeometryBuildQuality; RTC_BUILD_QUALITY_REFIT makes the next commit after a vertex update refit the tree."""
        self.L.rtcSetGeometryBuildQuality(self.L.rtcGetGeometry(self.h, gid), quality)
        self.dev.check()

    def update_vertices(self, gid, verts):
        """The dynamic-scene idiom of the tutorials: write the vertex buffer in place, rtcUpdateGeometryBuffer, rtcCommitGeometry
        (the caller commits the scene).  For meshes added with host buffers (shared or library-owned)."""
        L = self.L
        g = L.rtcGetGeometry(self.h, gid)
        v = np.ascontiguousarray(verts, np.float32).reshape(-1, 3)
        p = L.rtcGetGeometryBufferData(g, RTC_BUFFER_TYPE_VERTEX, 0)
        self.dev.check()
        C.memmove(p, v.ctypes.data, v.nbytes)
        L.rtcUpdateGeometryBuffer(g, RTC_BUFFER_TYPE_VERTEX, 0)
        L.rtcCommitGeometry(g)
        self.dev.check()

    def commit(self):
        self.L.rtcCommitScene(self.h)
        self.dev.check()

    def touch(self, gid=0):
        """Marks the scene modified the way an application does: rtcUpdateGeometryBuffer(INDEX) + rtcCommitGeometry on one geometry, so that the
        next rtcCommitScene builds again (a commit of an unmodified scene returns at once, kernels/common/scene.cpp:831).  Nothing is uploaded for
        device-resident shared buffers."""
        g = self.L.rtcGetGeometry(self.h, gid)
        self.dev.check()
        self.L.rtcUpdateGeometryBuffer(g, RTC_BUFFER_TYPE_INDEX, 0)
        self.L.rtcCommitGeometry(g)
        self.dev.check()

    def bounds(self):
        b = RTCBounds()
        self.L.rtcGetSceneBounds(self.h, C.byref(b))
        self.dev.check()
        return (np.array([b.lower_x, b.lower_y, b.lower_z], np.float32), np.array([b.upper_x, b.upper_y, b.upper_z], np.float32))

    def bvh(self):
        return self.L.rtcGetSceneBVH_mi355(self.h)

    def info(self):
        i = BvhInfo()
        self.L.mi355_bvh_get_info(self.bvh(), C.byref(i))
        return i.as_dict()

    def download_bvh(self):
        i = self.info()
        nodes = np.zeros(i["num_nodes"], NODE_DTYPE)
        tris = np.zeros(i["num_triangles"], TRI_DTYPE)
        rc = self.L.mi355_bvh_download(self.bvh(), nodes.ctypes.data, nodes.nbytes, tris.ctypes.data, tris.nbytes)
        if rc:
            raise RuntimeError(self.L.mi355_last_error().decode())
        return nodes, tris

    # -- queries on host arrays (numpy, modified in place) --
    def intersect1(self, rayhit_record):
        self.L.rtcIntersect1(self.h, rayhit_record.ctypes.data, None)
        self.dev.check()

    def occluded1(self, ray_record):
        self.L.rtcOccluded1(self.h, ray_record.ctypes.data, None)
        self.dev.check()

    def intersect1M(self, rayhits, args=None):
        """args: a QueryArguments (flags, filter) or None"""
        assert rayhits.dtype == RAYHIT_DTYPE and rayhits.flags["C_CONTIGUOUS"]
        self.L.rtcIntersect1M(self.h, rayhits.ctypes.data, rayhits.shape[0], 96, C.addressof(args) if args is not None else None)
        self.dev.check()

    def occluded1M(self, rays, args=None):
        assert rays.dtype == RAY_DTYPE and rays.flags["C_CONTIGUOUS"]
        self.L.rtcOccluded1M(self.h, rays.ctypes.data, rays.shape[0], 48, C.addressof(args) if args is not None else None)
        self.dev.check()

    # -- filter callbacks (host functions: the host-array entry points run them between launches) --
    def set_filters(self, gid, intersect=None, occluded=None, from_arguments=False):
        """intersect / occluded: FILTER_FN instances (keep them alive) or None"""
        g = self.L.rtcGetGeometry(self.h, gid)
        self._keep += [intersect, occluded]
        self.L.rtcSetGeometryIntersectFilterFunction(g, C.cast(intersect, C.c_void_p) if intersect else None)
        self.L.rtcSetGeometryOccludedFilterFunction(g, C.cast(occluded, C.c_void_p) if occluded else None)
        self.L.rtcSetGeometryEnableFilterFunctionFromArguments(g, bool(from_arguments))
        self.dev.check()

    def set_filter_rule(self, gid, rule):
        """rtcSetGeometryFilterRule on geometry `gid` (a FilterRule or None); takes effect with the next commit()"""
        g = self.L.rtcGetGeometry(self.h, gid)
        self.L.rtcSetGeometryFilterRule(g, C.byref(rule) if rule is not None else None)
        self.dev.check()

    # -- queries on device memory --
    def intersect1M_device(self, dptr, count, stride=96, stream=None, args=None):
        self.L.rtcIntersect1MDevice(self.h, dptr, count, stride, C.addressof(args) if args is not None else None, stream)
        self.dev.check()

    def occluded1M_device(self, dptr, count, stride=48, stream=None, args=None):
        self.L.rtcOccluded1MDevice(self.h, dptr, count, stride, C.addressof(args) if args is not None else None, stream)
        self.dev.check()

    def query_device_sharded(self, dptrs, counts, stride=96, streams=None, any_hit=False, args=None):
        """rtcIntersect1MDeviceSharded / rtcOccluded1MDeviceSharded: shard k (counts[k] records at device pointer dptrs[k], memory of replica k's GPU) is traced
        by replica k on streams[k]; nothing crosses xGMI"""
        n = len(dptrs)
        P = (C.c_void_p * n)(*[int(p) if p else None for p in dptrs])
        N = (C.c_uint32 * n)(*[int(c) for c in counts])
        S = (C.c_void_p * n)(*[(x.value if isinstance(x, C.c_void_p) else x) for x in streams]) if streams is not None else None
        fn = self.L.rtcOccluded1MDeviceSharded if any_hit else self.L.rtcIntersect1MDeviceSharded
        fn(self.h, n, P, N, stride, C.addressof(args) if args is not None else None, S)
        self.dev.check()

    def replica_bvh(self, k):
        """the tree on replica k of a device over several GPUs (rtcNewDevice("gpus=N")); None beyond the last"""
        return self.L.rtcGetSceneReplicaBVH_mi355(self.h, k)

    def trace_status(self, stream=None):
        """mi355_trace_status: synchronises `stream`, returns (and clears) the flags the traversal kernels raised on it (0 = no work was dropped)."""
        f = C.c_uint32(0)
        if self.L.mi355_trace_status(self.bvh(), stream, C.byref(f)) != 0:
            raise RuntimeError(self.L.mi355_last_error().decode())
        return f.value

    def trace_stats(self, dptr, count, stride, any_hit=False):
        out = (C.c_uint64 * 32)()
        rc = self.L.mi355_trace_stats(self.bvh(), dptr, count, stride, int(any_hit), out)
        if rc:
            raise RuntimeError(self.L.mi355_last_error().decode())
        return dict(nodes=out[0], tris=out[1], rays=out[2], spills=out[3], max_depth=out[4], wave_iters=out[5],
                    node_blocks=out[6], tri_blocks=out[7], lanes_idle=out[8], lanes_wait_batch=out[9],
                    lanes_wait_drain=out[10], lanes_blocked=out[11], empty_nodes=out[12], culled_groups=out[13],
                    refill_clocks=out[14], refill_events=out[15], loop_clocks=out[16], node_step_clocks=out[17],
                    unique_nodes=out[18], unique_tris=out[19])

    def release(self):
        if self.h:
            self.L.rtcReleaseScene(self.h)
            self.h = None
            for k in self._keep:
                if isinstance(k, DeviceArray):
                    k.free()
            self._keep = []


def make_scene(device, meshes, masks=None, flags=0, quality=None, **kw):
    s = Scene(device, flags, quality)
    for i, (v, t) in enumerate(meshes):
        s.add_triangle_mesh(v, t, None if masks is None else masks[i], **kw)
    s.commit()
    return s
