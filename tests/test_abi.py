"""CPU-side checks of the drop-in boundary: the C-ABI library loads without a GPU and exports every symbol the
headers declare; struct layouts match the reference's default ABI; no compute calls here."""
import ctypes
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib_path():
    from embree_amd import build
    return build.build()


def _declared(header, macro):
    src = open(os.path.join(ROOT, "include", header)).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    src = "\n".join(l for l in src.splitlines() if not l.lstrip().startswith("#"))     # drop the macro definitions themselves
    return sorted(set(re.findall(macro + r"\s+[^;{(]*?\b(\w+)\s*\(", src)))


def test_library_exports_every_declared_symbol(lib_path):
    from embree_amd import api
    L = ctypes.CDLL(lib_path)
    rtc = _declared("embree4/rtcore.h", "RTC_API")
    mi = _declared("embree_amd_hip.h", "MI355_API")
    assert len(rtc) > 60 and len(mi) > 20
    for name in rtc + mi:
        assert hasattr(L, name), "library does not export " + name
    # the Python binding's lists are the same sets (so tests/bench bind exactly what the headers declare)
    assert sorted(api.RTC_SYMBOLS) == rtc
    assert sorted(api.MI355_SYMBOLS) == mi


def test_exports_are_c_abi_only(lib_path):
    out = subprocess.check_output(["nm", "-D", "--defined-only", lib_path]).decode()
    names = [l.split()[-1] for l in out.splitlines() if " T " in l]
    assert names and all(n.startswith(("rtc", "mi355_")) for n in names), [n for n in names if not n.startswith(("rtc", "mi355_"))][:5]
    needed = subprocess.check_output(["readelf", "-d", lib_path]).decode()
    assert "libamdhip64" in needed and "torch" not in needed


def test_struct_layouts_match_reference_default_abi(tmp_path):
    """sizeof/offsetof as the reference's default build: RTCRay 48, RTCHit 48, RTCRayHit 96, packets 336/672/1344."""
    src = tmp_path / "abi.c"
    src.write_text(r'''
#include <stdio.h>
#include <stddef.h>
#include <embree4/rtcore.h>
int main(void) {
  printf("%zu %zu %zu %zu %zu %zu ", sizeof(struct RTCRay), sizeof(struct RTCHit), sizeof(struct RTCRayHit),
         sizeof(struct RTCRayHit4), sizeof(struct RTCRayHit8), sizeof(struct RTCRayHit16));
  printf("%zu %zu %zu %zu %zu ", offsetof(struct RTCRay, tfar), offsetof(struct RTCRayHit, hit), offsetof(struct RTCHit, primID),
         offsetof(struct RTCHit, instID), sizeof(struct RTCBounds));
  printf("%zu %zu %d %d %d\n", sizeof(struct RTCIntersectArguments), _Alignof(struct RTCRayHit16), RTC_FORMAT_FLOAT3, RTC_FORMAT_UINT3, RTC_GEOMETRY_TYPE_TRIANGLE);
  return 0; }''')
    exe = tmp_path / "abi"
    subprocess.check_call(["gcc", "-std=c11", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])   # header is valid C
    vals = subprocess.check_output([str(exe)]).decode().split()
    assert vals == ["48", "48", "96", "336", "672", "1344", "32", "48", "20", "28", "32", "32", "64", str(0x9003), str(0x5003), "0"], vals
    from embree_amd.rtypes import RAY_DTYPE, RAYHIT_DTYPE
    assert RAY_DTYPE.fields["tfar"][1] == 32 and RAYHIT_DTYPE.fields["Ng_x"][1] == 48 and RAYHIT_DTYPE.fields["primID"][1] == 68


def test_kernel_abi_structs_match_their_ctypes_mirrors(tmp_path):
    """include/embree_amd_hip.h is valid C, and the structs the ctypes mirror (embree_amd/api.py) hands to it have the header's sizes and field offsets --
    mi355_bvh_info grew in round 4 (build_attempts): a mirror that lags behind would let the library write past the Python object."""
    src = tmp_path / "kabi.c"
    src.write_text(r'''
#include <stdio.h>
#include <stddef.h>
#include <embree_amd_hip.h>
int main(void) {
  printf("%zu %zu %zu %zu %zu %zu\n", sizeof(mi355_bvh_info), offsetof(mi355_bvh_info, build_ms), offsetof(mi355_bvh_info, bytes_refit),
         offsetof(mi355_bvh_info, num_launches), offsetof(mi355_bvh_info, build_attempts), sizeof(mi355_build_params));
  return 0; }''')
    exe = tmp_path / "kabi"
    subprocess.check_call(["gcc", "-std=c11", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    vals = [int(v) for v in subprocess.check_output([str(exe)]).decode().split()]
    from embree_amd import api
    I = api.BvhInfo
    assert vals[:5] == [C.sizeof(I), I.build_ms.offset, I.bytes_refit.offset, I.num_launches.offset, I.build_attempts.offset], (vals, C.sizeof(I))
    assert vals[5] == C.sizeof(api.BuildParams), (vals[5], C.sizeof(api.BuildParams))


def test_header_against_real_reference_header(tmp_path):
    """Where the reference is mounted (this container): every enum value / struct size we declare equals the
    reference's own header (compiled side by side)."""
    ref_inc = "/root/reference/include"
    gen = os.path.join(ROOT, "oracle", "_ref", "gen", "include", "embree4")
    if not (os.path.isdir(ref_inc) and os.path.exists(os.path.join(gen, "rtcore_config.h"))):
        pytest.skip("reference headers not available here")
    body = r'''
#include <stdio.h>
#include <stddef.h>
#include <embree4/rtcore.h>
int main(void) {
  printf("%zu %zu %zu %zu %zu %zu %zu %zu ", sizeof(struct RTCRay), sizeof(struct RTCHit), sizeof(struct RTCRayHit), sizeof(struct RTCRayHit4),
    sizeof(struct RTCRayHit8), sizeof(struct RTCRayHit16), sizeof(struct RTCIntersectArguments), sizeof(struct RTCOccludedArguments));
  printf("%d %d %d %d %d %d %d %d %d %d %d\n", RTC_FORMAT_FLOAT3, RTC_FORMAT_UINT3, RTC_FORMAT_FLOAT16, RTC_BUFFER_TYPE_VERTEX, RTC_GEOMETRY_TYPE_INSTANCE_ARRAY,
    RTC_ERROR_CANCELLED, RTC_SCENE_FLAG_ROBUST, RTC_DEVICE_PROPERTY_TASKING_SYSTEM, RTC_BUILD_QUALITY_REFIT, (int)RTC_FEATURE_FLAG_INSTANCE_ARRAY, RTC_RAY_QUERY_FLAG_COHERENT);
  return 0; }'''
    src = tmp_path / "hdr.cpp"
    src.write_text(body)
    outs = []
    for inc in ([os.path.join(ROOT, "include")], [ref_inc, gen]):
        exe = tmp_path / ("h%d" % len(outs))
        cmd = ["g++", "-std=c++17", str(src), "-o", str(exe)]
        for i in inc:
            cmd += ["-I", i]
        subprocess.check_call(cmd)
        outs.append(subprocess.check_output([str(exe)]).decode())
    assert outs[0] == outs[1], outs


def test_no_gpu_means_loud_failure(lib_path):
    """No CPU fallback: without a HIP device rtcNewDevice returns NULL and reports RTC_ERROR_UNSUPPORTED_CPU."""
    L = ctypes.CDLL(lib_path)
    L.mi355_device_count.restype = ctypes.c_int
    if L.mi355_device_count() > 0:
        pytest.skip("a GPU is present")
    L.rtcNewDevice.restype = ctypes.c_void_p
    L.rtcNewDevice.argtypes = [ctypes.c_char_p]
    L.rtcGetDeviceError.argtypes = [ctypes.c_void_p]
    assert L.rtcNewDevice(b"") is None
    assert L.rtcGetDeviceError(None) == 5
    from embree_amd import api
    with pytest.raises(api.RTCErrorException):
        api.Device("")


def test_product_never_imports_the_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "embree_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h")):
                txt = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "oracle" not in txt.replace("oracle/_ref", "").replace("# oracle", "") or f == "workloads.py" and "oracle in tests" in txt, \
                    os.path.join(dirpath, f) + " mentions the oracle"


def test_every_export_of_the_reference_library_resolves(lib_path):
    """tests/golden/ref_exports.txt = `nm -D` of the real libembree4.so built by oracle/ref.mk (154 rtc* symbols, default CPU build): an application linked
    against the reference resolves every one of them in this library (what is outside the triangle / quad / instance path records an error when called)."""
    here = os.path.dirname(os.path.abspath(__file__))
    want = set(open(os.path.join(here, "golden", "ref_exports.txt")).read().split())
    assert len(want) == 154
    out = subprocess.run(["nm", "-D", "--defined-only", lib_path], capture_output=True, text=True, check=True).stdout
    have = {ln.split()[-1] for ln in out.splitlines() if " T " in ln}
    assert not (want - have), sorted(want - have)
