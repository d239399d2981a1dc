"""Builds embree_amd/lib/libembree4_mi355.so: the HIP kernels + the C ABI, for gfx950 only.

hipcc cross-compiles without a GPU; the .so is built in-tree so that it travels with gpurun.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libembree4_mi355.so")
SOURCES = ["build.hip", "trace.hip", "shard.hip", "rtcore_api.cpp"]
HEADERS = ["bvh_common.h", "internal.h", "../../include/embree4/rtcore.h", "../../include/embree_amd_hip.h",
           "build_common.inl", "build_primref.inl", "build_presplit.inl", "build_binning.inl", "build_top.inl", "build_spatial.inl", "build_small.inl",
           "build_morton.inl", "build_wide.inl", "build_leaves.inl"]    # parts of build.hip (one translation unit)
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fvisibility=hidden",
         "-Wall", "-Wno-unused-function", "-Wno-unused-variable", "-Wno-unused-value", "-Wno-unused-result"]


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    if not force and not _stale():
        return LIB
    os.makedirs(LIBDIR, exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objs = []
    for src in SOURCES:
        obj = os.path.join(LIBDIR, src + ".o")
        cmd = [hipcc, "-x", "hip"] + FLAGS + ["-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
        objs.append(obj)
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-Wl,-z,now", "-o", LIB] + objs + ["-ldl"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
