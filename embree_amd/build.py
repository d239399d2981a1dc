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
SOURCES = ["build.hip", "trace.hip", "trace_fptr.hip", "shard.hip", "rtcore_api.cpp"]
HEADERS = ["trace.hip", "bvh_common.h", "internal.h", "../../include/embree4/rtcore.h", "../../include/embree_amd_hip.h",
           "build_common.inl", "build_primref.inl", "build_presplit.inl", "build_binning.inl", "build_top.inl", "build_spatial.inl", "build_small.inl",
           "build_morton.inl", "build_sort.inl", "build_wide.inl", "build_leaves.inl"]    # parts of build.hip (one translation unit)
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fvisibility=hidden",
         "-Wall", "-Wno-unused-function", "-Wno-unused-variable", "-Wno-unused-value", "-Wno-unused-result"]
# per-source flags.  trace.hip: the SLP vectoriser pairs scalar fp32 operations of the node set-up and of the triangle test into v_pk_mul / v_pk_add / v_pk_fma_f32 and
# pays for it with v_mov (to bring operands into adjacent registers) and v_and (|x| has no packed form) -- on gfx950 a packed fp32 instruction issues in the time of two
# scalar ones (profiles/r02_valu_issue_costs.txt), so the pairs buy nothing and the moves cost: +3.7 % rays per second without it (profiles/r05_trace.md).  The packed
# FMAs of the slab test are written by hand (test4) and stay.  Same arithmetic either way (-ffp-contract=off): results are bit-identical.
# trace_fptr.hip = trace.hip again, for the kernels that call a device filter function only.  Round 5 built them at -O1: above it the loop around the indirect call lost rays or
# faulted with ANY callee.  Round 6 bisected that on the GPU (tools/r06_fptr_variants.sh, profiles/r06_device_filter.md): it is the record PREFETCH of step 0 -- triangle records
# asked for at the top of an iteration and still in flight, together with the five node loads of step 3a, when step 4 calls through the pointer -- and none of the other
# hand-written pieces (SGPR lane masks, the DPP scan, the "undefined register" asm): without the prefetch the calling kernels are correct at -O2 and -O3, bit for bit what -O1 gave.
EXTRA_FLAGS = {"trace.hip": ["-fno-slp-vectorize"], "trace_fptr.hip": ["-fno-slp-vectorize", "-DMI355_TRI_PREFETCH=0"]}


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
        cmd = [hipcc, "-x", "hip"] + FLAGS + EXTRA_FLAGS.get(src, []) + ["-c", os.path.join(CSRC, src), "-o", obj]
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
