"""CPU-only probe of the reference baseline on the GPU box's host (test infrastructure): why does rtcIntersect1 over 16 x 2^20 rays run 13 x slower per ray than over
2^20?  Variants: tiles, dynamic blocks vs static per-thread ranges, huge pages, BVH built by 16 vs all tasking threads, interleaved memory policy.
    python tools/cpu_probe.py [--interleave]"""
import ctypes
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if "--interleave" in sys.argv:                                # set_mempolicy(MPOL_INTERLEAVE, all nodes) before anything is allocated (x86_64 syscall 238)
    nodes = [int(d[4:]) for d in os.listdir("/sys/devices/system/node") if d.startswith("node")]
    mask = (ctypes.c_ulong * 16)()
    for n in nodes:
        mask[n // 64] |= 1 << (n % 64)
    rc = ctypes.CDLL(None, use_errno=True).syscall(238, 3, mask, 1024)
    print("set_mempolicy(interleave over %d nodes) rc=%d errno=%d" % (len(nodes), rc, ctypes.get_errno()))
from embree_amd import workloads as W   # noqa: E402
from oracle import refembree, restate   # noqa: E402

hw = refembree.hw_threads()
print("host threads", hw, "numa nodes", len([d for d in os.listdir("/sys/devices/system/node") if d.startswith("node")]),
      "autonuma", open("/proc/sys/kernel/numa_balancing").read().strip() if os.path.exists("/proc/sys/kernel/numa_balancing") else "?",
      "thp", open("/sys/kernel/mm/transparent_hugepage/enabled").read().strip())
meshes = W.synthetic_crown(num_phi=158)
o = restate.OracleScene()
rays = W.incoherent_rays(1 << 20, [2.0, 2.0, 1.5], seed=5)     # (no GPU here: incoherent rays from the scene centre)
for bt in (16, hw):
    s = refembree.RefScene("threads=%d,start_threads=1,set_affinity=1" % bt)
    for v, t in meshes:
        s.add_mesh(v, t)
    t0 = time.time(); s.commit(); tb = time.time() - t0
    s.run_tiled(rays, 1, hw)
    line = "BVH built by %3d threads (%.2f s):" % (bt, tb)
    for mode in (0, 1, 2, 3):
        for tiles in (1, 4, 16):
            best = min(s.run_tiled(rays, tiles, hw, mode=mode) for _ in range(3))
            line += "  m%d x%-2d %6.1f" % (mode, tiles, tiles * rays.shape[0] / best / 1e6)
    print(line, flush=True)
    for th in (64, 128):
        best = min(s.run_tiled(rays, 16, th, mode=1) for _ in range(2))
        print("   %d threads, static, x16: %.1f Mrays/s" % (th, 16 * rays.shape[0] / best / 1e6), flush=True)
    s.close()
