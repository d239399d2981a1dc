"""GPU box tool (not a pytest file): time of the Morton build's radix sort on its own.  [MI355_LIB=embree_amd/lib/variant_x.so] python tests/gpu_sort_time.py [n] [reps] [check]
keys: 63-bit Morton-like codes of n points on a few surfaces (skewed high digits, like a scene's), or uniform random with `random`."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from embree_amd import api                                       # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4762764
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 12
check = len(sys.argv) > 3 and sys.argv[3] == "check"
L = api.load()
rng = np.random.default_rng(3)
if os.environ.get("KEYS") == "random":
    keys = rng.integers(0, 1 << 63, n, dtype=np.uint64)
else:                                                            # points on three spheres: a surface in a 2^21 grid
    u, v = rng.random(n), rng.random(n)
    th, ph = 2 * np.pi * u, np.arccos(2 * v - 1)
    r = np.choice = rng.choice([0.2, 0.33, 0.45], n)
    P = 0.5 + r[:, None] * np.stack([np.sin(ph) * np.cos(th), np.sin(ph) * np.sin(th), np.cos(ph)], 1)
    q = np.clip((P * 2097152.0), 0, 2097151).astype(np.uint64)

    def spread(x):
        x = x & np.uint64(0x1FFFFF)
        x = (x | x << np.uint64(32)) & np.uint64(0x1F00000000FFFF); x = (x | x << np.uint64(16)) & np.uint64(0x1F0000FF0000FF)
        x = (x | x << np.uint64(8)) & np.uint64(0x100F00F00F00F00F); x = (x | x << np.uint64(4)) & np.uint64(0x10C30C30C30C30C3)
        return (x | x << np.uint64(2)) & np.uint64(0x1249249249249249)
    keys = spread(q[:, 0]) | (spread(q[:, 1]) << np.uint64(1)) | (spread(q[:, 2]) << np.uint64(2))
dk = api.DeviceArray.from_numpy(keys)
ok, oi = api.DeviceArray(n * 8), api.DeviceArray(n * 4)
ms = C.c_float()
t = []
for _ in range(reps):
    assert L.mi355_sort_keys63(0, dk.ptr, ok.ptr, oi.ptr, n, C.byref(ms)) == 0, L.mi355_last_error().decode()
    t.append(ms.value)
t = sorted(t)
good = ""
if check:
    want = np.argsort(keys, kind="stable").astype(np.uint32)
    good = " correct" if np.array_equal(oi.download(np.uint32), want) else " WRONG ORDER"
print("SORT %s n=%d keys=%s: min %.1f us, median %.1f us (seven passes)%s" % (os.path.basename(os.environ.get("MI355_LIB", "product")), n, os.environ.get("KEYS", "surface"), t[0] * 1e3, t[len(t) // 2] * 1e3, good))
