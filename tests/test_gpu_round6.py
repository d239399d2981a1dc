"""Round 6 GPU tests (pytest -m gpu): the boundary items of VERDICT r05 / ADVICE r05 -- overlapping index elements through a COMMIT, the property table -- and the
gates of the round's kernel work (small-batch launches give the bytes of large-batch launches; the commit's trees are bit-identical across rebuilds).  All through the
C ABI; the checker is the REAL reference (oracle/_ref)."""
import ctypes as C
import hashlib
import os
import subprocess
import sys

import numpy as np
import pytest

from embree_amd import workloads as W
from embree_amd.rtypes import rays_of, INVALID_ID, RAYHIT_DTYPE, RAY_DTYPE, make_rayhits
from tests.helpers import compare_closest, compare_occluded

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def api():
    from embree_amd import api as a
    return a


@pytest.fixture(scope="module")
def dev(api):
    d = api.Device("gpu=0")
    yield d
    d.release()


@pytest.fixture(scope="module")
def ref():
    from oracle import refembree
    assert refembree.available(), "oracle/_ref is missing on the GPU box: make -f oracle/ref.mk must have run in the build container before the snapshot was taken"
    return refembree


# ------------------------------------------------------------------------------------------- overlapping index elements through rtcCommitScene
@pytest.mark.parametrize("flags", [0, 4])                       # fast, RTC_SCENE_FLAG_ROBUST
def test_quad_indices_with_12_byte_stride_commit_and_trace_vs_reference(api, dev, ref, flags):
    """A UINT4 quad index view with a 12-byte stride -- consecutive quads share one word -- is what the reference's BufferStrideTest binds
    (tutorials/verify/verify.cpp:995-1008), and the reference reads quad i at offset + i * stride whatever the stride (kernels/common/buffer.h BufferView::operator[]).
    Round 5 accepted the view in rtcSetSharedGeometryBuffer and refused it in rtcCommitScene; verify never commits that scene, so nothing showed.  Here the scene IS
    committed and traced: a strip of non-planar quads whose quad i is the words [3 i, 3 i + 4) of ONE flat array, against the real reference given the same quads written out
    with the usual 16-byte stride."""
    rng = np.random.default_rng(12)
    nq = 300                                                      # a strip of nq quads stacked in y; quad i = (w[3i], w[3i+1], w[3i+2], w[3i+3]) of ONE word array, every word a vertex of its own:
    words = np.arange(3 * nq + 1, dtype=np.uint32)                # ... vertex 3i = (0, y_i), 3i+1 = (1, y_i), 3i+2 = (1, y_i + h), 3i+3 = (0, y_i + h) = the first corner of quad i + 1
    h = 1.0 / nq
    v = np.zeros((3 * nq + 1, 3), np.float32)
    i = np.arange(nq)
    v[3 * i] = np.stack([np.zeros(nq), i * h, np.zeros(nq)], -1)
    v[3 * i + 1] = np.stack([np.ones(nq), i * h, np.zeros(nq)], -1)
    v[3 * i + 2] = np.stack([np.ones(nq), (i + 1) * h, np.zeros(nq)], -1)
    v[3 * nq] = (0.0, 1.0, 0.0)
    v[:, 2] = 0.05 * rng.random(3 * nq + 1, dtype=np.float32)     # non-planar quads, cracks along the seams (every quad has corners of its own)
    quads = np.stack([words[0:3 * nq:3], words[1:3 * nq:3], words[2:3 * nq:3], words[3:3 * nq + 1:3]], -1).astype(np.uint32)   # the same quads, element by element
    assert (quads[1:, 0] == quads[:-1, 3]).all()                  # (the shared word)
    s = api.Scene(dev, flags)
    assert s.add_quad_mesh(v, quads, index_words=words, index_stride=12) == 0
    s.commit()
    dev.check()                                                    # no error from the commit
    assert s.info()["num_triangles"] == 2 * nq
    R = ref.RefScene("threads=4", flags=flags)
    R.add_quads(v, quads)
    R.commit()
    assert R.error() == 0
    n = 20000
    org = np.stack([rng.random(n, dtype=np.float32), rng.random(n, dtype=np.float32), np.full(n, 1.0, np.float32)], -1)
    tgt = np.stack([rng.random(n, dtype=np.float32), rng.random(n, dtype=np.float32), np.zeros(n, np.float32)], -1)
    rays = make_rayhits(org, tgt - org)
    want, got = rays.copy(), rays.copy()
    R.intersect1(want, threads=4)
    s.intersect1M(got)

    def quad_t(rr, geom, prim):                                   # t of a named quad (either half), fp64: the tie classifier of tests/helpers.py
        out = np.zeros(rr.shape[0], np.float32)
        for j in range(rr.shape[0]):
            q = quads[int(prim[j])]
            o = np.array([rr["org_x"][j], rr["org_y"][j], rr["org_z"][j]], np.float64)
            d = np.array([rr["dir_x"][j], rr["dir_y"][j], rr["dir_z"][j]], np.float64)
            best = np.inf
            for a, b, c in ((q[0], q[1], q[3]), (q[2], q[3], q[1])):
                A, B, Cc = v[a].astype(np.float64), v[b].astype(np.float64), v[c].astype(np.float64)
                nrm = np.cross(B - A, Cc - A)
                den = np.dot(nrm, d)
                if den != 0.0:
                    t = np.dot(nrm, A - o) / den
                    if abs(t - rr["tfar"][j]) < abs(best - rr["tfar"][j]):
                        best = t
            out[j] = best
        return out
    st = compare_closest(got, want, rays, quad_t, max_tie_frac=0.01, label="12-byte-stride quads vs reference (flags %d)" % flags)
    assert st["hits"] > 0.3 * n
    wr, gr = rays_of(rays), rays_of(rays)
    R.occluded1(wr, threads=4)
    s.occluded1M(gr)
    compare_occluded(gr["tfar"], wr["tfar"], rays_of(rays)["tfar"], label="12-byte-stride quads, occlusion")
    R.close()
    s.release()


def test_property_table_tells_what_is_built(api, dev):
    """rtcGetDeviceProperty against what the library implements (reference table: kernels/common/device.cpp:480-600).  FILTER_FUNCTION_SUPPORTED answered
    0 for three rounds although geometry and argument filters run on every entry point, which hid the reference's intersection_filter test group."""
    L = api.load()
    get = lambda p: L.rtcGetDeviceProperty(dev.h, p)
    assert get(66) == 1                                            # RTC_DEVICE_PROPERTY_FILTER_FUNCTION_SUPPORTED
    assert get(129) == 1                                           # RTC_DEVICE_PROPERTY_JOIN_COMMIT_SUPPORTED (rtcJoinCommitScene commits under the scene's lock)
    assert get(96) == 1 and get(97) == 1 and get(64) == 1          # triangles, quads, ray masks
    assert get(32) == 1 and get(33) == 1 and get(34) == 1          # rtcIntersect4/8/16 are entry points of their own
    for p in (62, 63, 65, 67, 68, 98, 99, 100, 101, 128, 130, 140, 141):   # off in the reference's default build, or not built here
        assert get(p) == 0, p
    assert get(142) == 1                                           # RTC_DEVICE_PROPERTY_HIP_DEVICE (extension)
    dev.check()
