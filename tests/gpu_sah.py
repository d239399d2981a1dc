"""SAH of this builder's trees beside the reference's (compare stat->sah()), in the reference's own metric, + what the
trees cost to traverse.  Scenes: the crown stand-in, and the long-triangle scene of test_high_quality_presplit."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from embree_amd import api, workloads as W
from embree_amd.rtypes import make_rayhits
from oracle import refembree
from tests import bvh_check
from tests.test_gpu_reference_suite import _sticks
dev = api.Device("gpu=0")
devPre = api.Device("gpu=0,presplits=1")
scenes = {"sticks+sphere (25k)": [_sticks(300, 5), W.triangle_sphere(np.array([0.5, 0.5, 0.5], np.float32), 0.25, 80, noise=0.1, seed=2)],
          "crown stand-in (phi=%d)" % int(os.environ.get("PHI", "60")): W.synthetic_crown(num_phi=int(os.environ.get("PHI", "60")))}
for name, meshes in scenes.items():
    lo, hi = W.scene_bounds(meshes)
    org = np.random.default_rng(3).random((200000, 3), dtype=np.float32) * (hi - lo) * 1.2 + lo - 0.1 * (hi - lo)
    tgt = np.random.default_rng(4).random((200000, 3), dtype=np.float32) * (hi - lo) + lo
    rays = make_rayhits(org, tgt - org)
    print("== %s: %d triangles" % (name, W.num_triangles(meshes)))
    for q, qn in ((1, "MEDIUM"), (2, "HIGH")):
        rs = refembree.build_stats(meshes, quality=q, threads=16)
        print("   reference %-6s %-42s sah %.3f (nodes %.3f leaves %.3f) nodes %d depth %d prims %d" % (qn, rs.get("builder"), rs["sah"], rs["sah_nodes"], rs["sah_leaves"], rs["nodes"], rs["depth"], rs["primitives"]))
    for q, qn in ((None, "MEDIUM"), (api.RTC_BUILD_QUALITY_HIGH, "HIGH"), ("pre", "HIGH presplits=1"), (api.RTC_BUILD_QUALITY_LOW, "LOW")):
        s = api.make_scene(devPre if q == "pre" else dev, meshes, quality=api.RTC_BUILD_QUALITY_HIGH if q == "pre" else q)
        nodes, tris = s.download_bvh()
        m = bvh_check.embree_metric_sah(nodes)
        d = api.DeviceArray.from_numpy(rays)
        st = s.trace_stats(d.ptr, rays.shape[0], 96)
        i = s.info()
        print("   mi355     %-16s sah(ref metric) %.3f (nodes %.3f leaves %.3f) own metric %.3f nodes %d depth %d refs %d | nodes/ray %.2f tris/ray %.2f build %.2f ms"
              % (qn, m["sah"], m["sah_nodes"], m["sah_leaves"], i["sah"], i["num_nodes"], i["depth"], i["num_triangles"], st["nodes"] / rays.shape[0], st["tris"] / rays.shape[0], i["build_ms"]))
        d.free(); s.release()
