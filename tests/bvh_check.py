"""Structural validation of a downloaded MI355X BVH (QNode / TriRec arrays, see embree_amd/csrc/bvh_common.h).

Checks what the traversal kernel relies on:
  * every valid input triangle appears in exactly one leaf, with v0/e1/e2/ids/mask as TriangleM::fill would store them
  * leaf ranges tile [0, num_triangles) without overlap; every leaf holds 1..max_leaf triangles
  * every node is referenced exactly once; child slots are filled from 0 and `count` matches
  * each child's DECODED quantised box contains all triangles below it (conservative quantisation)
"""
import numpy as np

LEAF = 0x80000000
EMPTY = 0xFFFFFFFF


def decode_child_boxes(node):
    scale = (node["exp"].astype(np.uint32) << 23).view(np.float32)          # 2^(e-127)
    w = node["child"]
    qlo = np.stack([w[:, 0] & 0xFF, (w[:, 0] >> 8) & 0xFF, (w[:, 0] >> 16) & 0xFF], -1).astype(np.float32)
    qhi = np.stack([w[:, 0] >> 24, w[:, 1] & 0xFF, (w[:, 1] >> 8) & 0xFF], -1).astype(np.float32)
    lo = (node["org"][None, :] + qlo * scale[None, :]).astype(np.float32)
    hi = (node["org"][None, :] + qhi * scale[None, :]).astype(np.float32)
    return lo, hi, w[:, 2]


def validate(nodes, tris, root_ref, meshes, masks=None, geom_ids=None, max_leaf=32):
    n_tris = tris.shape[0]
    # --- triangle records against the input meshes
    expect = {}
    for gi, (v, t) in enumerate(meshes):
        gid = gi if geom_ids is None else geom_ids[gi]
        v = np.asarray(v, np.float32)
        t = np.asarray(t, np.uint32)
        ok = (t < v.shape[0]).all(1)
        tv = v[np.where(ok[:, None], t, 0)]
        ok &= np.isfinite(tv).all((1, 2)) & (np.abs(tv) < 1.844e18).all((1, 2))
        for p in np.nonzero(ok)[0]:
            expect[(gid, int(p))] = tv[p]
    assert n_tris == len(expect), f"tree holds {n_tris} triangles, input has {len(expect)} valid ones"
    seen = set()
    for i in range(n_tris):
        key = (int(tris["geomID"][i]), int(tris["primID"][i]))
        assert key in expect and key not in seen, f"triangle record {i} {key} unexpected or duplicated"
        seen.add(key)
        a, b, c = expect[key]
        assert (tris["v0"][i] == a).all() and (tris["e1"][i] == a - b).all() and (tris["e2"][i] == c - a).all(), f"record {i} geometry"
        if masks is not None:
            gi = key[0] if geom_ids is None else geom_ids.index(key[0])
            assert tris["mask"][i] == masks[gi]
    if n_tris == 0:
        assert root_ref == EMPTY
        return dict(nodes=0, leaves=0, depth=0)
    tv1 = tris["v0"] - tris["e1"]
    tv2 = tris["v0"] + tris["e2"]
    tlo = np.minimum(np.minimum(tris["v0"], tv1), tv2)
    thi = np.maximum(np.maximum(tris["v0"], tv1), tv2)
    # e1/e2 are rounded differences, so v1/v2 are only recovered to ~1 ulp: allow that slack in containment
    slack = 4e-7 * np.maximum(np.abs(tlo), np.abs(thi)).max()

    covered = np.zeros(n_tris, np.int32)
    node_seen = np.zeros(nodes.shape[0], np.int32)
    stats = dict(nodes=0, leaves=0, depth=0)

    def visit(ref, depth):
        """returns (lo, hi) actual bounds of everything below ref"""
        stats["depth"] = max(stats["depth"], depth)
        if ref & LEAF:
            first, cnt = (ref & 0x7FFFFFFF) >> 5, (ref & 31) + 1
            assert cnt <= max_leaf and first + cnt <= n_tris, f"leaf ref {ref:#x}"
            covered[first:first + cnt] += 1
            ids = (tris["primID"][first:first + cnt].astype(np.uint64) << 32) | tris["geomID"][first:first + cnt]
            assert (np.diff(ids.astype(np.int64)) > 0).all() or cnt == 1, "leaf not sorted by (primID, geomID)"
            stats["leaves"] += 1
            return tlo[first:first + cnt].min(0), thi[first:first + cnt].max(0)
        assert ref < nodes.shape[0], f"node ref {ref} out of range"
        node_seen[ref] += 1
        stats["nodes"] += 1
        nd = nodes[ref]
        lo, hi, refs = decode_child_boxes(nd)
        cnt = int(nd["count"])
        assert 2 <= cnt <= 8, f"node {ref} count {cnt}"
        assert (refs[:cnt] != EMPTY).all() and (refs[cnt:] == EMPTY).all(), f"node {ref} slots"
        blo, bhi = np.full(3, np.inf, np.float32), np.full(3, -np.inf, np.float32)
        for i in range(cnt):
            clo, chi = visit(int(refs[i]), depth + 1)
            assert (lo[i] <= clo + slack).all() and (hi[i] >= chi - slack).all(), \
                f"node {ref} child {i}: decoded box {lo[i]}..{hi[i]} does not contain {clo}..{chi}"
            blo, bhi = np.minimum(blo, clo), np.maximum(bhi, chi)
        return blo, bhi

    import sys
    sys.setrecursionlimit(10000)
    visit(int(root_ref), 0)
    assert (covered == 1).all(), f"{int((covered != 1).sum())} triangles not covered exactly once"
    assert (node_seen == 1).all(), f"{int((node_seen != 1).sum())} nodes not referenced exactly once"
    return stats
