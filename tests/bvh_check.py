"""Structural validation of a downloaded MI355X BVH (CNode / TriRec arrays, see embree_amd/csrc/bvh_common.h).

Checks what the traversal kernel relies on:
  * every valid input triangle appears in exactly one leaf slot, with v0/e1/e2/ids/mask as TriangleM::fill would store them
  * the root is node 0; every other node is referenced exactly once, through childBase + rank of its slot in imask
  * meta bytes: inner slots (1 << 5) | (24 + slot) and the imask bit set; leaf slots unary count (1..max_leaf) << 5 | offset,
    offsets of a node's leaf slots tile [0, #triangles of the node) in slot order, <= 24 triangles per node; empty slots 0
  * each child's DECODED quantised box contains all triangles below it (conservative quantisation)
"""
import sys

import numpy as np

EMPTY = 0xFFFFFFFF


def decode_child_boxes(node):
    # the EXACT planes org + q * 2^(e-127) (fp64 holds them exactly): the fast traversal path evaluates q * (scale * rdir) + (org - ray.org) * rdir,
    # which never rounds the plane itself, so the exact plane is what must lie outside the child's geometry (build_wide.inl, quantise_slots)
    scale = (node["exp"].astype(np.uint32) << 23).view(np.float32).astype(np.float64)          # 2^(e-127), one per axis
    lo = node["org"][None, :].astype(np.float64) + node["qlo"].T.astype(np.float64) * scale[None, :]   # [slot][axis]
    hi = node["org"][None, :].astype(np.float64) + node["qhi"].T.astype(np.float64) * scale[None, :]
    return lo, hi


def validate(nodes, tris, root_ref, meshes, masks=None, geom_ids=None, max_leaf=3, allow_splits=False):
    """allow_splits: the tree may hold a triangle several times (spatial splits: every record of a cut triangle bounds only its piece, so the containment
    check is made for the triangles that appear once; every triangle must still be there at least once)"""
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
    assert n_tris == len(expect) or (allow_splits and n_tris >= len(expect)), f"tree holds {n_tris} triangles, input has {len(expect)} valid ones"
    seen = {}
    for i in range(n_tris):
        key = (int(tris["geomID"][i]), int(tris["primID"][i]))
        assert key in expect and (allow_splits or key not in seen), f"triangle record {i} {key} unexpected or duplicated"
        seen[key] = seen.get(key, 0) + 1
        a, b, c = expect[key]
        assert (tris["v0"][i] == a).all() and (tris["e1"][i] == a - b).all() and (tris["e2"][i] == c - a).all(), f"record {i} geometry"
        if masks is not None:
            gi = key[0] if geom_ids is None else geom_ids.index(key[0])
            assert tris["mask"][i] == masks[gi]
    assert len(seen) == len(expect), "a triangle is missing from the tree"
    whole = np.array([seen[(int(tris["geomID"][i]), int(tris["primID"][i]))] == 1 for i in range(n_tris)], bool) if n_tris else np.zeros(0, bool)
    if n_tris == 0:
        assert root_ref == EMPTY
        return dict(nodes=0, leaves=0, depth=0)
    assert root_ref == 0 and nodes.shape[0] >= 1
    tv1 = tris["v0"] - tris["e1"]
    tv2 = tris["v0"] + tris["e2"]
    tlo = np.minimum(np.minimum(tris["v0"], tv1), tv2)
    thi = np.maximum(np.maximum(tris["v0"], tv1), tv2)
    # e1/e2 are rounded differences, so v1/v2 are only recovered to ~1 ulp: allow that slack in containment
    slack = 4e-7 * np.maximum(np.abs(tlo), np.abs(thi)).max()

    covered = np.zeros(n_tris, np.int32)
    node_seen = np.zeros(nodes.shape[0], np.int32)
    stats = dict(nodes=0, leaves=0, depth=0)

    def visit(idx, depth):
        """returns (lo, hi) actual bounds of everything below node idx"""
        stats["depth"] = max(stats["depth"], depth)
        assert idx < nodes.shape[0], f"node index {idx} out of range"
        node_seen[idx] += 1
        stats["nodes"] += 1
        nd = nodes[idx]
        lo, hi = decode_child_boxes(nd)
        imask = int(nd["imask"])
        blo, bhi = np.full(3, np.inf, np.float32), np.full(3, -np.inf, np.float32)
        next_ofs, rank, used = 0, 0, 0
        for s in range(8):
            m = int(nd["meta"][s])
            if m == 0:
                assert not (imask >> s) & 1, f"node {idx} slot {s}: empty slot flagged inner"
                continue
            used += 1
            if (imask >> s) & 1:
                assert m == (1 << 5) | (24 + s), f"node {idx} slot {s}: inner meta {m:#x}"
                clo, chi = visit(int(nd["childBase"]) + rank, depth + 1)
                rank += 1
            else:
                bits, ofs = m >> 5, m & 31
                assert bits in (1, 3, 7), f"node {idx} slot {s}: unary count {bits:#b}"
                cnt = {1: 1, 3: 2, 7: 3}[bits]
                assert cnt <= max_leaf and ofs == next_ofs and ofs + cnt <= 24, f"node {idx} slot {s}: leaf offset {ofs} count {cnt}"
                next_ofs += cnt
                first = int(nd["triBase"]) + ofs
                assert first + cnt <= n_tris
                covered[first:first + cnt] += 1
                ids = (tris["primID"][first:first + cnt].astype(np.uint64) << 32) | tris["geomID"][first:first + cnt]
                assert cnt == 1 or (np.diff(ids.astype(np.int64)) >= (0 if allow_splits else 1)).all(), "leaf not sorted by (primID, geomID)"
                stats["leaves"] += 1
                w = whole[first:first + cnt]
                if not w.any():                                   # only pieces of cut triangles: nothing this validator can bound
                    continue
                clo, chi = tlo[first:first + cnt][w].min(0), thi[first:first + cnt][w].max(0)
            assert (lo[s] <= clo + slack).all() and (hi[s] >= chi - slack).all(), \
                f"node {idx} slot {s}: decoded box {lo[s]}..{hi[s]} does not contain {clo}..{chi}"
            blo, bhi = np.minimum(blo, clo), np.maximum(bhi, chi)
        assert used >= 1, f"node {idx} has no children"
        return blo, bhi

    sys.setrecursionlimit(10000)
    visit(0, 1)
    assert (covered == 1).all(), f"{int((covered != 1).sum())} triangles not covered exactly once"
    assert (node_seen == 1).all(), f"{int((node_seen != 1).sum())} nodes not referenced exactly once"
    return stats


def embree_metric_sah(nodes, tris=None):
    """SAH of a downloaded tree in the REFERENCE's metric (BVHNStatistics, kernels/bvh/bvh_statistics.cpp:42-160): every inner node costs its half area,
    every leaf its half area x the number of Triangle4 blocks (here: a leaf slot holds <= 3 triangles = one block), divided by the root's half area.
    Areas are those of the decoded (quantised, i.e. slightly enlarged) child boxes, so the figure is a little pessimistic for this tree."""
    scale = (nodes["exp"].astype(np.uint32) << 23).view(np.float32).astype(np.float64)                 # [node][axis]
    org = nodes["org"].astype(np.float64)
    lo = org[:, :, None] + nodes["qlo"].astype(np.float64) * scale[:, :, None]                        # [node][axis][slot]
    hi = org[:, :, None] + nodes["qhi"].astype(np.float64) * scale[:, :, None]
    d = np.maximum(hi - lo, 0.0)
    area = d[:, 0] * (d[:, 1] + d[:, 2]) + d[:, 1] * d[:, 2]                                            # half area per slot
    used = nodes["meta"] != 0
    inner = used & (((nodes["imask"][:, None] >> np.arange(8)[None, :]) & 1) != 0)
    leaf = used & ~inner
    # the root's own box = union of its children
    rlo = np.where(used[0][None, :], lo[0], np.inf).min(1)
    rhi = np.where(used[0][None, :], hi[0], -np.inf).max(1)
    rd = rhi - rlo
    root = rd[0] * (rd[1] + rd[2]) + rd[1] * rd[2]
    nodes_sah = (root + area[inner].sum()) / root
    leaves_sah = area[leaf].sum() / root
    return dict(sah=nodes_sah + leaves_sah, sah_nodes=nodes_sah, sah_leaves=leaves_sah, nodes=int(nodes.shape[0]), leaves=int(leaf.sum()))
