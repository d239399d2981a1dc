"""Shared comparison logic of the parity tests.

Bar (BASELINE.json north_star): t and Ng within 1e-4 relative, primID/geomID bit-exact.
Exact-tie rule: two correct tracers may legally report different
IDs when a ray hits two triangles at the same t (shared edges, coplanar duplicates); such a
ray is accepted iff the t the *checker* computes for the triangle the tested path reported
equals the checker's own t to within TIE_ULPS units in the last place ("1 ulp-scale", A.5; the
two paths' triangle arithmetic is the same operation for operation and differs only in the
reciprocal's starting value, v_rcp_f32 vs RCPPS, before the Newton step) AND the tested path's own
t is that value to within TIE_ULPS.  Ties are counted, bounded (max_tie_frac, 1e-4 of the rays
unless a test says why its scene has more: wall seams, duplicated triangles) and returned.
"""
import numpy as np

from embree_amd.rtypes import INVALID_ID

RTOL = 1e-4
TIE_ULPS = 4


def _ulps(a, b):
    """distance of two float32 arrays in units in the last place (ordered-integer view; NaN / inf -> huge)"""
    a = np.ascontiguousarray(a, np.float32)
    b = np.ascontiguousarray(b, np.float32)
    ia = a.view(np.int32).astype(np.int64)
    ib = b.view(np.int32).astype(np.int64)
    ia = np.where(ia < 0, -(ia & 0x7FFFFFFF), ia)
    ib = np.where(ib < 0, -(ib & 0x7FFFFFFF), ib)
    d = np.abs(ia - ib)
    return np.where(np.isfinite(a) & np.isfinite(b), d, np.int64(1) << 40)


def _tie(t_named, t_got, t_want):
    return (_ulps(t_named, t_want) <= TIE_ULPS) & (_ulps(t_got, t_named) <= TIE_ULPS)


def _rel(a, b):
    return np.abs(a - b) <= RTOL * np.maximum(np.abs(a), np.abs(b)) + 1e-30


def compare_closest(got, want, rays_in=None, tri_t=None, max_tie_frac=1e-4, label=""):
    """got/want: traced RTCRayHit arrays.  tri_t(rays_in, geomID, primID) -> t of a named triangle."""
    n = got.shape[0]
    g_hit = got["geomID"] != INVALID_ID
    w_hit = want["geomID"] != INVALID_ID
    same_ids = (got["geomID"] == want["geomID"]) & (got["primID"] == want["primID"])
    suspect = ~same_ids
    ties = 0
    bad = np.zeros(n, bool)
    if suspect.any():
        idx = np.nonzero(suspect)[0]
        if tri_t is None or rays_in is None:
            bad[idx] = True
        else:
            both = idx[g_hit[idx] & w_hit[idx]]
            t_named = tri_t(rays_in[both], got["geomID"][both], got["primID"][both])
            ok = _tie(t_named, got["tfar"][both], want["tfar"][both])
            ties = int(ok.sum())
            bad[both[~ok]] = True
            bad[idx[~(g_hit[idx] & w_hit[idx])]] = True     # hit/miss disagreement is never a tie
    assert not bad.any(), f"{label}: {int(bad.sum())}/{n} rays differ (first {np.nonzero(bad)[0][:8]})"
    assert ties <= max(2, max_tie_frac * n), f"{label}: too many ties {ties}/{n}"
    m = same_ids & w_hit
    assert _rel(got["tfar"][m], want["tfar"][m]).all(), f"{label}: tfar outside {RTOL}"
    for f in ("Ng_x", "Ng_y", "Ng_z"):
        ng_scale = np.sqrt(want["Ng_x"][m] ** 2 + want["Ng_y"][m] ** 2 + want["Ng_z"][m] ** 2)
        assert (np.abs(got[f][m] - want[f][m]) <= RTOL * ng_scale + 1e-30).all(), f"{label}: {f} outside {RTOL}"
    assert (np.abs(got["u"][m] - want["u"][m]) <= 1e-4).all() and (np.abs(got["v"][m] - want["v"][m]) <= 1e-4).all()
    # a miss leaves the whole record untouched (doc/src/api/rtcIntersect1.md)
    miss = ~w_hit & ~g_hit
    if rays_in is not None and miss.any():
        assert (got[miss].tobytes() == rays_in[miss].tobytes()), f"{label}: a missed ray was modified"
    return dict(rays=n, hits=int(w_hit.sum()), ties=ties)


def compare_occluded(got_tfar, want_tfar, rays_tfar_in, max_flip_frac=0.0, label=""):
    """Occluded: tfar == -inf iff occluded, untouched otherwise (doc/src/api/rtcOccluded1.md)."""
    g = np.isneginf(got_tfar)
    w = np.isneginf(want_tfar)
    flips = int((g != w).sum())
    assert flips <= max_flip_frac * g.shape[0], f"{label}: {flips} occlusion results differ"
    keep = ~g
    assert (got_tfar[keep] == rays_tfar_in[keep]).all(), f"{label}: unoccluded ray was modified"
    return dict(rays=g.shape[0], occluded=int(w.sum()), flips=flips)


def compare_closest_arbitrated(got, want_fast, want_robust, rays_in, tri_t, max_tie_frac=1e-4, max_ref_miss_frac=1e-4, label=""):
    """Fast-mode parity on geometry where the REFERENCE's fast mode is itself not exact.  Embree's default node test (node_intersector1.h:484-531,
    rdir from an approximate reciprocal, no safety margin) loses hits on long thin or axis-aligned geometry; RTC_SCENE_FLAG_ROBUST exists for that
    reason.  A ray on which the tested path and the fast reference disagree is therefore accepted iff it is an exact-t tie or the tested
    path reports exactly what the ROBUST reference reports (a hit the fast reference lost); and the tested path itself must never be farther than the
    robust reference's hit (it must not lose anything).  Returns counts; asserts on anything else.
    max_ref_miss_frac: the share of the rays that may follow the robust reference -- 1e-4 (round 5; 1e-3 before: twenty times what is measured, 55 of 2^20 on the
    powerplant stand-in, room for a regression to hide in); tests on geometry that provokes the reference's fast mode (a scene 1e5 away from the origin, the
    fuzzed degenerate scenes) state their own, larger share."""
    n = got.shape[0]
    same = (got["geomID"] == want_fast["geomID"]) & (got["primID"] == want_fast["primID"])
    idx = np.nonzero(~same)[0]
    g_hit, f_hit = got["geomID"][idx] != INVALID_ID, want_fast["geomID"][idx] != INVALID_ID
    ties = ref_missed = 0
    bad = []
    if idx.size:
        t_named = np.full(idx.size, np.inf, np.float32)
        both = g_hit & f_hit
        if both.any():
            t_named[both] = tri_t(rays_in[idx[both]], got["geomID"][idx[both]], got["primID"][idx[both]])
        is_tie = both & _tie(t_named, got["tfar"][idx], want_fast["tfar"][idx])
        # "what the robust reference reports": the same triangle, or one at the same distance (an exact-t tie with the robust answer)
        r_hit = want_robust["geomID"][idx] != INVALID_ID
        t_self = np.full(idx.size, np.inf, np.float32)
        gh = np.nonzero(g_hit)[0]
        if gh.size:
            t_self[gh] = tri_t(rays_in[idx[gh]], got["geomID"][idx[gh]], got["primID"][idx[gh]])
        as_robust = g_hit & r_hit & _rel(got["tfar"][idx], want_robust["tfar"][idx]) & _rel(t_self, want_robust["tfar"][idx]) & \
            (got["tfar"][idx] < np.where(f_hit, want_fast["tfar"][idx], np.inf))
        ties, ref_missed = int(is_tie.sum()), int((as_robust & ~is_tie).sum())
        bad = idx[~(is_tie | as_robust)]
    assert len(bad) == 0, f"{label}: {len(bad)}/{n} rays differ from the fast AND the robust reference (first {bad[:8]})"
    r_hit = want_robust["geomID"] != INVALID_ID
    # nothing may be lost against the robust reference -- except where the fast reference gives the very same answer (the Moeller-Trumbore test is not
    # watertight: a ray through an edge can slip between two triangles in the reference's fast mode as well, and bit-identical arithmetic slips with it)
    lost = r_hit & ((got["geomID"] == INVALID_ID) | (got["tfar"] > want_robust["tfar"] * np.float32(1 + RTOL))) & ~same
    assert not lost.any(), f"{label}: {int(lost.sum())} rays end farther than the robust reference's hit (first {np.nonzero(lost)[0][:8]})"
    assert ties <= max(2, max_tie_frac * n), f"{label}: too many ties {ties}/{n}"
    assert ref_missed <= max_ref_miss_frac * n, f"{label}: {ref_missed} hits missing in the fast reference"
    m = same & (want_fast["geomID"] != INVALID_ID)
    assert _rel(got["tfar"][m], want_fast["tfar"][m]).all(), f"{label}: tfar outside {RTOL}"
    return dict(rays=n, hits=int((got["geomID"] != INVALID_ID).sum()), ties=ties, fast_reference_missed=ref_missed)
