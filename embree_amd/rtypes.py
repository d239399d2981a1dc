"""numpy views of the Embree-4 ray structures (default ABI).

Layouts follow include/embree4/rtcore.h (which mirrors the reference's
include/embree4/rtcore_ray.h:11-52): RTCRay = 48 B, RTCHit = 48 B (the last
12 B are alignment padding), RTCRayHit = 96 B.
"""
import numpy as np

INVALID_ID = 0xFFFFFFFF

_ray_fields = [
    ("org_x", "<f4"), ("org_y", "<f4"), ("org_z", "<f4"), ("tnear", "<f4"),
    ("dir_x", "<f4"), ("dir_y", "<f4"), ("dir_z", "<f4"), ("time", "<f4"),
    ("tfar", "<f4"), ("mask", "<u4"), ("id", "<u4"), ("flags", "<u4"),
]
_hit_fields = [
    ("Ng_x", "<f4"), ("Ng_y", "<f4"), ("Ng_z", "<f4"), ("u", "<f4"), ("v", "<f4"),
    ("primID", "<u4"), ("geomID", "<u4"), ("instID", "<u4"), ("instPrimID", "<u4"),
    ("_pad", "<u4", (3,)),
]
RAY_DTYPE = np.dtype(_ray_fields)
RAYHIT_DTYPE = np.dtype(_ray_fields + _hit_fields)
assert RAY_DTYPE.itemsize == 48 and RAYHIT_DTYPE.itemsize == 96


def make_rayhits(org, dir, tnear=0.0, tfar=np.inf, mask=0xFFFFFFFF):
    """Build an RTCRayHit array the way every reference tutorial does
    (e.g. tutorials/minimal/minimal.cpp:118-135): geomID/primID/instID invalid."""
    org = np.asarray(org, np.float32).reshape(-1, 3)
    dir = np.asarray(dir, np.float32).reshape(-1, 3)
    n = org.shape[0]
    rh = np.zeros(n, RAYHIT_DTYPE)
    rh["org_x"], rh["org_y"], rh["org_z"] = org[:, 0], org[:, 1], org[:, 2]
    rh["dir_x"], rh["dir_y"], rh["dir_z"] = dir[:, 0], dir[:, 1], dir[:, 2]
    rh["tnear"] = tnear
    rh["tfar"] = tfar
    rh["mask"] = mask
    rh["id"] = np.arange(n, dtype=np.uint32)
    rh["primID"] = INVALID_ID
    rh["geomID"] = INVALID_ID
    rh["instID"] = INVALID_ID
    rh["instPrimID"] = INVALID_ID
    return rh


def rays_of(rayhits):
    """RTCRay copy (48 B records) of an RTCRayHit array, for rtcOccluded*."""
    r = np.zeros(rayhits.shape[0], RAY_DTYPE)
    for name in RAY_DTYPE.names:
        r[name] = rayhits[name]
    return r
