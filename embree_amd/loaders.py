"""Scene loaders for the triangle path: the reference tutorials' asset formats, so that real crown.ecs /
powerplant.ecs drop into bench.py and the tests when they are available ($EMBREE_MODEL_DIR; they are not shipped with the reference).

  load_scene(path)      dispatch on the extension: .ecs, .xml, .obj           -> Scene(meshes, camera)
  load_ecs(path)        tutorial command files (tutorials/common/tutorial/tutorial.cpp: "-i file", "-c file", "-vp/-vi/-vu/-fov")
  load_xml(path)        Embree XML scene graph + side-car "<file>.bin" (tutorials/common/scenegraph/xml_loader.cpp): Group, Transform
                        (AffineSpace: 12 row-major floats or translate= / scale= / rotate_[xyz]=), ref, TriangleMesh (<positions ofs= size=> /
                        <triangles ofs= size=> into the .bin, or inline text, :452-471,515-534,1018-1051), QuadMesh (each quad = triangles
                        (v0,v1,v3),(v2,v3,v1), the split Embree's quad intersector uses).  The scene graph is flattened: one world-space
                        mesh per TriangleMesh / QuadMesh instance (this core has no instancing).
  load_obj(path)        Wavefront OBJ: 'v' and 'f' records, negative indices, polygons fan-triangulated (0,k,k+1)
                        (tutorials/common/scenegraph/obj_loader.cpp); one mesh, or one per 'usemtl'/'g'/'o' group with split_groups=True
  save_xml(path, meshes, camera)   writer for the XML + .bin pair (tests, and to hand a generated scene to the reference's viewer)

meshes = [(verts float32 [nv,3], tris uint32 [nt,3]), ...] -- what embree_amd.api.make_scene takes.
"""
import os
import xml.etree.ElementTree as ET
from collections import namedtuple

import numpy as np

Scene = namedtuple("Scene", "meshes camera")     # camera: dict(vp, vi, vu, fov) or None


# ------------------------------------------------------------------------------------------------- OBJ
def load_obj(path, split_groups=False):
    verts, groups, cur = [], {}, "default"
    order = []
    with open(path) as f:
        for line in f:
            p = line.split()
            if not p or p[0].startswith("#"):
                continue
            if p[0] == "v":
                verts.append([float(x) for x in p[1:4]])
            elif p[0] in ("usemtl", "g", "o") and split_groups:
                cur = " ".join(p[1:]) or "default"
            elif p[0] == "f":
                idx = []
                for tok in p[1:]:
                    i = int(tok.split("/")[0])
                    idx.append(i - 1 if i > 0 else len(verts) + i)
                if cur not in groups:
                    groups[cur] = []
                    order.append(cur)
                for k in range(1, len(idx) - 1):
                    groups[cur].append([idx[0], idx[k], idx[k + 1]])
    v = np.asarray(verts, np.float32).reshape(-1, 3)
    meshes = []
    for name in order:
        t = np.asarray(groups[name], np.int64).reshape(-1, 3)
        if split_groups:                                  # compact the vertex array per group
            used, inv = np.unique(t, return_inverse=True)
            meshes.append((v[used].copy(), inv.reshape(-1, 3).astype(np.uint32)))
        else:
            meshes.append((v, t.astype(np.uint32)))
    return Scene(meshes, None)


# ------------------------------------------------------------------------------------------------- XML + BIN
def _floats(text):
    return np.array(text.split(), np.float64) if text and text.strip() else np.zeros(0)


def _affine(el):
    """3x4 [A | p] as a 4x4 matrix.  xml_loader.cpp:374-450: attribute forms, or 12 floats row-major (the 16-float form is column-major)."""
    m = np.eye(4)
    a = el.attrib
    if "translate" in a:
        m[:3, 3] = _floats(a["translate"])
    elif "scale" in a:
        m[0, 0], m[1, 1], m[2, 2] = _floats(a["scale"])
    elif any(k in a for k in ("rotate_x", "rotate_y", "rotate_z")):
        ax = [k for k in ("rotate_x", "rotate_y", "rotate_z") if k in a][0]
        r = np.deg2rad(float(a[ax]))
        c, s = np.cos(r), np.sin(r)
        i, j = {"rotate_x": (1, 2), "rotate_y": (2, 0), "rotate_z": (0, 1)}[ax]
        m[i, i], m[i, j], m[j, i], m[j, j] = c, -s, s, c
    else:
        b = _floats(el.text)
        if b.size == 12:
            m[:3, :] = b.reshape(3, 4)
        elif b.size == 16:
            m = b.reshape(4, 4).T.copy()
            m[3, :] = (0, 0, 0, 1)
        elif b.size:
            raise ValueError("AffineSpace with %d values" % b.size)
    return m


class _XML:
    def __init__(self, path):
        self.path = path
        self.bin = None
        bpath = path + ".bin"
        if os.path.exists(bpath):
            self.bin = np.memmap(bpath, np.uint8, "r")
        self.nodes = {}
        self.meshes = []
        self.camera = None

    def array(self, el, dtype, width):
        """loadBinary / inline body (xml_loader.cpp:452-471, 515-534): `size` counts elements (vectors), not scalars."""
        if el is None:
            return np.zeros((0, width), dtype)
        if el.get("ofs") is not None:
            n = int(el.get("size") or 0) or int(el.get("num") or 0)
            ofs, nbytes = int(el.get("ofs")), n * width * 4
            if self.bin is None or ofs + nbytes > self.bin.shape[0]:
                raise ValueError("%s: array [%d, +%d) outside the .bin file" % (self.path, ofs, nbytes))
            return np.frombuffer(self.bin[ofs:ofs + nbytes].tobytes(), dtype).reshape(n, width).copy()
        return _floats(el.text).astype(dtype).reshape(-1, width)

    def geometry(self, el):
        pos = self.array(el.find("positions"), np.float32, 3)
        if el.tag == "TriangleMesh":
            tri = self.array(el.find("triangles"), np.int32, 3).astype(np.uint32)
        else:                                              # QuadMesh: (v0,v1,v3) + (v2,v3,v1)
            q = self.array(el.find("indices"), np.int32, 4).astype(np.uint32)
            tri = np.concatenate([q[:, [0, 1, 3]], q[:, [2, 3, 1]]], 0) if q.size else np.zeros((0, 3), np.uint32)
        return ("geom", pos, tri)

    def node(self, el):
        tag = el.tag
        if tag == "ref":
            n = self.nodes[el.get("id")]
        elif tag in ("TriangleMesh", "QuadMesh"):
            n = self.geometry(el)
        elif tag in ("Group", "scene"):
            n = ("group", [c for c in (self.node(ch) for ch in el) if c is not None])
        elif tag in ("Transform", "TransformAnimation"):
            kids = list(el)
            n = ("xfm", _affine(kids[0]), [c for c in (self.node(ch) for ch in kids[1:]) if c is not None])
        elif tag == "PerspectiveCamera":
            if self.camera is None:
                self.camera = dict(vp=_floats(el.get("from")), vi=_floats(el.get("to")), vu=_floats(el.get("up", "0 1 0")), fov=float(el.get("fov", 90)))
            return None
        else:                                              # lights, materials, curves, ... : not on the triangle path
            return None
        if el.get("id") is not None and tag != "ref":
            self.nodes[el.get("id")] = n
        return n

    def flatten(self, n, m):
        if n[0] == "geom":
            if n[2].shape[0]:
                p = n[1].astype(np.float64) @ m[:3, :3].T + m[:3, 3]
                self.meshes.append((p.astype(np.float32), n[2]))
        elif n[0] == "group":
            for c in n[1]:
                self.flatten(c, m)
        else:
            for c in n[2]:
                self.flatten(c, m @ n[1])


def load_xml(path):
    x = _XML(path)
    root = ET.parse(path).getroot()
    top = x.node(root)
    x.flatten(top, np.eye(4))
    return Scene(x.meshes, x.camera)


def save_xml(path, meshes, camera=None):
    """Writes <path> and <path>.bin in the layout load_xml (and the reference's XMLLoader) read."""
    blob, parts = bytearray(), ['<?xml version="1.0"?>', "<scene>"]
    if camera:
        parts.append('  <PerspectiveCamera name="cam" from="%s" to="%s" up="%s" fov="%g"/>' % (
            " ".join("%.9g" % v for v in camera["vp"]), " ".join("%.9g" % v for v in camera["vi"]), " ".join("%.9g" % v for v in camera["vu"]), camera["fov"]))
    parts.append('  <Group id="0">')
    for i, (v, t) in enumerate(meshes):
        v = np.ascontiguousarray(v, np.float32).reshape(-1, 3)
        t = np.ascontiguousarray(t, np.int32).reshape(-1, 3)
        po = len(blob); blob += v.tobytes()
        to = len(blob); blob += t.tobytes()
        parts += ['    <TriangleMesh id="%d">' % (i + 1), '      <positions ofs="%d" size="%d"/>' % (po, v.shape[0]),
                  '      <triangles ofs="%d" size="%d"/>' % (to, t.shape[0]), "    </TriangleMesh>"]
    parts += ["  </Group>", "</scene>", ""]
    with open(path, "w") as f:
        f.write("\n".join(parts))
    with open(path + ".bin", "wb") as f:
        f.write(bytes(blob))


# ------------------------------------------------------------------------------------------------- ECS
def load_ecs(path, _depth=0):
    """Tutorial command file: whitespace-separated options, '#' comments; paths are relative to the file."""
    if _depth > 8:
        raise ValueError("ecs include depth")
    base = os.path.dirname(os.path.abspath(path))
    toks = []
    for line in open(path):
        toks += line.split("#")[0].split()
    meshes, cam = [], {}
    i = 0
    while i < len(toks):
        t = toks[i]
        if t in ("-i", "-c") and i + 1 < len(toks):
            sub = toks[i + 1] if os.path.isabs(toks[i + 1]) else os.path.join(base, toks[i + 1])
            s = load_ecs(sub, _depth + 1) if t == "-c" else load_scene(sub)
            meshes += s.meshes
            if s.camera and not cam:
                cam = dict(s.camera)
            i += 2
        elif t in ("-vp", "-vi", "-vu") and i + 3 < len(toks):
            cam[t[1:]] = np.array([float(x) for x in toks[i + 1:i + 4]])
            i += 4
        elif t == "-fov" and i + 1 < len(toks):
            cam["fov"] = float(toks[i + 1])
            i += 2
        else:
            i += 1
    if cam:
        cam.setdefault("vu", np.array([0.0, 1.0, 0.0]))
        cam.setdefault("fov", 90.0)
    return Scene(meshes, cam if ("vp" in cam and "vi" in cam) else None)


def load_scene(path):
    ext = os.path.splitext(path)[1].lower()
    if ext == ".ecs":
        return load_ecs(path)
    if ext == ".xml":
        return load_xml(path)
    if ext == ".obj":
        return load_obj(path)
    raise ValueError("unsupported scene file: " + path)


def find_model(name):
    """$EMBREE_MODEL_DIR/<name>/<name>.ecs (how the reference's benchmarks address crown / powerplant), or None."""
    d = os.environ.get("EMBREE_MODEL_DIR")
    if not d:
        return None
    for cand in (os.path.join(d, name, name + ".ecs"), os.path.join(d, name + ".ecs"), os.path.join(d, name, name + ".xml")):
        if os.path.exists(cand):
            return cand
    return None
