"""Scene loaders: OBJ, Embree XML + .bin, .ecs command files -- CPU only.
Checked against the reference's own assets when /root/reference is mounted (this container), otherwise against files the tests write."""
import os

import numpy as np
import pytest

from embree_amd import loaders as Ld, workloads as W

REF_MODELS = "/root/reference/tutorials/models"


def tri_soup(meshes):
    a = np.concatenate([np.asarray(v, np.float32)[np.asarray(t, np.int64)].reshape(-1, 9) for v, t in meshes])
    return a[np.lexsort(a.T[::-1])]


def test_xml_bin_round_trip(tmp_path):
    meshes = W.synthetic_crown(num_phi=8)
    cam = dict(vp=[1, 2, 3], vi=[0, 0, 0], vu=[0, 1, 0], fov=37.0)
    p = str(tmp_path / "scene.xml")
    Ld.save_xml(p, meshes, cam)
    s = Ld.load_scene(p)
    assert len(s.meshes) == len(meshes)
    for (v, t), (v2, t2) in zip(meshes, s.meshes):
        assert v2.dtype == np.float32 and t2.dtype == np.uint32
        assert (v2 == v).all() and (t2 == t).all()
    assert np.allclose(s.camera["vp"], [1, 2, 3]) and s.camera["fov"] == 37.0


def test_xml_transforms_refs_quads_inline(tmp_path):
    p = tmp_path / "g.xml"
    p.write_text("""<?xml version="1.0"?>
<scene>
  <Group>
    <TriangleMesh id="7"><positions>0 0 0  1 0 0  0 1 0</positions><triangles>0 1 2</triangles></TriangleMesh>
    <Transform><AffineSpace translate="10 0 0"/><ref id="7"/></Transform>
    <Transform><AffineSpace>2 0 0 0  0 2 0 5  0 0 2 0</AffineSpace>
      <Transform><AffineSpace rotate_z="90"/><ref id="7"/></Transform>
    </Transform>
    <QuadMesh><positions>0 0 1  1 0 1  1 1 1  0 1 1</positions><indices>0 1 2 3</indices></QuadMesh>
    <PointLight/>
  </Group>
</scene>""")
    s = Ld.load_xml(str(p))
    assert len(s.meshes) == 4 and s.camera is None
    v0 = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0]], np.float32)
    assert (s.meshes[0][0] == v0).all()
    assert (s.meshes[1][0] == v0 + [10, 0, 0]).all()
    # rotate_z(90): (x,y) -> (-y,x); then scale 2 and +5 in y
    want = np.array([[0, 5, 0], [0, 7, 0], [-2, 5, 0]], np.float32)
    assert np.allclose(s.meshes[2][0], want, atol=1e-6)
    assert (s.meshes[3][1] == np.array([[0, 1, 3], [2, 3, 1]], np.uint32)).all()       # Embree's quad split


def test_obj_negative_indices_and_polygons(tmp_path):
    p = tmp_path / "m.obj"
    p.write_text("# c\nv 0 0 0\nv 1 0 0\nv 1 1 0\nv 0 1 0\nusemtl a\nf -4 -3 -2 -1\nv 0 0 1\nusemtl b\nf 1/1/1 2/2/2 5/3/3\n")
    one = Ld.load_obj(str(p))
    assert len(one.meshes) == 1 and (one.meshes[0][1] == np.array([[0, 1, 2], [0, 2, 3], [0, 1, 4]], np.uint32)).all()
    two = Ld.load_obj(str(p), split_groups=True)
    assert [t.shape[0] for _, t in two.meshes] == [2, 1]
    assert (tri_soup(two.meshes) == tri_soup(one.meshes)).all()


def test_ecs_includes_and_camera(tmp_path):
    Ld.save_xml(str(tmp_path / "a.xml"), [W.cube_and_plane()[0]])
    (tmp_path / "inner.ecs").write_text("-i a.xml   # the cube\n")
    (tmp_path / "top.ecs").write_text("-c inner.ecs\n-vp 1.5 1.5 -1.5 -vi 0 0 0 -fov 90\n-pointlight 1 2 3 4 5 6\n")
    s = Ld.load_scene(str(tmp_path / "top.ecs"))
    assert len(s.meshes) == 1 and s.meshes[0][1].shape[0] == 12
    assert np.allclose(s.camera["vp"], [1.5, 1.5, -1.5]) and np.allclose(s.camera["vu"], [0, 1, 0]) and s.camera["fov"] == 90


def test_bad_bin_offsets_are_rejected(tmp_path):
    p = tmp_path / "b.xml"
    p.write_text('<scene><TriangleMesh><positions ofs="0" size="100"/><triangles ofs="0" size="1"/></TriangleMesh></scene>')
    (tmp_path / "b.xml.bin").write_bytes(b"\0" * 36)
    with pytest.raises(ValueError):
        Ld.load_xml(str(p))


@pytest.mark.skipif(not os.path.isdir(REF_MODELS), reason="the reference's tutorial models are only mounted in the build container")
def test_reference_cornell_box_assets(golden_dir):
    """cornell_box.ecs -> .obj, cornell_box.xml + .xml.bin: the same 34 triangles, equal to the committed fixture."""
    ecs = Ld.load_scene(os.path.join(REF_MODELS, "cornell_box.ecs"))
    xml = Ld.load_scene(os.path.join(REF_MODELS, "cornell_box.xml"))
    g = np.load(os.path.join(golden_dir, "cornell_box.npz"))
    fixture = tri_soup([(g["verts"], g["tris"])])
    assert W.num_triangles(ecs.meshes) == W.num_triangles(xml.meshes) == 34 and len(xml.meshes) == 8
    assert (tri_soup(ecs.meshes) == fixture).all() and (tri_soup(xml.meshes) == fixture).all()
    for cam in (ecs.camera, xml.camera):                     # cornell_box.ecs:3
        assert np.allclose(cam["vp"], [278, 273, -800]) and np.allclose(cam["vi"], [278, 273, 0]) and cam["fov"] == 37
