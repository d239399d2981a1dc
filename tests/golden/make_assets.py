#!/usr/bin/env python3
"""Copies the reference's own Cornell-box ASSETS (scene data, not source code) into tests/golden/models/ so that the loader -> GPU -> reference
chain can be tested on the GPU box, where /root/reference does not exist.

    python tests/golden/make_assets.py          # run in the build container; the copies are committed

Files: tutorials/models/cornell_box.{ecs,obj,mtl,xml,xml.bin} (6.9 KB together): the .ecs command file names the OBJ and the camera
(configs[1] of BASELINE.json: -vp 278 273 -800 -vi 278 273 0 -vu 0 1 0 -fov 37), the .xml/.xml.bin pair is the same scene in the
reference's XML + binary-blob format (tutorials/common/scenegraph/xml_loader.cpp).
tutorials/triangle_geometry/triangle_geometry.exr (41 KB): the image the reference's own CTest compares the triangle_geometry tutorial with (--compare, tutorial.cpp:646-660);
tests/test_gpu_round5.py runs the tutorial, compiled unmodified against this repository's library (tests/golden/ref_tests.mk), with the same option."""
import os
import shutil

SRC = "/root/reference/tutorials/models"
DST = os.path.join(os.path.dirname(os.path.abspath(__file__)), "models")
FILES = ["cornell_box.ecs", "cornell_box.obj", "cornell_box.mtl", "cornell_box.xml", "cornell_box.xml.bin"]

EXTRA = [("/root/reference/tutorials/triangle_geometry/triangle_geometry.exr", "triangle_geometry.exr")]

if __name__ == "__main__":
    os.makedirs(DST, exist_ok=True)
    for f in FILES:
        shutil.copyfile(os.path.join(SRC, f), os.path.join(DST, f))
        print("copied", f, os.path.getsize(os.path.join(DST, f)), "bytes")
    for src, f in EXTRA:
        shutil.copyfile(src, os.path.join(DST, f))
        os.chmod(os.path.join(DST, f), 0o644)
        print("copied", f, os.path.getsize(os.path.join(DST, f)), "bytes")
