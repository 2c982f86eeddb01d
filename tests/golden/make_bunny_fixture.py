"""Generates tests/golden/bunny_mesh.npz from the reference's shipped res/models/stanford-bunny.obj.

The file is DATA (vertex positions and triangle indices of the public Stanford bunny, as the OBJ lists them),
repacked so that GPU-box runs, which have no /root/reference, can load the same scene.  Run in the build
container:  python tests/golden/make_bunny_fixture.py
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import raytracer_amd as ra  # noqa: E402

src = "/root/reference/res/models/stanford-bunny.obj"
mesh = ra.Mesh.load_obj(src)
p, i = mesh.arrays()
assert p.shape == (35947, 3) and i.shape == (69451, 3)
out = os.path.join(os.path.dirname(__file__), "bunny_mesh.npz")
np.savez_compressed(out, positions=p, indices=i.astype(np.uint32))
print(out, os.path.getsize(out))
