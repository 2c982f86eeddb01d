"""Generates tests/golden/glb_meshes.npz from the reference's shipped res/models/box.glb and town_hall.glb.

The file is DATA (vertex positions, texture coordinates, triangle indices and the material factors of every mesh primitive,
in file order, as the product's glTF reader returns them), repacked so that GPU-box runs, which have no /root/reference, can build the same scenes.  The
arrays are cross-checked here against an independent pure-Python parse of the .glb container (tests/glb_util.py).
Run in the build container:  python tests/golden/make_glb_fixture.py
"""
import os
import sys

import numpy as np

ROOT = os.path.join(os.path.dirname(__file__), "..", "..")
sys.path.insert(0, ROOT)
import raytracer_amd as ra  # noqa: E402
from tests import glb_util  # noqa: E402

out = {}
for name in ("box", "town_hall"):
    src = "/root/reference/res/models/%s.glb" % name
    meshes = ra.Mesh.load_all(src)
    want = glb_util.parse_glb(open(src, "rb").read())
    shading = glb_util.parse_glb_shading(open(src, "rb").read())
    assert len(meshes) == len(want) == len(shading)
    out[name + "_count"] = np.int64(len(meshes))
    for k, (m, (wp, wi)) in enumerate(zip(meshes, want)):
        p, i = m.arrays()
        assert np.array_equal(p, wp) and np.array_equal(i, wi)
        out["%s_%d_positions" % (name, k)] = p
        out["%s_%d_indices" % (name, k)] = i.astype(np.uint32)
        uv, base, emissive = shading[k]
        kind, got_base, got_emissive = m.material()
        assert kind == 1 and np.array_equal(m.texcoords(), uv) and np.array_equal(got_base, base) and np.array_equal(got_emissive, emissive)
        out["%s_%d_texcoords" % (name, k)] = uv
        out["%s_%d_base_color" % (name, k)] = base
        out["%s_%d_emissive" % (name, k)] = emissive
path = os.path.join(os.path.dirname(__file__), "glb_meshes.npz")
np.savez_compressed(path, **out)
print(path, os.path.getsize(path))
