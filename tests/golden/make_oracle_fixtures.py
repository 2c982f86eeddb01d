"""Generates the committed golden vectors of tests/golden/ from the CPU oracle (oracle/), run in the build container:

    python tests/golden/make_oracle_fixtures.py

The reference ships no tests or golden vectors for this path, so these files pin the ORACLE against
regressions; the oracle itself is pinned to the reference by the hand-derived known-answer tests in
tests/test_oracle_kat.py.  Each file holds inputs (scene recipe name, camera, sizes) and expected outputs only.
"""
import os
import sys

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
from oracle import binding as ob  # noqa: E402
from tests import scenes  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def save(name, **kw):
    path = os.path.join(OUT, name)
    np.savez_compressed(path, **kw)
    print(path, os.path.getsize(path))


# (i) rand01 table
xyz = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [0, 0, 1], [1, 2, 3], [255, 255, 0], [256, 0, 7], [1919, 1079, 15], [7679, 4319, 255],
                [0xFFFFFFFF, 0xFFFFFFFF, 0xFFFFFFFF], [123456789, 987654321, 42]], np.uint32)
save("rand01.npz", xyz=xyz, out=np.array([ob.rand01(*r) for r in xyz]))

# (ii) primary rays: corners + centre of 512^2 with the main.cpp camera, samples 0..3
cam = ob.make_camera(**scenes.MAIN_CAMERA)
whs = np.array([[w, h, s] for s in range(4) for (w, h) in [(0, 0), (511, 0), (0, 511), (511, 511), (256, 256)]], np.uint32)
rays = [ob.primary_ray(cam, 512, 512, *[int(v) for v in r]) for r in whs]
save("primary_rays_512.npz", whs=whs, origin=np.array([r[0] for r in rays]), direction=np.array([r[1] for r in rays]), rnd=np.array([r[2] for r in rays]))

# (iii) cube scene 64x64: hit ids, shadow flags, accumulator at 1 spp and 4 spp
spec = scenes.cube_scene()
o = spec.build_oracle()
for n in (1, 4):
    r = o.render(spec.oracle_camera(), 64, 64, n, max_depth=12, aov_depths=2, workers=4)
    save(f"cube_64x64_{n}spp.npz", tri_id=r["tri_id"], shadow=r["shadow"], accumulator=r["accumulator"],
         counters=np.array([r["counters"]["extendPrimary"], r["counters"]["extendSecondary"], r["counters"]["shadow"]] + r["counters"]["hits"], np.uint64))

# (iv) bunny 128x128 primary hit ids
spec = scenes.bunny_scene()
o = spec.build_oracle()
r = o.render(spec.oracle_camera(), 128, 128, 1, max_depth=1, aov_depths=1, workers=8)
save("bunny_128x128_primary.npz", tri_id=r["tri_id"][0, 0], shadow=r["shadow"][0, 0], accumulator=r["accumulator"])

# (v) model tree topology of the multi-model scene (pairing order of createRayTracingHierarchy)
spec = scenes.multi_model_scene()
o = spec.build_oracle()
boxes, topo, model_of = o.model_tree()
r = o.render(spec.oracle_camera(), 96, 80, 2, max_depth=6, aov_depths=2, workers=8)
save("multi_model.npz", boxes=boxes, topo=topo, model_of=model_of, tri_id=r["tri_id"], shadow=r["shadow"], accumulator=r["accumulator"])
