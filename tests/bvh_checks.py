"""Structural checks of a flattened scene (include/rtflat.h), shared by the host-builder tests (no GPU) and the tests of the
device-built BVH: every triangle sits in exactly one leaf, every stored (dequantised) child box strictly contains the
child's padded content, unused slots are inverted boxes, children come after their parents, and the recorded depth is the
real one."""
import sys

import numpy as np


def assert_bvh_sound(f):
    nodes, tris, info = f["nodes"], f["tris"], f["info"]
    seen = np.zeros(info.numTris, np.int32)

    def leaf_range(ref):
        return (ref & 0x7FFFFFFF) >> 3, (ref & 7) + 1

    def box_of_leaf(ref):
        first, count = leaf_range(ref)
        t = tris[first:first + count]
        pts = np.concatenate([t["p0"], t["p0"] + t["e0"], t["p0"] + t["e1"]])
        seen[first:first + count] += 1
        return pts.min(0), pts.max(0)

    def child_box(n, k):
        step = np.ldexp(1.0, n["exp"].astype(np.int32) - 127)
        return n["origin"] + n["qlo"][:, k] * step, n["origin"] + n["qhi"][:, k] * step

    # post-order containment check: each stored (dequantised) child box strictly contains the child's content
    def check(ref, depth):
        if ref & 0x80000000:
            assert leaf_range(ref)[1] <= 8
            return box_of_leaf(ref) + (depth,)
        n = nodes[ref]
        assert 2 <= n["numChildren"] <= 4
        lo, hi, deepest = np.full(3, np.inf), np.full(3, -np.inf), depth
        for k in range(4):
            c = int(n["child"][k])
            if k >= n["numChildren"]:
                assert c == 0xFFFFFFFF and np.all(n["qlo"][:, k] == 255) and np.all(n["qhi"][:, k] == 0)  # inverted: never hit
                continue
            assert c & 0x80000000 or c > ref  # children after parents
            clo, chi, d = check(c, depth + 1)
            blo, bhi = child_box(n, k)
            assert np.all(blo < clo) and np.all(bhi > chi), "a child box must contain the (padded) child"
            lo, hi, deepest = np.minimum(lo, clo), np.maximum(hi, chi), max(deepest, d)
        return lo, hi, deepest

    sys.setrecursionlimit(10000)
    deepest = 0
    for mn in f["model_nodes"]:
        if mn["triCount"] > 0:
            deepest = max(deepest, check(int(mn["blas"]), 0)[2])  # depth counted in 4-wide nodes
    assert np.all(seen == 1)
    assert deepest == info.maxDepthBlas
    assert info.padAbs > 0 and info.numDegenerate == 0
