"""Probe (not a test): how the ORDER of the local tile list changes the frame time.  A frame that is cut into two batches takes the
first and the second half of the list; row-major order makes those the upper and the lower half of the image, which cost very
different amounts on a terrain seen from above.
usage: python tools/tile_order_probe.py <workload> [frames]"""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import bench  # noqa: E402
from raytracer_amd import dist as rdist  # noqa: E402


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "hf10m"
    frames = int(sys.argv[2]) if len(sys.argv) > 2 else 6
    wl = bench.workloads()[name]
    spec = wl["spec"]()
    W, H, N, depth = wl["width"], wl["height"], wl["spp"], wl["max_depth"]
    r = spec.build_product(options=dict(maxDepth=depth, deviceBvh=1))
    cam = spec.product_camera()
    ids = rdist.assign_tiles(W, H, 1)[0]
    tx, ty = rdist.tile_grid(W, H)
    rows = ids.reshape(ty, tx)
    orders = {
        "row-major": ids,
        "even tiles, then odd tiles": np.concatenate([ids[0::2], ids[1::2]]),
        "even rows, then odd rows": np.concatenate([rows[0::2].ravel(), rows[1::2].ravel()]),
        "left half, then right half (rows kept)": np.concatenate([rows[:, : tx // 2].ravel(), rows[:, tx // 2:].ravel()]),
    }
    for rep in range(2):
        for label, order in orders.items():
            r.set_tiles(np.ascontiguousarray(order, np.uint32))
            for k in range(2):
                t = time.perf_counter()
                for _ in range(frames):
                    r.begin_frame(cam, W, H, N)
                    r.enqueue_passes(0, N)
                r.synchronize()
                ms = (time.perf_counter() - t) * 1e3 / frames
            print("%s, %s: %.3f ms per frame (traversal mode %d)" % (name, label, ms, r.refill_mode()[0]), flush=True)
    r.close()


main()
