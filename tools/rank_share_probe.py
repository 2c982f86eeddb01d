"""Probe (not a test): what ONE rank of an N-GPU strong-scaling run of a workload has to do -- the round-robin tile share of rank 0
for N = 1, 2, 4, 8 rendered on this GPU, frames queued back to back as bench.py does -- and how far its frame time is from
(frame time at N = 1) / N.  No gather, no other rank: the per-rank ceiling of the scaling efficiency.
usage: python tools/rank_share_probe.py <workload> [frames]"""
import sys
import time

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
    base = None
    for world in (1, 2, 4, 8):
        r.set_tiles(rdist.assign_tiles(W, H, world)[0])
        for rep in range(2):
            t = time.perf_counter()
            for _ in range(frames * world):
                r.begin_frame(cam, W, H, N)
                r.enqueue_passes(0, N)
            r.synchronize()
            ms = (time.perf_counter() - t) * 1e3 / (frames * world)
        base = base or ms
        mode, _ = r.refill_mode()
        print("%s rank 0 of %d: %.3f ms per frame, ideal %.3f, per-rank efficiency %.3f (traversal mode %d)" % (name, world, ms, base / world, base / world / ms, mode), flush=True)
    r.close()


main()
