"""Probe (not a test): what more launch batches in flight would buy -- R renderers (contexts) on one GPU, each with its own
two slots, all rendering the same workload; reports the aggregate frame rate.
usage: python tools/multi_ctx_probe.py <workload> <renderers> [frames]"""
import os
import sys
import time

sys.path.insert(0, ".")
import bench  # noqa: E402


def main():
    name = sys.argv[1]
    R = int(sys.argv[2])
    frames = int(sys.argv[3]) if len(sys.argv) > 3 else 8
    wl = bench.workloads()[name]
    spec = wl["spec"]()
    W, H, N, depth = wl["width"], wl["height"], wl["spp"], wl["max_depth"]
    rs = [spec.build_product(options=dict(maxDepth=depth)) for _ in range(R)]
    share = int(os.environ.get("RT_SHARE", "1"))  # every renderer renders rank 0's tiles of a `share`-GPU run
    if share > 1:
        from raytracer_amd import dist as rdist
        for r in rs:
            r.set_tiles(rdist.assign_tiles(W, H, share)[0])
    cam = spec.product_camera()
    for rep in range(3):
        t = time.perf_counter()
        for _ in range(frames):
            for r in rs:
                r.begin_frame(cam, W, H, N)
                r.enqueue_passes(0, N)
        for r in rs:
            r.synchronize()
        ms = (time.perf_counter() - t) * 1e3 / (frames * R)
    c = rs[0].counters()
    rays = c["extendPrimary"] + c["extendSecondary"] + c["shadow"]
    print("%s x%d renderers: %.3f ms per frame, %.1f Mrays/s" % (name, R, ms, rays / ms / 1e3), flush=True)
    for r in rs:
        r.close()


main()
