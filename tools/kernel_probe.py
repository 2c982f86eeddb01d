"""Probe (not a test): frame time with batches overlapped / one at a time, and per-kernel-family device time when every
kernel runs alone.   usage: python tools/kernel_probe.py <workload> [frames]   (workloads: bench.py's)
RT_PROBE_MATERIALS=1: every mesh gets planar texture coordinates and a 256x256 albedo texture, scene option "materials" on;
RT_PROBE_POINT_SHADOWS=1: point lights cast shadow rays."""
import sys
import time

sys.path.insert(0, ".")
import bench  # noqa: E402


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "hf10m4"
    frames = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    wl = bench.workloads()[name]
    spec = wl["spec"]()
    W, H, N, depth = wl["width"], wl["height"], wl["spp"], wl["max_depth"]
    import os
    N = int(os.environ.get("RT_PROBE_SPP", N))
    depth = int(os.environ.get("RT_PROBE_DEPTH", depth))
    extra = {}
    if os.environ.get("RT_PROBE_MATERIALS"):
        import numpy as np
        import raytracer_amd as ra
        textured = []
        for mesh, t in spec.meshes:
            p, i = mesh.arrays()
            m = ra.Mesh.from_arrays(p, i, np.ascontiguousarray(p[:, [0, 2]] * np.float32(0.45) + np.float32(0.5)))
            m.set_material(ra.capi.MAT_FACTORS, (0.9, 0.9, 0.9), (0, 0, 0), ra.scenes.checker_texture(256, 256, cells=32))
            textured.append((m, t))
        spec.meshes = textured
        extra["materials"] = 1
    if os.environ.get("RT_PROBE_POINT_SHADOWS"):
        extra["pointShadows"] = 1
    for kv in filter(None, os.environ.get("RT_PROBE_OPTS", "").split(",")):  # e.g. RT_PROBE_OPTS=maxLeafTriangles=8,deviceBvh=1
        k, v = kv.split("=")
        extra[k] = float(v)
    for pipelining in ((0,) if os.environ.get("RT_PROBE_ALONE") else (1, 0)):
        r = spec.build_product(options=dict(maxDepth=depth, pipelining=pipelining, **extra))
        cam = spec.product_camera()
        for rep in range(2):
            t = time.perf_counter()
            for _ in range(frames):
                r.begin_frame(cam, W, H, N)
                r.enqueue_passes(0, N)
            r.synchronize()
            ms = (time.perf_counter() - t) * 1e3 / frames
        c = r.counters()
        rays = c["extendPrimary"] + c["extendSecondary"] + c["shadow"]
        mode, cal = r.refill_mode()
        print("%s pipelining=%d: %.3f ms per frame, %.1f Mrays/s   (traversal mode %d, list launches on %d blocks per CU; calibration low/high/packet/camera-packet %.3f / %.3f / %.3f / %.3f ms; build host %.2f s device %.1f ms, %d nodes)" % (
            name, pipelining, ms, rays / ms / 1e3, mode, r.grid_blocks(), cal[0], cal[1], cal[2], cal[3], *r.build_times(), int(r.flat()["info"].numNodes)), flush=True)
        if not pipelining:
            r.set_option("kernelTiming", 1)
            r.begin_frame(cam, W, H, N)
            r.render_passes(0, N)
            print("   kernels alone (ms, launches):", {k: (round(v[0], 3), v[1]) for k, v in r.kernel_times().items()}, flush=True)
            r.set_option("kernelTiming", 0)
            r.set_option("countWork", 1)
            r.begin_frame(cam, W, H, N)
            r.render_passes(0, N)
            c = r.counters()
            print("   nodes/ray %.2f tris/ray %.2f node-step lane use %.3f leaf-step tris per lane slot %.3f hits %s" % (
                c["nodesVisited"] / rays, c["trisTested"] / rays, c["nodesVisited"] / max(1, 64 * c["waveNodeSteps"]),
                c["trisTested"] / max(1, 64 * c["waveLeafSteps"]), c["hits"][:6]), flush=True)
        r.close()


main()
