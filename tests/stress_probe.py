"""Stress probe (not a test): big frames through the multi-batch and two-slot paths; prints timings and checksums."""
import sys, time, zlib
sys.path.insert(0, '.')
import numpy as np
from tests import scenes


def frame(spec, W, H, N, depth, **opts):
    r = spec.build_product(options=dict(maxDepth=depth, **opts))
    cam = spec.product_camera()
    best = 1e9
    for _ in range(2):
        r.begin_frame(cam, W, H, N)
        t = time.perf_counter()
        r.enqueue_passes(0, N)
        r.synchronize()
        best = min(best, (time.perf_counter() - t) * 1e3)
    c = r.counters()
    acc = r.accumulator()
    rays = c['extendPrimary'] + c['extendSecondary'] + c['shadow']
    print("%s %dx%d %dspp depth%d %s: %.2f ms  %.0f Mrays/s  crc %08x  mean %.5f" % (spec.name, W, H, N, depth, opts, best, rays / best / 1e3,
          zlib.crc32(acc.tobytes()), float(acc.mean())), flush=True)
    r.close()


cube = scenes.cube_scene()
frame(cube, 7680, 4320, 4, 1, pipelining=1)
frame(cube, 7680, 4320, 4, 1, pipelining=0)
frame(cube, 7680, 4320, 3, 12)
hf = scenes.heightfield_scene(2236, 0x5EED0002)
frame(hf, 3840, 2160, 16, 12)
frame(hf, 3840, 2160, 16, 1, pipelining=1)
frame(hf, 3840, 2160, 16, 1, pipelining=0)
