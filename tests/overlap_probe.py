"""Experiment (not a test): does running two half-frames concurrently (two contexts, two host threads) beat one full frame?"""
import sys, time, threading
sys.path.insert(0, '.')
from tests import scenes


def run(spec_fn, W, H, spp, depth, K, nctx):
    specs = [spec_fn() for _ in range(nctx)]
    rs = [s.build_product(options=dict(maxDepth=depth)) for s in specs]
    cam = specs[0].product_camera()

    def work(r, n):
        for _ in range(n):
            r.begin_frame(cam, W, H, spp // nctx)
            r.enqueue_passes(0, spp // nctx)
        r.synchronize()
    for r in rs:
        work(r, 3)
    t0 = time.perf_counter()
    ts = [threading.Thread(target=work, args=(r, K)) for r in rs]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    dt = (time.perf_counter() - t0) / K * 1e3
    rays = sum((lambda c: c['extendPrimary'] + c['extendSecondary'] + c['shadow'])(r.counters()) for r in rs)
    print("%s %dx%d %dspp depth%d contexts=%d: %.3f ms per full frame, %.0f Mrays/s" % (specs[0].name, W, H, spp, depth, nctx, dt, rays / dt / 1e3), flush=True)
    for r in rs:
        r.close()


for n in (1, 2, 4):
    run(scenes.cube_scene, 1920, 1080, 16, 1, 30, n)
for n in (1, 2):
    run(scenes.bunny_scene, 1920, 1080, 16, 1, 10, n)
for n in (1, 2, 4):
    run(lambda: scenes.soup_scene(1 << 20, 0x5EED0001, 0.02), 1920, 1080, 8, 4, 5, n)
