"""Tuning probe (not a test): times a few workloads under the RTHIP_TUNE environment setting and prints the SIMD
utilisation of the node / leaf steps from the work counters."""
import os
import sys
sys.path.insert(0, '.')
from tests import scenes


def run(spec, W, H, N, depth):
    opts = dict(maxDepth=depth)
    for name in ("maxLeafTriangles", "sahBins"):
        if os.environ.get("RT_" + name) is not None:
            opts[name] = int(os.environ["RT_" + name])
    r = spec.build_product(options=opts)
    cam = spec.product_camera()
    best = 1e9
    for rep in range(4):
        r.begin_frame(cam, W, H, N)
        ms = r.render_passes(0, N)
        best = min(best, ms)
    c = r.counters()
    rays = c['extendPrimary'] + c['extendSecondary'] + c['shadow']
    r.set_option("kernelTiming", 1)
    r.begin_frame(cam, W, H, N)
    r.render_passes(0, N)
    kt = r.kernel_times()
    r.set_option("kernelTiming", 0)
    r.set_option("countWork", 1)
    r.begin_frame(cam, W, H, N)
    r.render_passes(0, N)
    c = r.counters()
    un = c['nodesVisited'] / max(1, 64 * c['waveNodeSteps'])
    ul = c['trisTested'] / max(1, 64 * c['waveLeafSteps'])
    print("  %s %dspp: %.2f ms %.0f Mrays/s  extend %.2f shadow %.2f | nodes/ray %.1f tris/ray %.1f node-step util %.3f leaf-step tris/lane-slot %.3f  wave steps node %.2fM leaf %.2fM"
          % (spec.name, N, best, rays / best / 1e3, kt['extend'][0], kt['shadow'][0], c['nodesVisited'] / rays, c['trisTested'] / rays, un, ul,
             c['waveNodeSteps'] / 1e6, c['waveLeafSteps'] / 1e6), flush=True)
    r.close()


print("TUNE", os.environ.get("RTHIP_TUNE"), flush=True)
which = sys.argv[1:] or ["soup1", "soup8", "bunny"]
if "soup1" in which:
    run(scenes.soup_scene(1 << 20, 0x5EED0001, 0.02), 1920, 1080, 1, 4)
if "soup8" in which:
    run(scenes.soup_scene(1 << 20, 0x5EED0001, 0.02), 1920, 1080, 8, 4)
if "bunny" in which:
    run(scenes.bunny_scene(), 1920, 1080, 4, 12)
