"""GPU tests of BASELINE.json configs[2], [3] and [4] at their FULL sizes (2^20-triangle soup at 1080p; the 9,999,392-triangle
height field at 4K 64 spp and at 8K 256 spp, depth cap 12).  The oracle cannot render such frames in a test, so each case
checks, on the full frame: the ray-count identities, nonFiniteAccepts == 0, and bit-equality between differently batched
renders of the same frame (one call / split calls, two slots / one slot); and it pins the frame to the oracle through a
64x64-pixel window rendered by the oracle's variant A -- the reference's own algorithm, a brute-force scan of every
triangle (src/cpu/CPUScene.cpp:187-252) -- compared bit for bit (accumulator after the first sample passes; hit ids and
shadow flags of the window's tile).  This is where the padding of the quantised BVH boxes and the closest-hit cull slack,
both heuristics, meet 10^6 and 10^7 triangles."""
import numpy as np
import pytest

from tests import scenes
from tests.conftest import heartbeat

pytestmark = pytest.mark.gpu

WORKERS = 14


def _identities(c, W, H, S, depth, num_dir):
    """Ray-count identities of a frame (CPUScene.cpp:6-65): one primary ray per camera sample; every hit that survives the
    roulette casts one shadow ray per directional light and, below the depth cap, one bounce ray.  With the hard-coded
    albedo the roulette (depth > 5) only ends a path when rnd.z is exactly 1.0: about 3e-8 per path and depth."""
    assert c["nonFiniteAccepts"] == 0
    assert c["extendPrimary"] == W * H * S
    hits = c["hits"][:depth]
    assert hits[0] > 0.05 * W * H * S, "the camera should see the scene"
    kills_allowed = 0 if depth <= 6 else int(1e-6 * W * H * S) + 8
    lost_shadow = num_dir * sum(hits) - c["shadow"]
    assert 0 <= lost_shadow <= num_dir * kills_allowed and lost_shadow % num_dir == 0
    lost_extend = sum(hits[: depth - 1]) - c["extendSecondary"]
    assert 0 <= lost_extend <= kills_allowed


def _window_tiles(win, W, ts=64):
    tx = (W + ts - 1) // ts
    x0, y0, x1, y1 = win
    return np.array([ty * tx + txx for ty in range(y0 // ts, (y1 - 1) // ts + 1) for txx in range(x0 // ts, (x1 - 1) // ts + 1)], np.uint32)


def _check_window_against_variant_a(spec, oracle, r, W, H, S, depth, win, prefix):
    """Renders the first `prefix` sample passes of an S-sample frame and compares the window with the oracle's brute-force
    render of the same passes, bit for bit; then the window's tile alone with AOVs: hit ids and shadow flags."""
    x0, y0, x1, y1 = win
    with heartbeat("variant A, %dx%d window" % (x1 - x0, y1 - y0)):
        want = oracle.render(spec.oracle_camera(), W, H, S, max_depth=depth, window=win, samples=(0, prefix), workers=WORKERS, aov_depths=2)
    assert want["counters"]["nonFiniteAccepts"] == 0
    r.begin_frame(spec.product_camera(), W, H, S)
    r.render_passes(0, prefix)
    acc = r.accumulator()
    got, ref = acc[y0:y1, x0:x1], want["accumulator"][y0:y1, x0:x1]
    assert ref.any(), "the window should show the scene"
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32)), f"window differs from variant A in {(got != ref).any(axis=2).sum()} pixels"
    # hit ids / shadow flags: the same passes on the window's tile(s) only, with AOVs
    r.set_option("aovDepths", 2)
    r.set_tiles(_window_tiles(win, W))
    r.begin_frame(spec.product_camera(), W, H, S)
    for s in range(prefix):
        r.render_passes(s, 1)
        for d in range(2):
            tri, sh = r.aov(s, d)
            assert np.array_equal(tri[y0:y1, x0:x1], want["tri_id"][s, d][y0:y1, x0:x1]), (s, d)
            assert np.array_equal(sh[y0:y1, x0:x1], want["shadow"][s, d][y0:y1, x0:x1]), (s, d)
    r.set_option("aovDepths", 0)
    r.set_tiles(np.zeros(0, np.uint32))
    return acc


def test_config2_soup_1m_triangles_1080p_three_bounces():
    spec = scenes.soup1m_scene()
    W, H, S, depth = 1920, 1080, 1, 4
    r = spec.build_product(options=dict(maxDepth=depth))
    assert r.flat()["info"].numTris == 1 << 20
    oracle = spec.build_oracle()
    win = (928, 508, 992, 572)
    acc = _check_window_against_variant_a(spec, oracle, r, W, H, S, depth, win, 1)
    # the full frame: the passes above were the whole frame (1 spp); identities and determinism across the two-slot schedule
    r.begin_frame(spec.product_camera(), W, H, S)
    r.render_passes(0, S)
    c = r.counters()
    _identities(c, W, H, S, depth, len(spec.dir_lights))
    again = r.accumulator()
    assert np.array_equal(acc.view(np.uint32), again.view(np.uint32))
    r.set_option("pipelining", 0)
    r.begin_frame(spec.product_camera(), W, H, S)
    r.render_passes(0, S)
    assert np.array_equal(again.view(np.uint32), r.accumulator().view(np.uint32))
    assert r.counters() == c
    assert again.min() >= 0 and np.isfinite(again).all()
    r.close()


@pytest.fixture(scope="module")
def hf10m():
    spec = scenes.hf10m_scene()
    r = spec.build_product(options=dict(maxDepth=12))
    assert r.flat()["info"].numTris == 9999392
    oracle = spec.build_oracle()
    yield spec, r, oracle
    r.close()


def test_config3_heightfield_10m_triangles_4k_64spp(hf10m):
    spec, r, oracle = hf10m
    W, H, S, depth = 3840, 2160, 64, 12
    win = (1888, 1240, 1952, 1304)
    _check_window_against_variant_a(spec, oracle, r, W, H, S, depth, win, 1)
    # all 64 samples of a 4x4 corner of that window against variant A: the batch layout the benchmark runs
    # (1 pixel x 64 samples per wave) pinned to the oracle directly
    small = (win[0], win[1], win[0] + 4, win[1] + 4)
    with heartbeat("variant A, 4x4 window, 64 spp"):
        want = oracle.render(spec.oracle_camera(), W, H, S, max_depth=depth, window=small, workers=WORKERS)
    r.begin_frame(spec.product_camera(), W, H, S)
    r.render_passes(0, S)
    full = r.accumulator()
    c = r.counters()
    _identities(c, W, H, S, depth, len(spec.dir_lights))
    x0, y0, x1, y1 = small
    assert np.array_equal(full[y0:y1, x0:x1].view(np.uint32), want["accumulator"][y0:y1, x0:x1].view(np.uint32))
    # determinism across passesPerCall: 64 passes in one call (1 x 64 layout) == calls of 5 passes (32-pixel x 5 layout)
    r.begin_frame(spec.product_camera(), W, H, S)
    for s in range(0, S, 5):
        r.render_passes(s, min(5, S - s))
    assert np.array_equal(full.view(np.uint32), r.accumulator().view(np.uint32))
    assert r.counters() == c
    assert full.min() >= 0 and np.isfinite(full).all()
    # every traversal mode (per-lane with early / late refill, packets for camera and first shadow rays) at full size: same bits
    for mode in (0, 1, 2, 3):
        r.set_option("refillMode", mode)
        r.begin_frame(spec.product_camera(), W, H, S)
        assert r.refill_mode()[0] == mode
        r.render_passes(0, S)
        assert np.array_equal(full.view(np.uint32), r.accumulator().view(np.uint32)), mode
        assert r.counters() == c, mode
    r.set_option("refillMode", -1)


def test_config4_heightfield_8k_256spp_depth12(hf10m):
    spec, r, oracle = hf10m
    W, H, S, depth = 7680, 4320, 256, 12
    win = (3808, 2480, 3872, 2544)
    _check_window_against_variant_a(spec, oracle, r, W, H, S, depth, win, 1)
    r.begin_frame(spec.product_camera(), W, H, S)
    r.render_passes(0, S)
    full = r.accumulator()
    c = r.counters()
    _identities(c, W, H, S, depth, len(spec.dir_lights))
    assert full.min() >= 0 and np.isfinite(full).all()
    # all 256 sample passes of a 2x2 corner of the window against variant A (round 3): the whole samp-ordered accumulation
    # of this configuration pinned to the brute-force oracle, not only its first pass
    small = (win[0] + 30, win[1] + 30, win[0] + 32, win[1] + 32)
    with heartbeat("variant A, 2x2 window, 256 spp"):
        want = oracle.render(spec.oracle_camera(), W, H, S, max_depth=depth, window=small, workers=WORKERS)
    x0, y0, x1, y1 = small
    assert want["accumulator"][y0:y1, x0:x1].any()
    assert np.array_equal(full[y0:y1, x0:x1].view(np.uint32), want["accumulator"][y0:y1, x0:x1].view(np.uint32))
    # the same frame on one slot, in calls of 100 passes: other batches, same bits
    r.set_option("pipelining", 0)
    r.begin_frame(spec.product_camera(), W, H, S)
    for s in range(0, S, 100):
        r.render_passes(s, min(100, S - s))
    assert np.array_equal(full.view(np.uint32), r.accumulator().view(np.uint32))
    assert r.counters() == c
    r.set_option("pipelining", 1)
