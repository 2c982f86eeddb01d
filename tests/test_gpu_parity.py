"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle on the same seeded inputs.
Bar: bit-exact hit ids, shadow flags AND accumulated radiance (integer/index work and, because the device
restates the reference's fp32 operation order, the floating-point results too)."""
import numpy as np
import pytest

import raytracer_amd as ra
from oracle import binding as ob
from tests import scenes

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def probe():
    r = scenes.cube_scene().build_product()
    yield ra.Probe(r.context), r
    r.close()


def test_probe_rand01_bitexact(probe):
    p, _ = probe
    rng = np.random.default_rng(1)
    xyz = rng.integers(0, 2**32, size=(20000, 3), dtype=np.uint64).astype(np.uint32)
    xyz[:64] = [[i, 0, 0] for i in range(64)]
    got = p.rand01(xyz)
    want = np.array([ob.rand01(*row) for row in xyz[:3000]])
    assert np.array_equal(got[:3000].view(np.uint32), want.view(np.uint32))


def test_probe_sincos_bitexact(probe):
    p, _ = probe
    x = np.concatenate([np.linspace(0, 2 * np.pi, 5000, dtype=np.float32), np.float32(2 * np.pi) * np.random.default_rng(2).random(5000, np.float32)])
    s, c = p.sincos(x)
    want = np.array([ob.sincos(v) for v in x])
    assert np.array_equal(s.view(np.uint32), want[:, 0].astype(np.float32).view(np.uint32))
    assert np.array_equal(c.view(np.uint32), want[:, 1].astype(np.float32).view(np.uint32))


@pytest.mark.parametrize("lens", [False, True])
def test_probe_primary_rays_bitexact(probe, lens):
    p, _ = probe
    cam = dict(scenes.MAIN_CAMERA)
    if lens:
        cam.update(f=0.7, A=0.35, S_O=20)  # Camera.h defaults: thin lens active
    rng = np.random.default_rng(3)
    W, H = 1920, 1080
    whs = np.stack([rng.integers(0, W, 2000), rng.integers(0, H, 2000), rng.integers(0, 64, 2000)], 1).astype(np.uint32)
    o, d = p.primary_rays(ra.make_camera(**cam), W, H, whs)
    ocam = ob.make_camera(**cam)
    for k in range(whs.shape[0]):
        oo, dd, _ = ob.primary_ray(ocam, W, H, *[int(v) for v in whs[k]])
        assert np.array_equal(o[k].view(np.uint32), oo.view(np.uint32)), (k, whs[k])
        assert np.array_equal(d[k].view(np.uint32), dd.view(np.uint32)), (k, whs[k])


def test_probe_bounce_bitexact(probe):
    p, _ = probe
    rng = np.random.default_rng(4)
    n = rng.normal(size=(3000, 3)).astype(np.float32)
    n /= np.linalg.norm(n, axis=1, keepdims=True).astype(np.float32)
    n[:6] = [[1, 0, 0], [0, 1, 0], [0, 0, 1], [-1, 0, 0], [0.1, 0.9, 0.42], [0.099999, 0.5, 0.86]]
    r = rng.random((3000, 3), np.float32)
    got = p.bounce(n, r)
    want = np.array([ob.bounce_dir(n[k], r[k]) for k in range(n.shape[0])])
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


@pytest.mark.parametrize("name,W,H,N,depth", [("cube", 256, 192, 4, 12), ("cube", 250, 130, 2, 1), ("cube", 512, 512, 1, 1), ("bunny", 160, 160, 2, 4),
                                             ("soup", 200, 200, 2, 6), ("hf", 192, 128, 2, 5), ("multi", 192, 160, 2, 6),
                                             ("box_glb", 200, 150, 3, 8), ("town_hall", 200, 150, 3, 8)])
def test_render_matches_oracle_bitexact(name, W, H, N, depth):
    spec = {"cube": scenes.cube_scene, "bunny": scenes.bunny_scene, "soup": scenes.soup_scene, "hf": scenes.heightfield_scene,
            "multi": scenes.multi_model_scene, "box_glb": scenes.box_glb_scene, "town_hall": scenes.town_hall_scene}[name]()
    aov = min(depth, 3)
    oracle = spec.build_oracle()
    want = oracle.render(spec.oracle_camera(), W, H, N, max_depth=depth, aov_depths=aov, workers=14)
    assert want["counters"]["nonFiniteAccepts"] == 0
    r = spec.build_product(options=dict(maxDepth=depth, aovDepths=aov))
    r.begin_frame(spec.product_camera(), W, H, N)
    bad = []
    for s in range(N):
        r.render_passes(s, 1)
        for d in range(aov):
            tri, sh = r.aov(s, d)
            nt, ns = int((tri != want["tri_id"][s, d]).sum()), int((sh != want["shadow"][s, d]).sum())
            if nt or ns:
                ys, xs = np.nonzero(tri != want["tri_id"][s, d])
                bad.append((s, d, nt, ns, [(int(x), int(y), int(tri[y, x]), int(want["tri_id"][s, d][y, x])) for y, x in zip(ys[:4], xs[:4])]))
    assert not bad, f"hit ids / shadow flags differ (samp, depth, #tri, #shadow, first (x,y,got,want)): {bad}"
    acc = r.accumulator()
    cnt = r.counters()
    assert cnt["nonFiniteAccepts"] == 0
    oc = want["counters"]
    assert cnt["extendPrimary"] == oc["extendPrimary"]
    assert cnt["hits"][:depth] == oc["hits"][:depth]
    assert cnt["shadow"] == oc["shadow"]
    # the reference traces (and discards) one more closest-hit ray for every path that reaches the depth cap
    # (src/cpu/CPUScene.cpp:8 runs before the check at :15); the device never casts that ray
    alive_at_cap = sum(oc["hits"][: depth]) and (oc["extendPrimary"] + oc["extendSecondary"] - cnt["extendPrimary"] - cnt["extendSecondary"])
    assert 0 <= alive_at_cap <= oc["hits"][depth - 1]
    assert np.array_equal(acc.view(np.uint32), want["accumulator"].view(np.uint32)), \
        f"accumulator differs: max abs {np.abs(acc - want['accumulator']).max()}, pixels {(acc != want['accumulator']).any(axis=2).sum()}"
    img = r.image()
    np.testing.assert_allclose(img, want["image"], rtol=2e-6, atol=1e-7)
    r.close()


LIGHT_SETUPS = {
    # no directional light: no shadow launch exists, the final light fold runs in k_shade
    "points_only": dict(dir_lights=[], point_lights=[((1.6, 0.4, 0.9), (1, 0.5, 0.25), 3.0), ((0, 2.5, 0), (0.2, 0.9, 0.4), 4.0)]),
    # three directional lights (one shining upwards from below) and two point lights in range: the fold that the
    # shadow kernel performs at the last depth must add them in the reference's order
    "mixed": dict(dir_lights=[(scenes.main_light_dir(), (1, 1, 1)), (np.array([0.6, -0.64, 0.48], np.float32), (0.5, 0.25, 0.125)),
                              (np.array([0.05, 0.99, 0.13], np.float32), (0.3, 0.3, 0.9))],
                  point_lights=[((1.6, 0.4, 0.9), (1, 0.5, 0.25), 3.0), ((0, 2.5, 0), (0.2, 0.9, 0.4), 4.0)]),
    "no_lights": dict(dir_lights=[], point_lights=[]),
    # two and four directional lights: from depth 2 on the shadow launches take one item per (path, light) and write one byte
    # of mask[] per light (k_trace<SHADOW>, splitLg = 1 and 2), which k_shade folds back into the occlusion bits
    "two_dir": dict(dir_lights=[(scenes.main_light_dir(), (1, 1, 1)), (np.array([0.6, -0.64, 0.48], np.float32), (0.5, 0.25, 0.125))], point_lights=[]),
    "four_dir": dict(dir_lights=[(scenes.main_light_dir(), (1, 1, 1)), (np.array([0.6, -0.64, 0.48], np.float32), (0.5, 0.25, 0.125)),
                                 (np.array([0.05, 0.99, 0.13], np.float32), (0.3, 0.3, 0.9)), (np.array([-0.7, -0.5, -0.5], np.float32), (0.2, 0.7, 0.4))],
                     point_lights=[((1.6, 0.4, 0.9), (1, 0.5, 0.25), 3.0)]),
}


@pytest.mark.gpu
@pytest.mark.parametrize("setup", sorted(LIGHT_SETUPS))
@pytest.mark.parametrize("depth", [1, 2, 7])
def test_light_configurations_match_oracle_bitexact(setup, depth):
    base = scenes.multi_model_scene()
    spec = scenes.SceneSpec("lights_" + setup, base.meshes, base.camera, **LIGHT_SETUPS[setup])
    W, H, N = 160, 120, 3
    aov = min(depth, 2)
    want = spec.build_oracle().render(spec.oracle_camera(), W, H, N, max_depth=depth, aov_depths=aov, workers=14)
    assert want["counters"]["nonFiniteAccepts"] == 0
    r = spec.build_product(options=dict(maxDepth=depth, aovDepths=aov))
    r.begin_frame(spec.product_camera(), W, H, N)
    for s in range(N):
        r.render_passes(s, 1)
        for d in range(aov):
            tri, sh = r.aov(s, d)
            assert np.array_equal(tri, want["tri_id"][s, d])
            if LIGHT_SETUPS[setup]["dir_lights"]:
                assert np.array_equal(sh, want["shadow"][s, d])
    assert r.counters()["shadow"] == want["counters"]["shadow"]
    assert np.array_equal(r.accumulator().view(np.uint32), want["accumulator"].view(np.uint32))
    r.close()


def _render(spec, W, H, N, depth, options=None, tiles=None, passes_per_call=None):
    r = spec.build_product(options=dict(maxDepth=depth, **(options or {})))
    if tiles is not None:
        r.set_tiles(tiles)
    r.begin_frame(spec.product_camera(), W, H, N)
    step = passes_per_call or N
    for s in range(0, N, step):
        r.render_passes(s, min(step, N - s))
    return r


def test_multi_model_with_fixed_edge_indexing_matches_the_fixed_oracle():
    spec = scenes.multi_model_scene()
    W, H, N, depth = 160, 128, 2, 6
    want = spec.build_oracle(fix_edge_offset=True).render(spec.oracle_camera(), W, H, N, max_depth=depth, workers=14)
    r = _render(spec, W, H, N, depth, options=dict(referenceEdgeIndexing=0))
    assert np.array_equal(r.accumulator().view(np.uint32), want["accumulator"].view(np.uint32))
    faithful = spec.build_oracle().render(spec.oracle_camera(), W, H, N, max_depth=depth, workers=14)
    assert not np.array_equal(faithful["accumulator"], want["accumulator"])  # the two indexings really differ on this scene
    r.close()


def test_unfused_generate_kernel_gives_the_same_bits():
    spec = scenes.soup_scene(20000, 3, 0.05, lens=True)
    a = _render(spec, 200, 136, 3, 6)
    b = _render(spec, 200, 136, 3, 6, options=dict(fusedPrimary=0))
    assert np.array_equal(a.accumulator().view(np.uint32), b.accumulator().view(np.uint32))
    assert a.counters() == b.counters()
    a.close(), b.close()


def test_batching_of_sample_passes_does_not_change_the_result():
    spec = scenes.heightfield_scene(64, 9)
    a = _render(spec, 192, 128, 5, 8)
    b = _render(spec, 192, 128, 5, 8, passes_per_call=1)
    c = _render(spec, 192, 128, 5, 8, passes_per_call=2)
    assert np.array_equal(a.accumulator().view(np.uint32), b.accumulator().view(np.uint32))
    assert np.array_equal(a.accumulator().view(np.uint32), c.accumulator().view(np.uint32))
    for x in (a, b, c):
        x.close()


def test_tile_subsets_assemble_to_the_full_frame():
    """Two renderers, each owning every other 64x64 tile (what two GPUs would do), reproduce the single-renderer frame
    bit for bit; pixels of tiles a renderer does not own are left untouched by its read-back."""
    from raytracer_amd import dist as rdist
    spec = scenes.bunny_scene()
    W, H, N, depth = 250, 200, 2, 5
    full = _render(spec, W, H, N, depth)
    want = full.accumulator()
    assignment = rdist.assign_tiles(W, H, 2, 64)
    frame = np.full((H, W, 3), -1.0, np.float32)
    total = {"extendPrimary": 0, "shadow": 0}
    for tiles in assignment:
        r = _render(spec, W, H, N, depth, tiles=tiles)
        out = frame.copy()
        rc = ra.capi.host_lib().rt_read_accumulator(r._h, ra.capi._fp(out))
        assert rc == 0
        w, h, valid = rdist.local_pixel_coords(tiles, W, H, 64)
        owned = np.zeros((H, W), bool)
        owned[h[valid], w[valid]] = True
        assert np.all(out[~owned] == frame[~owned])
        frame[owned] = out[owned]
        for k in total:
            total[k] += r.counters()[k]
        r.close()
    assert np.array_equal(frame.view(np.uint32), want.view(np.uint32))
    assert total["extendPrimary"] == full.counters()["extendPrimary"] == W * H * N
    assert total["shadow"] == full.counters()["shadow"]
    full.close()


def test_probe_trace_matches_oracle_on_random_rays():
    spec = scenes.soup_scene(20000, 3, 0.05)
    r = spec.build_product()
    o = spec.build_oracle()
    rng = np.random.default_rng(7)
    n = 4000
    org = rng.normal(size=(n, 3))
    org = (2.5 * org / np.linalg.norm(org, axis=1, keepdims=True)).astype(np.float32)
    tgt = rng.uniform(-0.8, 0.8, size=(n, 3))
    d = tgt - org
    d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    d[:50, 1] = 0.0  # axis-parallel components
    tri, t, anyh = ra.Probe(r.context).trace(org, d)
    for k in range(n):
        wt, tt = o.closest_hit(org[k], d[k])
        assert tri[k] == wt, k
        if wt != 0xFFFFFFFF:
            assert np.float32(tt) == t[k]
        assert bool(anyh[k]) == o.any_hit(org[k], d[k])
    r.close()


def test_full_size_properties_of_the_benchmark_configuration():
    """BASELINE configs[1] at full size (1920x1080, 16 spp, primary + shadow): too big for the oracle in a test, so check
    size-independent properties: ray counts, run-to-run determinism, agreement of the 1080p accumulator with an
    oracle render of a window of it, and non-negativity."""
    spec = scenes.cube_scene()
    W, H, N = 1920, 1080, 16
    a = _render(spec, W, H, N, 1)
    acc = a.accumulator()
    c = a.counters()
    assert c["extendPrimary"] == W * H * N and c["extendSecondary"] == 0 and c["shadow"] == c["hits"][0] and c["nonFiniteAccepts"] == 0
    b = _render(spec, W, H, N, 1, passes_per_call=4)
    assert np.array_equal(acc.view(np.uint32), b.accumulator().view(np.uint32))
    win = (900, 500, 1028, 564)
    want = spec.build_oracle().render(spec.oracle_camera(), W, H, N, max_depth=1, window=win, workers=14)["accumulator"]
    assert np.array_equal(acc[win[1]:win[3], win[0]:win[2]].view(np.uint32), want[win[1]:win[3], win[0]:win[2]].view(np.uint32))
    assert acc.min() >= 0 and np.isfinite(acc).all()
    img = a.image()
    np.testing.assert_allclose(img, np.power(np.maximum(acc, 0), np.float32(0.45)), rtol=3e-6, atol=1e-7)
    a.close(), b.close()


def test_saved_image_is_the_tonemapped_frame(tmp_path):
    spec = scenes.cube_scene()
    r = _render(spec, 96, 64, 2, 3)
    path = tmp_path / "frame.ppm"
    r.save_image(str(path))
    data = path.read_bytes()
    assert data.startswith(b"P6\n96 64\n255\n")
    px = np.frombuffer(data[len(b"P6\n96 64\n255\n"):], np.uint8).reshape(64, 96, 3)
    want = np.floor(np.clip(r.image(), 0.0, 1.0) * np.float32(255.0) + np.float32(0.5)).astype(np.uint8)
    assert np.array_equal(px, want) and px.max() > 100
    r.close()


def test_frames_queued_without_waiting_give_the_same_bits():
    """rt_enqueue_passes: three different frames queued back to back (no host synchronisation in between); the last one
    must equal a blocking render of the same frame, counters included (the totals live on the device)."""
    spec = scenes.multi_model_scene()
    W, H, N, depth = 160, 128, 3, 1  # depth cap 1: a frame needs no host round trip at all
    ref = _render(spec, W, H, N, depth)
    want_acc, want_cnt = ref.accumulator(), ref.counters()
    ref.close()
    r = spec.build_product(options=dict(maxDepth=depth))
    cam = spec.product_camera()
    other = ra.make_camera(position=(2.0, 1.0, 7.0), target=(0, 0, 0), S_O=6, f=0, A=0)
    for camera in (other, cam, other, cam):
        r.begin_frame(camera, W, H, N)
        r.enqueue_passes(0, 2)
        r.enqueue_passes(2, N - 2)
    got_cnt = r.counters()  # synchronises
    assert np.array_equal(r.accumulator().view(np.uint32), want_acc.view(np.uint32))
    for k in ("extendPrimary", "extendSecondary", "shadow", "hits"):
        assert got_cnt[k] == want_cnt[k]
    r.close()


def test_accumulator_and_image_read_before_any_pass_are_zero():
    spec = scenes.cube_scene()
    r = spec.build_product(options=dict(maxDepth=2))
    r.begin_frame(spec.product_camera(), 100, 70, 4)
    assert not r.accumulator().any() and not r.image().any()
    r.render_passes(0, 1)  # the first pass after that early read must still start from zero
    first = r.accumulator()
    r.begin_frame(spec.product_camera(), 100, 70, 4)
    r.render_passes(0, 1)
    assert np.array_equal(first.view(np.uint32), r.accumulator().view(np.uint32)) and first.any()
    r.close()


@pytest.mark.parametrize("name,N", [("multi", 5), ("bunny", 4)])
def test_pipelined_batches_match_oracle_and_the_unpipelined_path(name, N):
    """Depth cap 1: a call's passes run as two pieces on two slots and consecutive frames overlap; the accumulator must be
    what one batch at a time gives, which must be what the oracle gives."""
    spec = {"multi": scenes.multi_model_scene, "bunny": scenes.bunny_scene}[name]()
    W, H = 200, 152
    want = spec.build_oracle().render(spec.oracle_camera(), W, H, N, max_depth=1, workers=14)
    accs = {}
    for pipelining in (1, 0):
        r = spec.build_product(options=dict(maxDepth=1, pipelining=pipelining))
        for _ in range(3):  # three frames back to back: the second and third overlap their predecessor
            r.begin_frame(spec.product_camera(), W, H, N)
            r.enqueue_passes(0, N)
        accs[pipelining] = r.accumulator()
        cnt = r.counters()
        assert cnt["extendPrimary"] == want["counters"]["extendPrimary"] and cnt["shadow"] == want["counters"]["shadow"]
        assert cnt["hits"][0] == want["counters"]["hits"][0]
        r.close()
    assert np.array_equal(accs[1].view(np.uint32), accs[0].view(np.uint32))
    assert np.array_equal(accs[1].view(np.uint32), want["accumulator"].view(np.uint32))


def test_error_codes_of_the_kernel_abi():
    """Every rthip_* call reports misuse with a code instead of rendering something else: camera outside the region the BVH
    padding was sized for, passes outside the frame, calls before a frame exists, sizes beyond the compiled-in limits."""
    import ctypes as C
    spec = scenes.cube_scene()
    r = spec.build_product(options=dict(maxDepth=2))
    hl, ctx = ra.capi.hip_lib(), r.context
    assert hl.rthip_enqueue_passes(ctx, 0, 1) == ra.capi.RTHIP_E_STATE          # no rthip_begin yet
    acc = np.zeros((8, 8, 3), np.float32)
    assert hl.rthip_read_accumulator(ctx, acc.ctypes.data_as(C.POINTER(C.c_float))) == ra.capi.RTHIP_E_STATE
    far = ra.make_camera(position=(500.0, 0.0, 0.0), target=(0, 0, 0), S_O=6, f=0, A=0)
    frame = ra.capi.FrameDesc(width=64, height=64, numSamples=1, maxDepth=2)
    assert hl.rthip_begin(ctx, C.byref(far), C.byref(frame)) == ra.capi.RTHIP_E_BOUNDS
    assert "origin bound" in ra.capi.error_string(ra.capi.RTHIP_E_BOUNDS)
    # (the renderer reacts to that code by re-flattening for the camera: tests/test_gpu_materials.py, far camera)
    with pytest.raises(ra.RtError) as e:
        r.begin_frame(spec.product_camera(), 70000, 64, 1)                        # tile origins are packed in 16 bits
    assert e.value.code == ra.capi.RTHIP_E_ARG
    r.begin_frame(spec.product_camera(), 64, 64, 2)
    assert hl.rthip_enqueue_passes(ctx, 1, 2) == ra.capi.RTHIP_E_ARG             # passes [1,3) of a 2-sample frame
    assert hl.rthip_enqueue_passes(ctx, 0, 0) == ra.capi.RTHIP_E_ARG
    assert hl.rthip_enqueue_passes(None, 0, 1) == ra.capi.RTHIP_E_ARG
    r.render_passes(0, 2)                                                         # the context still works afterwards
    assert r.accumulator().any()
    with pytest.raises(ra.RtError) as e:
        r.set_option("maxDepth", 1000)
        r.begin_frame(spec.product_camera(), 64, 64, 1)
    assert e.value.code == ra.capi.RTHIP_E_LIMIT
    r.close()


def test_cancel_render_returns_promptly_and_the_renderer_stays_usable():
    """Renderer::startRender with main.cpp's 10 000 spp (src/main.cpp:29): `running` is polled between calls bounded to
    ~50 ms of device work (src/cpu/CPURenderer.cpp:146), so rt_cancel_render returns at once, sampleTimes holds one entry
    per pass that ran (:204-205), and the next render is untouched by the cancelled one."""
    import time
    spec = scenes.soup_scene(20000, 3, 0.05)
    W, H, N = 1920, 1080, 10000
    r = spec.build_product(options=dict(maxDepth=12))
    r.start_render(spec.product_camera(), W, H, N)
    time.sleep(0.5)
    t = time.perf_counter()
    r.cancel()
    waited = time.perf_counter() - t
    ran = len(r.sample_times())
    assert waited < 0.4, f"cancel took {waited:.3f} s"
    assert 0 < ran < N, ran
    assert (r.sample_times() > 0).all()
    # a fresh render after the cancel gives exactly what a renderer that was never cancelled gives
    r.render(spec.product_camera(), 160, 120, 3)
    assert len(r.sample_times()) == 3
    got = r.accumulator()
    ref = _render(spec, 160, 120, 3, 12)
    assert np.array_equal(got.view(np.uint32), ref.accumulator().view(np.uint32))
    r.close(), ref.close()


def test_gather_over_rccl_single_rank_communicator():
    """rthip_comm_init + rthip_gather with a one-rank communicator: the rank sends its tile-major accumulator to itself through
    RCCL (grouped ncclSend / ncclRecv), zero-padded to the common per-rank size; re-tiled with rthip_untile it is the frame."""
    import ctypes as C
    import torch
    from raytracer_amd import dist as rdist
    spec = scenes.bunny_scene()
    W, H, N, depth = 250, 200, 2, 5
    r = _render(spec, W, H, N, depth)
    want = r.accumulator()
    hl = ra.capi.hip_lib()
    ident = C.create_string_buffer(ra.capi.RTHIP_COMM_ID_BYTES)
    assert hl.rthip_comm_unique_id(ident) == 0, ra.capi.error_string(ra.capi.RTHIP_E_COMM)
    rc = hl.rthip_comm_init(r.context, ident.raw, 0, 1)
    assert rc == 0, ra.capi.error_string(rc)
    assert hl.rthip_comm_init(r.context, ident.raw, 0, 1) == ra.capi.RTHIP_E_STATE  # already joined
    tiles = rdist.assign_tiles(W, H, 1, 64)[0]
    mine = len(tiles) * 64 * 64 * 3
    per_rank = mine + 64 * 64 * 3  # as if another rank owned one tile more: this rank's block is zero-padded
    recv = torch.full((per_rank,), -7.0, dtype=torch.float32, device="cuda:0")
    frame = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda:0")
    torch.cuda.synchronize()
    rc = hl.rthip_gather(r.context, recv.data_ptr(), per_rank, 0)
    assert rc == 0, ra.capi.error_string(rc)
    rc = hl.rthip_untile(r.context, recv.data_ptr(), ra.capi._up(tiles), len(tiles), frame.data_ptr())  # same stream: ordered behind the gather
    assert rc == 0, ra.capi.error_string(rc)
    assert hl.rthip_synchronize(r.context) == 0
    got = recv.cpu().numpy()
    assert np.array_equal(got[:mine].reshape(-1, 3).view(np.uint32), rdist.to_tile_major(want, tiles).view(np.uint32))
    assert not got[mine:].any()
    assert np.array_equal(frame.cpu().numpy().view(np.uint32), want.view(np.uint32))
    assert hl.rthip_gather(r.context, None, per_rank, 0) == ra.capi.RTHIP_E_ARG      # the root needs a receive buffer
    assert hl.rthip_gather(r.context, recv.data_ptr(), 3, 0) == ra.capi.RTHIP_E_ARG   # smaller than this rank's accumulator
    assert hl.rthip_comm_destroy(r.context) == 0
    assert hl.rthip_gather(r.context, recv.data_ptr(), per_rank, 0) == ra.capi.RTHIP_E_STATE
    r.close()


@pytest.mark.parametrize("W,H,N", [(96, 64, 1), (640, 360, 4), (300, 200, 70), (1920, 1080, 2)])
def test_refill_modes_render_the_same_bits_and_calibration_stays_inside_the_buffers(W, H, N):
    """The refill thresholds of the traversal launches (rthip_set_refill_mode) and the size of their persistent grid are a
    matter of speed only: low, high, packets and the measured choice give the same accumulator.  The measuring run traces a sample of the frame before the first pass; frame
    shapes whose whole frame is smaller than that sample, or whose batches are cut in two for the two slots, must keep it
    within what rthip_begin allocated (it once did not)."""
    spec = scenes.heightfield_scene(64)
    frames = {}
    for mode in (0, 1, 2, 3, -1):
        r = spec.build_product(options=dict(maxDepth=5, refillMode=mode))
        r.begin_frame(spec.product_camera(), W, H, N)
        got, ms = r.refill_mode()
        if mode >= 0:
            assert got == mode and not ms.any()
        else:
            assert got in (0, 1, 2) and (ms[:3] > 0).all(), "the first frame of a scene measures every setting"
        assert r.grid_blocks() == 3
        r.render_passes(0, N)
        frames[mode] = r.accumulator()
        if mode < 0:
            r.begin_frame(spec.product_camera(), W, H, N)  # same scene, same shape: no second measurement
            assert np.array_equal(r.refill_mode()[1], ms)
            r.render_passes(0, N)
            assert np.array_equal(r.accumulator().view(np.uint32), frames[mode].view(np.uint32))
        r.close()
    assert frames[0].any()
    for mode in (1, 2, 3, -1):
        assert np.array_equal(frames[0].view(np.uint32), frames[mode].view(np.uint32)), mode


@pytest.mark.parametrize("name,W,H,N,depth", [("bunny", 160, 160, 2, 4), ("hf", 192, 128, 3, 5), ("multi", 192, 160, 2, 6), ("box_glb", 200, 150, 3, 8),
                                             ("soup", 160, 160, 2, 5), ("cube", 250, 130, 2, 3)])
def test_packet_traversal_matches_oracle_bitexact(name, W, H, N, depth):
    """The packet kernel (camera rays and the shadow rays of their hits traced one node at a time per wave, rt_packet.h) forced
    on, against the oracle: hit ids, shadow masks, ray counts, accumulator.  Waves here hold 64 DIFFERENT pixels (few samples),
    the worst case for packets and the one where lanes disagree most about which children to enter."""
    spec = {"cube": scenes.cube_scene, "bunny": scenes.bunny_scene, "soup": scenes.soup_scene, "hf": scenes.heightfield_scene,
            "multi": scenes.multi_model_scene, "box_glb": scenes.box_glb_scene}[name]()
    aov = min(depth, 3)
    want = spec.build_oracle().render(spec.oracle_camera(), W, H, N, max_depth=depth, aov_depths=aov, workers=14)
    assert want["counters"]["nonFiniteAccepts"] == 0
    r = spec.build_product(options=dict(maxDepth=depth, aovDepths=aov, refillMode=2))
    r.begin_frame(spec.product_camera(), W, H, N)
    assert r.refill_mode()[0] == 2
    for s in range(N):
        r.render_passes(s, 1)
        for d in range(aov):
            tri, sh = r.aov(s, d)
            assert np.array_equal(tri, want["tri_id"][s, d]), (s, d, int((tri != want["tri_id"][s, d]).sum()))
            assert np.array_equal(sh, want["shadow"][s, d]), (s, d, int((sh != want["shadow"][s, d]).sum()))
    cnt, oc = r.counters(), want["counters"]
    assert cnt["nonFiniteAccepts"] == 0 and cnt["extendPrimary"] == oc["extendPrimary"] and cnt["shadow"] == oc["shadow"]
    assert cnt["hits"][:depth] == oc["hits"][:depth]
    assert np.array_equal(r.accumulator().view(np.uint32), want["accumulator"].view(np.uint32))
    # the same frame without AOVs, all passes in one call (the layout packets are meant for: a pixel's samples side by side)
    r.set_option("aovDepths", 0)
    r.begin_frame(spec.product_camera(), W, H, N)
    r.render_passes(0, N)
    assert np.array_equal(r.accumulator().view(np.uint32), want["accumulator"].view(np.uint32))
    r.close()
