"""Dynamic tile ownership through the C ABI (include/rthip.h: rthip_begin_dynamic, rthip_enqueue_tiles, rthip_wait_tiles,
rthip_gather_tiles): tiles handed to a context in uneven batches and in any order while the frame renders, gathered over a
one-rank RCCL communicator with the variable-count protocol and re-tiled, give the frame a static single-GPU render gives,
bit for bit -- a whole tile (all its samples) is rendered by one rank, so per-pixel accumulation order never changes
(src/cpu/CPURenderer.cpp:197)."""
import ctypes as C

import numpy as np
import pytest

import raytracer_amd as ra
from raytracer_amd import dist as rdist
from tests import scenes

pytestmark = pytest.mark.gpu


def _static_frame(spec, W, H, N, depth):
    r = spec.build_product(options=dict(maxDepth=depth))
    r.begin_frame(spec.product_camera(), W, H, N)
    r.render_passes(0, N)
    acc, cnt = r.accumulator(), r.counters()
    r.close()
    return acc, cnt


@pytest.mark.parametrize("name,W,H,N,depth", [("hf", 500, 330, 5, 6), ("bunny", 250, 200, 2, 4), ("soup", 640, 360, 1, 4)])
def test_dynamic_tiles_gathered_with_variable_counts_equal_the_static_frame(name, W, H, N, depth):
    import torch
    spec = {"hf": lambda: scenes.heightfield_scene(96, 9), "bunny": scenes.bunny_scene, "soup": lambda: scenes.soup_scene(20000, 3, 0.05)}[name]()
    want, want_cnt = _static_frame(spec, W, H, N, depth)
    tx, ty = rdist.tile_grid(W, H, 64)
    total = tx * ty
    rng = np.random.default_rng(3)
    order = rng.permutation(total).astype(np.uint32)     # any order ...
    cuts = [0, 1, 4, 5, min(total, 11), total]           # ... in uneven batches
    r = spec.build_product(options=dict(maxDepth=depth))
    hl = ra.capi.hip_lib()
    ident = C.create_string_buffer(ra.capi.RTHIP_COMM_ID_BYTES)
    assert hl.rthip_comm_unique_id(ident) == 0
    assert hl.rthip_comm_init(r.context, ident.raw, 0, 1) == 0
    frame = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda:0")
    recv = torch.full((total * 64 * 64 * 3,), -3.0, dtype=torch.float32, device="cuda:0")
    ids_out = np.zeros(total, np.uint32)
    counts_out = np.zeros(1, np.uint32)
    for rep in range(2):  # two frames back to back: the second reuses tables, events and buffers
        r.begin_dynamic(spec.product_camera(), W, H, N, depth)
        for a, b in zip(cuts[:-1], cuts[1:]):
            if b > a:
                r.enqueue_tiles(order[a:b], max_in_flight=2)
        assert np.array_equal(r.local_tiles(), order)
        torch.cuda.synchronize()
        rc = hl.rthip_gather_tiles(r.context, recv.data_ptr(), recv.numel(), ra.capi._up(ids_out), total, ra.capi._up(counts_out), 0)
        assert rc == 0, ra.capi.error_string(rc)
        assert counts_out[0] == total and np.array_equal(ids_out, order)
        rc = hl.rthip_untile(r.context, recv.data_ptr(), ra.capi._up(ids_out), total, frame.data_ptr())
        assert rc == 0, ra.capi.error_string(rc)
        assert hl.rthip_synchronize(r.context) == 0
        assert np.array_equal(frame.cpu().numpy().view(np.uint32), want.view(np.uint32)), "frame %d" % rep
        cnt = r.counters()
        assert all(cnt[k] == want_cnt[k] for k in ("extendPrimary", "extendSecondary", "shadow", "hits"))
        assert np.array_equal(r.accumulator().view(np.uint32), want.view(np.uint32))  # the context's own read-back knows the arrival order too
        order = order[::-1].copy()
        frame.zero_()
    # a partial frame: only the tiles that came are there, the rest of the image is left alone
    r.begin_dynamic(spec.product_camera(), W, H, N, depth)
    r.enqueue_tiles(order[:3])
    got = r.accumulator()
    w, h, valid = rdist.local_pixel_coords(order[:3], W, H, 64)
    mask = np.zeros((H, W), bool)
    mask[h[valid], w[valid]] = True
    assert np.array_equal(got[mask].view(np.uint32), want[mask].view(np.uint32)) and not got[~mask].any()
    # errors: more tiles than the frame was begun for, a tile outside the image, a static frame
    assert hl.rthip_enqueue_tiles(r.context, ra.capi._up(np.arange(total, dtype=np.uint32)), total) == ra.capi.RTHIP_E_LIMIT
    assert hl.rthip_enqueue_tiles(r.context, ra.capi._up(np.array([total], np.uint32)), 1) == ra.capi.RTHIP_E_ARG
    r.begin_frame(spec.product_camera(), W, H, N)
    assert hl.rthip_enqueue_tiles(r.context, ra.capi._up(order[:1]), 1) == ra.capi.RTHIP_E_STATE
    # ... and a static frame through the variable-count gather (no padding travels)
    r.set_tiles(order[: total // 2])
    r.begin_frame(spec.product_camera(), W, H, N)
    r.render_passes(0, N)
    rc = hl.rthip_gather_tiles(r.context, recv.data_ptr(), recv.numel(), ra.capi._up(ids_out), total, ra.capi._up(counts_out), 0)
    assert rc == 0 and counts_out[0] == total // 2 and np.array_equal(ids_out[: total // 2], order[: total // 2])
    frame.zero_()
    assert hl.rthip_untile(r.context, recv.data_ptr(), ra.capi._up(ids_out), total // 2, frame.data_ptr()) == 0
    w, h, valid = rdist.local_pixel_coords(order[: total // 2], W, H, 64)
    got = frame.cpu().numpy()
    assert np.array_equal(got[h[valid], w[valid]].view(np.uint32), want[h[valid], w[valid]].view(np.uint32))
    assert hl.rthip_comm_destroy(r.context) == 0
    r.close()
