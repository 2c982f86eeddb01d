"""rthip_trace_rays_static, the allocation-free variant of rthip_probe_trace: same answers, bit for bit, on two calls in a
row (the second smaller, so stale contents of the shared static scratch would show), and a clear error beyond its limit.
Inputs and outputs are device memory (torch tensors: plumbing only); everything goes through the C ABI."""
import ctypes as C

import numpy as np
import pytest

import raytracer_amd as ra
from raytracer_amd import scenes

pytestmark = pytest.mark.gpu


def _rays(n, seed):
    rng = np.random.default_rng(seed)
    org = rng.normal(size=(n, 3))
    org = (2.5 * org / np.linalg.norm(org, axis=1, keepdims=True)).astype(np.float32)
    tgt = rng.uniform(-0.8, 0.8, size=(n, 3))
    d = tgt - org
    d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    d[: min(n, 40), 1] = 0.0  # axis-parallel components
    return org, d


def _static_trace(ctx, org, d, tmin=1e-4, tmax=1e20):
    import torch
    n = org.shape[0]
    dev = torch.device("cuda", 0)
    o_t, d_t = torch.from_numpy(org).to(dev).contiguous(), torch.from_numpy(d).to(dev).contiguous()
    tri_t = torch.full((n,), 0x7FFFFFFF, dtype=torch.int32, device=dev)
    t_t = torch.full((n,), -1.0, dtype=torch.float32, device=dev)
    any_t = torch.full((n,), 77, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    free_before, _ = torch.cuda.mem_get_info(dev)  # hipMemGetInfo: the call must leave the device's free memory as it found it
    rc = ra.capi.hip_lib().rthip_trace_rays_static(ctx, o_t.data_ptr(), d_t.data_ptr(), n, float(np.float32(tmin)), float(np.float32(tmax)),
                                                   tri_t.data_ptr(), t_t.data_ptr(), any_t.data_ptr())
    free_after, _ = torch.cuda.mem_get_info(dev)
    assert free_after == free_before, "rthip_trace_rays_static changed the device's free memory: %d -> %d" % (free_before, free_after)
    return rc, tri_t.cpu().numpy().view(np.uint32), t_t.cpu().numpy(), any_t.cpu().numpy()


@pytest.mark.parametrize("scene", ["soup", "hf_deep"])
def test_static_entry_point_matches_probe_trace_twice_in_a_row(scene):
    # hf_deep: a BVH deep enough that the traversal stack spills past its LDS part into the static scratch
    spec = scenes.soup_scene(20000, 3, 0.05) if scene == "soup" else scenes.heightfield_scene(300, 5)
    r = spec.build_product()
    probe = ra.Probe(r.context)
    for n, seed in ((ra.capi.RTHIP_STATIC_MAX_RAYS, 11), (3001, 12)):  # the second call is smaller than the first
        org, d = _rays(n, seed)
        if scene == "hf_deep":
            org[:, 1] = np.abs(org[:, 1]) + 0.5
        want_tri, want_t, want_any = probe.trace(org, d)
        rc, tri, t, anyh = _static_trace(r.context, org, d)
        assert rc == 0, ra.capi.error_string(rc)
        assert np.array_equal(tri, want_tri)
        assert np.array_equal(t.view(np.uint32), want_t.view(np.uint32))
        assert np.array_equal(anyh, want_any)
        assert (want_tri != 0xFFFFFFFF).sum() > n // 20 and want_any.sum() > n // 20  # the comparison is not vacuous
    r.close()


def test_static_entry_point_rejects_sizes_beyond_its_limit_and_leaves_frames_alone():
    spec = scenes.cube_scene()
    r = spec.build_product(options=dict(maxDepth=3))
    r.begin_frame(spec.product_camera(), 96, 64, 2)
    r.render_passes(0, 1)
    org, d = _rays(ra.capi.RTHIP_STATIC_MAX_RAYS + 1, 5)
    rc, *_ = _static_trace(r.context, org, d)
    assert rc == ra.capi.RTHIP_E_LIMIT and "limit" in ra.capi.error_string(rc)
    hl = ra.capi.hip_lib()
    assert hl.rthip_trace_rays_static(r.context, None, None, 4, 0.0, 1.0, None, None, None) == ra.capi.RTHIP_E_ARG
    # a call in the middle of a frame does not disturb it (the context's own buffers are untouched)
    org, d = _rays(500, 6)
    launches_before = r.kernel_times()
    rc, tri, t, anyh = _static_trace(r.context, org, d)
    assert rc == 0
    assert {k: v[1] for k, v in r.kernel_times().items()} == {k: v[1] for k, v in launches_before.items()}  # launch tallies of the frame untouched
    r.render_passes(1, 1)
    got = r.accumulator()
    ref = spec.build_product(options=dict(maxDepth=3))
    ref.begin_frame(spec.product_camera(), 96, 64, 2)
    ref.render_passes(0, 2)
    assert np.array_equal(got.view(np.uint32), ref.accumulator().view(np.uint32))
    want_tri, want_t, want_any = ra.Probe(ref.context).trace(org, d)
    assert np.array_equal(tri, want_tri) and np.array_equal(anyh, want_any)
    r.close(), ref.close()
