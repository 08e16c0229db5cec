"""ptx_render_accumulate_static (work memory in a static device array, nothing allocated)
against ptx_render_accumulate, through the C ABI, on two calls in a row, the second smaller.

Bar: the work counters (rays, shadow rays, shaded hits per depth) are integers and must be
equal; the accumulation buffers are fp32 sums whose order differs from run to run (float
atomics), so they are compared like images: after the resolve (average, tone map)
max abs <= 1e-5 and RMSE <= 1e-6, the 8-bit rows differ in at most 0.1 % of the bytes by at most 1.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

COUNT_KEYS = ("samples", "trace_rays", "shadow_rays", "shaded_hits", "per_depth_rays", "per_depth_shadow", "per_depth_hits")


def _buffers(torch, n):
    """accumulation buffer, colour buffer, 8-bit rows for images of up to n pixels (allocated once, up front)"""
    return (torch.zeros(n * 3, dtype=torch.float32, device="cuda:0"), torch.empty(n * 3, dtype=torch.float32, device="cuda:0"),
            torch.empty(n * 3, dtype=torch.uint8, device="cuda:0"))


def _render(pt, ds, torch, bufs, w, h, aa, nb, static, s0=0, s1=None):
    rp = pt.render_params(w, h, aa, nb, s0, s1)
    accum, color, rgb = (b[: w * h * 3] for b in bufs)
    accum.zero_()
    st = ds.render_accumulate(rp, accum.data_ptr(), None, check_overflow=True, want_stats=True, static=static)
    ds.resolve(accum.data_ptr(), w, h, rp.sample_end - rp.sample_begin, color.data_ptr(), rgb.data_ptr())
    torch.cuda.synchronize()
    return st, color.cpu().numpy().copy(), rgb.cpu().numpy().copy()


def _same(a, b):
    sa, ca, ra = a
    sb, cb, rb = b
    for k in COUNT_KEYS:
        assert sa[k] == sb[k], k
    assert sa["trace_rays"] > 0 and sa["shaded_hits"] > 0
    diff = ca.astype(np.float64) - cb.astype(np.float64)
    assert np.abs(diff).max() <= 1e-5
    assert np.sqrt(np.mean(diff ** 2)) <= 1e-6
    bd = np.abs(ra.astype(int) - rb.astype(int))
    assert bd.max() <= 1 and (bd > 0).mean() <= 1e-3


@pytest.mark.parametrize("name", ["cornell_buster", "example_fixed"])
def test_static_work_memory_gives_the_same_render(pt, scenes, name):
    import torch
    ds = scenes.device_scene(name)
    bufs = _buffers(torch, 320 * 200)
    _render(pt, ds, torch, bufs, 32, 32, 1, 1, True)     # the runtime's own first-launch set-up happens here
    free0 = torch.cuda.mem_get_info()[0]
    # first call: 320x200, 16 spp, 5 bounces; second, smaller: 96x64, samples 1..3 of a 2x2 grid, 3 bounces
    big = _render(pt, ds, torch, bufs, 320, 200, 4, 5, True)
    small = _render(pt, ds, torch, bufs, 96, 64, 2, 3, True, 1, 4)
    assert torch.cuda.mem_get_info()[0] == free0          # nothing was allocated (and nothing freed)
    _same(big, _render(pt, ds, torch, bufs, 320, 200, 4, 5, False))
    _same(small, _render(pt, ds, torch, bufs, 96, 64, 2, 3, False, 1, 4))


def test_static_work_memory_limits(pt, scenes):
    import torch
    ds = scenes.device_scene("cornell")
    accum = torch.zeros(4096 * 4096 * 3, dtype=torch.float32, device="cuda:0")
    with pytest.raises(pt.PtxError) as e:   # 4096 x 4096 x 4 samples > 2^24
        ds.render_accumulate(pt.render_params(4096, 4096, 2, 5), accum.data_ptr(), static=True)
    assert e.value.status == 1 and "static work memory" in str(e.value)
    with pytest.raises(pt.PtxError) as e:
        ds.render_accumulate(pt.render_params(64, 64, 1, 17), accum.data_ptr(), static=True)
    assert e.value.status == 1 and "max_ray_bounces" in str(e.value)
    # exactly at the limit it runs: 4096 x 4096 x 1
    st = ds.render_accumulate(pt.render_params(4096, 4096, 1, 5), accum.data_ptr(), check_overflow=True, want_stats=True, static=True)
    assert st["samples"] == 1 << 24
