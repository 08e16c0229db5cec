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


def test_static_call_under_graph_capture(pt, tmp_path):
    """A scene without two-child materials (mirrors and diffuse surfaces: every hit hands on at most one ray) rendered
    through the static entry point with check_overflow = 0 and no statistics is memsets and kernel launches only: the
    call is issued on a stream under capture, the captured graph is launched twice, and each launch adds the image
    the ordinary entry point gives."""
    import ctypes
    import torch
    sdf = tmp_path / "mirrors.sdf"
    sdf.write_text("define material mirror 0.1 0.1 0.1 0.2 0.2 0.2 0.9 0.9 0.9 20 1 1 1\n"
                   "define material wall 0.3 0.3 0.3 0.6 0.5 0.4 0.2 0.2 0.2 8 0 1 1\n"
                   "define shape box floor -6 -1 -14 6 0 2 wall\ndefine shape box back -6 0 -14 6 8 -13 wall\n"
                   "define shape sphere ball -1.5 1.5 -8 1.5 mirror\ndefine shape box block 1 0 -9 3 2.5 -7 mirror\n"
                   "define light bulb 0 7 -4 .8 .8 .8 20\ndefine ambient amb 1 1 1 0.5\n"
                   "define camera eye 60.0 0 2.5 0 0 0 0\n")
    ds = pt.DeviceScene(pt.HostScene(str(sdf)), 0, 1 << 28)
    w, h, aa, nb = 160, 100, 2, 4
    rp = pt.render_params(w, h, aa, nb, pipeline="staged")
    want = torch.zeros(w * h * 3, dtype=torch.float32, device="cuda:0")
    st = ds.render_accumulate(rp, want.data_ptr(), None, check_overflow=True, want_stats=True)
    assert st["per_depth_rays"][1] > 0 and st["shadow_rays"] > 0          # rays bounce and lights are asked for
    ds.render_accumulate(rp, torch.zeros_like(want).data_ptr(), None, static=True)   # first use outside the capture
    torch.cuda.synchronize()
    hip = ctypes.CDLL("libamdhip64.so")
    got = torch.zeros_like(want)
    s = torch.cuda.Stream()
    graph, gexec = ctypes.c_void_p(), ctypes.c_void_p()
    with torch.cuda.stream(s):
        assert hip.hipStreamBeginCapture(ctypes.c_void_p(s.cuda_stream), 0) == 0       # hipStreamCaptureModeGlobal
        try:
            ds.render_accumulate(rp, got.data_ptr(), s.cuda_stream, check_overflow=False, want_stats=False, static=True)
        finally:
            rc = hip.hipStreamEndCapture(ctypes.c_void_p(s.cuda_stream), ctypes.byref(graph))
        assert rc == 0 and graph.value
        assert hip.hipGraphInstantiate(ctypes.byref(gexec), graph, None, None, 0) == 0
        torch.cuda.synchronize()
        assert float(got.abs().max()) == 0.0                                           # captured, not run
        for _ in range(2):
            assert hip.hipGraphLaunch(gexec, ctypes.c_void_p(s.cuda_stream)) == 0
        s.synchronize()
    hip.hipGraphExecDestroy(gexec)
    hip.hipGraphDestroy(graph)
    mean = (got / 2).cpu().numpy()
    ref = want.cpu().numpy()
    assert np.isfinite(mean).all() and ref.max() > 0
    assert np.abs(mean - ref).max() <= 1e-4 * max(1.0, float(np.abs(ref).max()))       # float atomics: the order of the sums differs
    ds.close()
