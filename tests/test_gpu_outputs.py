"""The callers and data formats either side of the render path, on the device, through the C ABI:
several GPUs in one process (ptx_render_image_multi), image files written beside the rendering
(ptx_write_image_async), checkpoints of the accumulation buffer (ptx_accum_save / ptx_accum_load),
and the host-side Renderer keeping its scene on the GPUs between render() calls.

Bars: work counters (integers) equal; images to the bar of test_gpu_parity.py (only the order
of the fp32 sums differs); files byte-identical to the synchronous writers' and, for the ASCII
PPM, to the reference's own file (md5 of the golden)."""
import hashlib
import os

import numpy as np
import pytest

from conftest import load_golden
from test_gpu_parity import assert_image_close

pytestmark = pytest.mark.gpu


def test_one_image_on_several_gpus(pt, scenes):
    """ptx_render_image_multi with 1, 2 and 3 workers (every worker its own scene object and host
    thread; on this one-GPU box they share the device, on a node each sits on its own GPU): the
    sample slices add up to the image one worker renders alone, and to the same work."""
    hs = scenes.host_scene("cornell_buster")
    w, h, aa, nb = 240, 135, 4, 5
    one = scenes.device_scene("cornell_buster").render_image(w, h, aa, nb)
    workers = [pt.DeviceScene(hs, 0, 1 << 28) for _ in range(3)]
    for n in (1, 2, 3):
        r = pt.render_image_multi(workers[:n], w, h, aa, nb)
        assert_image_close(r["color"], one["color"], r["rgb"], one["rgb"])
        assert len(r["stats"]) == n
        for k in ("samples", "trace_rays", "shadow_rays", "shaded_hits"):
            assert sum(s[k] for s in r["stats"]) == one["stats"][k], k
        assert all(s["samples"] > 0 for s in r["stats"]) and r["seconds"] > 0
    # a sub-range of the samples, more workers than samples (the surplus workers render nothing), bad arguments
    r = pt.render_image_multi(workers, w, h, aa, nb, 5, 7)
    part = scenes.device_scene("cornell_buster").render_image(w, h, aa, nb, 5, 7)
    assert_image_close(r["color"], part["color"], r["rgb"], part["rgb"])
    assert sorted(s["samples"] for s in r["stats"]) == [0, w * h, w * h]
    with pytest.raises(pt.PtxError):
        pt.render_image_multi([workers[0], workers[0]], w, h, aa, nb)
    for d in workers:
        d.close()


def _rccl():
    """the RCCL library this process already holds (torch brings one), or the system's"""
    import ctypes
    for name in ("librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"):
        try:
            return ctypes.CDLL(name, mode=ctypes.RTLD_GLOBAL)
        except OSError:
            continue
    return None


def test_rccl_reduce_through_the_c_abi(pt):
    """ptx_reduce_accum = ncclReduce(sum, float) through the library's run-time lookup of RCCL, on a communicator of
    one rank (all this one-GPU box can hold: RCCL refuses two ranks on one device): the call resolves the library,
    queues the reduce on the stream given and leaves the buffer as it was; with PTX_REDUCE_RCCL two workers on one
    GPU are refused with a clear error."""
    import ctypes
    import torch
    lib = _rccl()
    if lib is None:
        pytest.skip("no librccl.so on this machine")
    comm = ctypes.c_void_p()
    devs = (ctypes.c_int * 1)(0)
    assert lib.ncclCommInitAll(ctypes.byref(comm), 1, devs) == 0
    try:
        x = torch.arange(100000, dtype=torch.float32, device="cuda:0") * 0.5
        want = x.clone()
        out = torch.zeros_like(x)
        stream = torch.cuda.current_stream().cuda_stream
        pt.reduce_accum(comm.value, x.data_ptr(), out.data_ptr(), x.numel(), 0, stream)
        pt.reduce_accum(comm.value, x.data_ptr(), None, x.numel(), 0, stream)     # in place
        torch.cuda.synchronize()
        assert torch.equal(out, want) and torch.equal(x, want)
        with pytest.raises(pt.PtxError):
            pt.reduce_accum(None, x.data_ptr(), None, x.numel())
    finally:
        lib.ncclCommDestroy(comm)
    hs = pt.HostScene(pt.SDF_DIR + "/cornell.sdf")
    workers = [pt.DeviceScene(hs, 0, 1 << 27) for _ in range(2)]
    with pytest.raises(pt.PtxError, match="GPU of its own"):
        pt.render_image_multi(workers, 64, 48, 2, 3, rccl=True)
    one = pt.render_image_multi(workers[:1], 64, 48, 2, 3, rccl=True)         # a single worker has nothing to sum
    assert np.isfinite(one["color"]).all()
    for d in workers:
        d.close()


def test_two_gpus_when_there_are_two(pt, scenes):
    """On a machine with two GPUs or more: one image over two GPUs of this process, summed by peer reads and by one
    RCCL reduce, and the two-rank bench (torch.distributed over the nccl backend = RCCL) with --verify.  Skipped
    on a one-GPU box, where only the single-device halves of these paths can run."""
    import json
    import os
    import subprocess
    import sys
    if pt.device_count() < 2:
        pytest.skip("one GPU: the multi-GPU legs have no second device to run on")
    hs = scenes.host_scene("cornell_buster")
    w, h, aa, nb = 240, 135, 4, 5
    one = scenes.device_scene("cornell_buster").render_image(w, h, aa, nb)
    workers = [pt.DeviceScene(hs, d, 1 << 30) for d in (0, 1)]
    for rccl in (False, True):
        r = pt.render_image_multi(workers, w, h, aa, nb, rccl=rccl)
        assert_image_close(r["color"], one["color"], r["rgb"], one["rgb"])
        for k in ("samples", "trace_rays", "shadow_rays", "shaded_hits"):
            assert sum(s[k] for s in r["stats"]) == one["stats"][k], k
    for d in workers:
        d.close()
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("PTX_BENCH_BACKEND", None)
    out = subprocess.run([sys.executable, os.path.join(repo, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--aa", "8", "--cpu-seconds", "0",
                          "--no-also", "--no-census", "--verify", "--queue-gib", "4"], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads(out.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["verify_max_abs_vs_single_rank"] <= 1e-5


def test_image_files_written_beside_the_rendering(pt, scenes, tmp_path):
    import torch
    g = load_golden("cornell")
    w, h, aa, nb = (int(v) for v in g["params"])
    ds = scenes.device_scene("cornell")
    spp = aa * aa
    accum = torch.zeros(w * h * 3, dtype=torch.float32, device="cuda:0")
    color = torch.empty(w * h * 3, dtype=torch.float32, device="cuda:0")
    rgb = torch.empty(w * h * 3, dtype=torch.uint8, device="cuda:0")
    stream = torch.cuda.current_stream().cuda_stream
    half = spp // 2
    # first half of the samples -> preview files queued -> second half rendered while they are written
    ds.render_accumulate(pt.render_params(w, h, aa, nb, 0, half), accum.data_ptr(), stream)
    ds.resolve(accum.data_ptr(), w, h, half, color.data_ptr(), rgb.data_ptr(), stream)
    torch.cuda.synchronize()
    preview_rgb, preview_color = rgb.cpu().numpy().reshape(h, w, 3).copy(), color.cpu().numpy().reshape(h, w, 3).copy()
    jobs = [pt.ImageWrite(tmp_path / "p.ppm", "ppm", 0, rgb.data_ptr(), w, h, stream),
            pt.ImageWrite(tmp_path / "p6.ppm", "ppm_binary", 0, rgb.data_ptr(), w, h, stream),
            pt.ImageWrite(tmp_path / "p.pfm", "pfm", 0, color.data_ptr(), w, h, stream)]
    ds.render_accumulate(pt.render_params(w, h, aa, nb, half, spp), accum.data_ptr(), stream)
    ds.resolve(accum.data_ptr(), w, h, spp, color.data_ptr(), rgb.data_ptr(), stream)   # overwrites what the jobs copy from: queued behind their copies
    final = pt.ImageWrite(tmp_path / "final.ppm", "ppm", 0, rgb.data_ptr(), w, h, stream)
    for j in jobs + [final]:
        j.wait()
    # the previews are what the synchronous writers give for the preview image
    pt.write_ppm(tmp_path / "s.ppm", preview_rgb)
    pt.write_ppm_binary(tmp_path / "s6.ppm", preview_rgb)
    pt.write_pfm(tmp_path / "s.pfm", preview_color)
    for a, b in (("p.ppm", "s.ppm"), ("p6.ppm", "s6.ppm"), ("p.pfm", "s.pfm")):
        assert (tmp_path / a).read_bytes() == (tmp_path / b).read_bytes(), a
    # the final file is the reference's file: same values (to the image bar), and when no byte differs, the same md5
    text = (tmp_path / "final.ppm").read_bytes()
    assert text.startswith(b"P3 %d %d 255 \n" % (w, h))
    vals = np.array(text.split()[4:], dtype=np.int32).astype(np.uint8).reshape(h, w, 3)
    bd = np.abs(vals.astype(int) - g["rgb"].astype(int))
    assert bd.max() <= 1 and (bd > 0).mean() <= 1e-3
    if not bd.any():
        assert hashlib.md5(text).hexdigest() == str(g["ppm_md5"])
    # errors: a directory that does not exist fails in wait(), not silently
    j = pt.ImageWrite(tmp_path / "no_such_dir" / "x.ppm", "ppm", 0, rgb.data_ptr(), w, h, stream)
    with pytest.raises(pt.PtxError):
        j.wait()


def test_checkpoint_and_resume(pt, scenes, tmp_path):
    """A render interrupted after some samples and resumed from the file gives the image an
    uninterrupted render gives; the file round-trips the buffer bit for bit; files for another
    image or damaged files are refused."""
    import torch
    ds = scenes.device_scene("example_fixed")
    w, h, aa, nb = 160, 90, 4, 5
    spp, k = aa * aa, 6
    full = ds.render_image(w, h, aa, nb)
    accum = torch.zeros(w * h * 3, dtype=torch.float32, device="cuda:0")
    ds.render_accumulate(pt.render_params(w, h, aa, nb, 0, k), accum.data_ptr())
    ck = tmp_path / "render.acc"
    pt.accum_save(ck, 0, accum.data_ptr(), w, h, aa, 0, k)
    assert os.path.getsize(ck) == 32 + w * h * 3 * 4 and not os.path.exists(str(ck) + ".part")
    resumed = torch.full((w * h * 3,), 7.0, dtype=torch.float32, device="cuda:0")
    assert pt.accum_load(ck, 0, resumed.data_ptr(), w, h, aa) == (0, k)
    assert torch.equal(resumed, accum)
    ds.render_accumulate(pt.render_params(w, h, aa, nb, k, spp), resumed.data_ptr())
    color = torch.empty(w * h * 3, dtype=torch.float32, device="cuda:0")
    rgb = torch.empty(w * h * 3, dtype=torch.uint8, device="cuda:0")
    ds.resolve(resumed.data_ptr(), w, h, spp, color.data_ptr(), rgb.data_ptr())
    torch.cuda.synchronize()
    assert_image_close(color.cpu().numpy().reshape(h, w, 3), full["color"], rgb.cpu().numpy().reshape(h, w, 3), full["rgb"])
    with pytest.raises(pt.PtxError):       # another image size
        pt.accum_load(ck, 0, resumed.data_ptr(), w, h + 1, aa)
    with pytest.raises(pt.PtxError):       # another sample grid
        pt.accum_load(ck, 0, resumed.data_ptr(), w, h, aa + 1)
    data = ck.read_bytes()
    (tmp_path / "short.acc").write_bytes(data[:-5])
    (tmp_path / "junk.acc").write_bytes(b"not a checkpoint" * 10)
    for bad in ("short.acc", "junk.acc", "missing.acc"):
        with pytest.raises(pt.PtxError):
            pt.accum_load(tmp_path / bad, 0, resumed.data_ptr(), w, h, aa)


def test_renderer_keeps_its_scene(pt, scenes):
    """Renderer::render twice on the same Scene with keep_scene: the second call skips flattening,
    BVH build and upload (set-up time exactly 0) and renders the same image; without it every call
    sets the scene up again, like the reference copies it per thread on every call."""
    hs = scenes.host_scene("cornell_buster")
    g = load_golden("cornell_buster")
    w, h, aa, nb = (int(v) for v in g["params"])
    r = pt.HostRenderer(w, h, aa, nb, keep_scene=True)
    a, b = r.render(hs), r.render(hs)
    assert a["setup_seconds"] > 0 and b["setup_seconds"] == 0
    for x in (a, b):
        assert float(np.abs(x["color"] - g["color"]).max()) <= 1e-5
    other = scenes.host_scene("cornell")        # another Scene object: set up anew
    c = r.render(other)
    assert c["setup_seconds"] > 0
    r.close()
    r = pt.HostRenderer(w, h, aa, nb, first_device=0, n_devices=1)
    a, b = r.render(hs), r.render(hs)
    assert a["setup_seconds"] > 0 and b["setup_seconds"] > 0
    assert float(np.abs(b["color"] - g["color"]).max()) <= 1e-5
    r.close()
