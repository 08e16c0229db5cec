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
