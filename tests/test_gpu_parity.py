"""Parity of the gfx950 path with the reference, through the C ABI of include/ptx.h.

Bars (stated here, asserted below):
  * hit identity  - the prim a ray hits names the same Shape as the reference's
                    HitPoint::hit_object, for every ray (exact);
  * hit distance  - the requirement is |t - t_ref| <= 1e-5 * |t_ref|; what is asserted is
                    stricter: t, position, normal and ray_direction are BIT-identical;
  * images        - same sample grid as the reference, so per-pixel RMSE of the tone-mapped
                    float image <= 1e-6, max abs error <= 1e-5, and the 8-bit PPM values
                    differ in at most 0.1 % of the bytes, never by more than 1
                    (only the order of the fp32 sums differs from the CPU loop).
"""
import hashlib

import numpy as np
import pytest

from conftest import SCENES, beq, load_golden

pytestmark = pytest.mark.gpu

RMSE_MAX = 1e-6
ABS_MAX = 1e-5


def image_errors(color, ref_color, rgb, ref_rgb):
    diff = color.astype(np.float64) - ref_color.astype(np.float64)
    both_nan = np.isnan(color) & np.isnan(ref_color)
    diff[both_nan] = 0
    rmse = float(np.sqrt(np.mean(diff ** 2)))
    bd = np.abs(rgb.astype(int) - ref_rgb.astype(int))
    return rmse, float(np.max(np.abs(diff))), float((bd > 0).mean()), int(bd.max())


def assert_image_close(color, ref_color, rgb, ref_rgb):
    rmse, mx, frac, worst = image_errors(color, ref_color, rgb, ref_rgb)
    assert rmse <= RMSE_MAX, rmse
    assert mx <= ABS_MAX, mx
    assert frac <= 1e-3 and worst <= 1, (frac, worst)


@pytest.mark.parametrize("name", SCENES)
def test_hits_equal_reference_golden(pt, scenes, name):
    g = load_golden(name)
    ds = scenes.device_scene(name)
    names = scenes.host_scene(name).prim_names()
    table = [str(x) for x in g["names"]]
    hw, hh = int(g["hit_w"]), int(g["hit_h"])
    prim, t = ds.primary_hits(hw, hh, 1, 0)
    got = [names[i] if i >= 0 else "null" for i in prim]
    assert got == [table[i] for i in g["primary_name"]]
    hit = prim >= 0
    assert beq(t[hit], g["primary_t"][hit])
    a = ds.intersect(g["rays_o"], g["rays_d"])
    got = [names[i] if i >= 0 else "null" for i in a["prim"]]
    assert got == [table[i] for i in g["rays_name"]]
    m = a["hit"]
    assert m.sum() > 100
    for k in ("t", "pos", "normal", "raydir"):
        assert beq(a[k][m], g["rays_" + k][m]), k


@pytest.mark.parametrize("name", SCENES)
def test_hits_equal_oracle_many_rays(pt, port, scenes, name):
    """More rays than the fixtures hold: 480x320 primary rays of two aa samples + 60 000 incoherent rays."""
    from oracle_lib import random_rays
    ds, s = scenes.device_scene(name), scenes.oracle_scene(name)
    o, d = port.primary_rays(s, 480, 320, 1)
    o2, d2 = random_rays(60000, 20241111)
    o = np.concatenate([o, o2])
    d = np.concatenate([d, d2])
    a, b = ds.intersect(o, d), port.intersect(s, o, d)
    assert np.array_equal(a["prim"], b["rank"])
    m = b["hit"]
    rel = np.abs(a["t"][m] - b["t"][m]) / np.abs(b["t"][m])
    assert rel.max() <= 1e-5
    for k in ("t", "pos", "normal", "raydir"):
        assert beq(a[k][m], b[k][m]), k
    # the aa grid: sample 5 of a 3x3 grid
    prim, t = ds.primary_hits(200, 120, 3, 5)
    o3, d3 = port.primary_rays(s, 200, 120, 3)
    b = port.intersect(s, o3[5::9], d3[5::9])
    assert np.array_equal(prim, b["rank"]) and beq(t[b["hit"]], b["t"][b["hit"]])


PIPELINES = ["fused", "staged"]


@pytest.mark.parametrize("pipeline", PIPELINES)
@pytest.mark.parametrize("name", SCENES)
def test_image_equals_reference_golden(pt, scenes, name, pipeline):
    g = load_golden(name)
    w, h, aa, nb = (int(v) for v in g["params"])
    r = scenes.device_scene(name).render_image(w, h, aa, nb, pipeline=pipeline)
    assert_image_close(r["color"], g["color"], r["rgb"], g["rgb"])
    assert r["stats"]["samples"] == w * h * aa * aa


@pytest.mark.parametrize("pipeline", PIPELINES)
def test_config1_image(pt, scenes, pipeline):
    """BASELINE config 1 (cornell.sdf 480x320, 16 spp, 5 bounces) against the reference's PPM."""
    g = load_golden("config1_cornell_480x320_aa4")
    w, h, aa, nb = (int(v) for v in g["params"])
    r = scenes.device_scene("cornell").render_image(w, h, aa, nb, pipeline=pipeline)
    bd = np.abs(r["rgb"].astype(int) - g["rgb"].astype(int))
    assert (bd > 0).mean() <= 1e-3 and bd.max() <= 1


@pytest.mark.parametrize("name,w,h,aa,nb", [("cornell", 320, 200, 3, 5), ("cornell", 64, 48, 2, 0), ("cornell", 64, 48, 2, 1),
                                            ("example_fixed", 320, 180, 2, 3), ("cornell_buster", 200, 120, 2, 5),
                                            ("superhot_colored", 256, 144, 2, 4), ("example", 1, 1, 1, 5), ("example", 7, 3, 5, 2)])
def test_image_equals_oracle(pt, port, scenes, name, w, h, aa, nb):
    ref = port.render(scenes.oracle_scene(name), w, h, aa, nb)
    for pipeline in PIPELINES:
        r = scenes.device_scene(name).render_image(w, h, aa, nb, pipeline=pipeline)
        assert_image_close(r["color"], ref["color"], r["rgb"], ref["rgb"])


def test_drop_in_renderer_writes_the_reference_ppm(pt, scenes, tmp_path):
    """Renderer{w,h,file,aa,bounces}.render(scene) through the host-side C++ class."""
    g = load_golden("cornell")
    w, h, aa, nb = (int(v) for v in g["params"])
    out = tmp_path / "img.ppm"
    color, sec = scenes.host_scene("cornell").render_file(w, h, aa, nb, str(out))
    text = out.read_bytes()
    vals = np.array(text.split()[4:], dtype=np.int32).astype(np.uint8).reshape(h, w, 3)
    assert text.startswith(b"P3 %d %d 255 \n" % (w, h))
    assert_image_close(color, g["color"], vals, g["rgb"])
    if np.array_equal(vals, g["rgb"]):
        assert hashlib.md5(text).hexdigest() == str(g["ppm_md5"])


def test_sample_ranges_add_up(pt, scenes):
    """Multi-GPU split: accumulating [0,k) and [k,n) gives the full image (fp32 sums reordered)."""
    import torch
    ds = scenes.device_scene("example_fixed")
    w, h, aa, nb = 160, 90, 4, 5
    full = ds.render_image(w, h, aa, nb)
    acc = torch.zeros(h * w * 3, dtype=torch.float32, device="cuda:0")
    stream = torch.cuda.current_stream().cuda_stream
    for (b, e), pipeline in zip(((0, 5), (5, 6), (6, 16)), ("staged", "fused", "staged")):
        ds.render_accumulate(pt.render_params(w, h, aa, nb, b, e, pipeline), acc.data_ptr(), stream, check_overflow=True)
    color = torch.empty(h * w * 3, dtype=torch.float32, device="cuda:0")
    rgb = torch.empty(h * w * 3, dtype=torch.uint8, device="cuda:0")
    ds.resolve(acc.data_ptr(), w, h, aa * aa, color.data_ptr(), rgb.data_ptr(), stream)
    torch.cuda.synchronize()
    assert_image_close(color.cpu().numpy().reshape(h, w, 3), full["color"], rgb.cpu().numpy().reshape(h, w, 3), full["rgb"])
    # an empty range adds nothing
    before = acc.clone()
    ds.render_accumulate(pt.render_params(w, h, aa, nb, 7, 7), acc.data_ptr(), stream, check_overflow=True)
    torch.cuda.synchronize()
    assert torch.equal(before, acc)


def test_work_counters(pt, port, scenes):
    """The kernels count their rays; primary rays equal samples and hits never exceed the
    oracle's count (rays whose weight is exactly zero are not traced on the GPU)."""
    w, h, aa, nb = 120, 80, 2, 5
    r = scenes.device_scene("cornell").render_image(w, h, aa, nb)
    ref = port.render(scenes.oracle_scene("cornell"), w, h, aa, nb, counters=True)
    st = r["stats"]
    assert st["per_depth_rays"][0] == w * h * aa * aa
    assert st["trace_rays"] == sum(st["per_depth_rays"])
    assert st["trace_rays"] <= int(ref["counters"][0]) and st["shadow_rays"] <= int(ref["counters"][1])
    assert st["shaded_hits"] <= st["trace_rays"]


def test_every_child_ray_on_request(pt, port, scenes):
    """PTX_TRACE_ZERO_WEIGHT: child rays of exactly zero weight are traced as well (example.sdf has
    them: light that passed the blue-only glass meets the mirror without blue in its ks).  Then the
    closest-hit queries number exactly what the reference's Renderer::trace is called, in both
    pipelines; without the flag there are fewer, and the image is the same either way."""
    w, h, aa, nb = 120, 68, 2, 5
    ds = scenes.device_scene("example_fixed")
    ref = port.render(scenes.oracle_scene("example_fixed"), w, h, aa, nb, counters=True)
    for pipeline in PIPELINES:
        lean = ds.render_image(w, h, aa, nb, pipeline=pipeline)
        full = ds.render_image(w, h, aa, nb, pipeline=pipeline, zero_weight=True)
        assert full["stats"]["trace_rays"] == int(ref["counters"][0])
        assert lean["stats"]["trace_rays"] < full["stats"]["trace_rays"]
        assert_image_close(full["color"], ref["color"], full["rgb"], ref["rgb"])
        assert_image_close(lean["color"], full["color"], lean["rgb"], full["rgb"])


def test_edge_cases(pt, tmp_path):
    # no shapes at all: every ray misses, image is black
    sdf = tmp_path / "empty.sdf"
    sdf.write_text("define camera eye 60 0 0 10 0 0 0\n")
    ds = pt.DeviceScene(pt.HostScene(str(sdf)), 0, 1 << 26)
    r = ds.render_image(16, 8, 2, 5)
    assert not r["color"].any() and not r["rgb"].any()
    a = ds.intersect([[0, 0, 0]], [[0, 0, -1]])
    assert a["prim"][0] == -1
    # a lone mesh, rays along coordinate axes (zero direction components) and from inside bounds
    sdf = tmp_path / "mesh.sdf"
    sdf.write_text("define shape obj sword\ndefine light l 0 20 20 1 1 1 1\ndefine camera eye 60 0 1 8 0 0 0\n")
    hs = pt.HostScene(str(sdf), pt.SDF_DIR + "/")
    ds = pt.DeviceScene(hs, 0, 1 << 26)
    import sys
    from oracle_lib import Port
    port = Port()
    s = port.load(str(sdf), pt.SDF_DIR + "/")
    o = np.array([[0, 1, 8], [0, 0.5, 0], [0.1, 5, 0.05], [3, 0.3, 0], [0, 0, 0]], np.float32)
    d = np.array([[0, 0, -1], [0, 1, 0], [0, -1, 0], [-1, 0, 0], [0, 0, 0]], np.float32)
    a, b = ds.intersect(o, d), port.intersect(s, o, d)
    assert np.array_equal(a["prim"], b["rank"])
    assert beq(a["t"][b["hit"]], b["t"][b["hit"]])
    # bad arguments
    with pytest.raises(pt.PtxError):
        ds.render_image(0, 8, 1, 1)
    with pytest.raises(pt.PtxError):
        ds.render_image(8, 8, 2, 1, 3, 9)


def test_deep_two_child_trees_in_small_queues(pt, port, tmp_path):
    """Every hit on this scene spawns a reflection AND a refraction ray.  With the smallest
    queue a depth holds more rays than one launch may take, so the library walks the rest of
    the ray tree depth-first in chunks (the host reading the counters after every launch);
    with a large queue every depth is one launch.  Both must give the reference's image, and
    none may drop a ray."""
    sdf = tmp_path / "glass.sdf"
    sdf.write_text("define material glass 0 0 1 0 0 1 1 1 1 10 0.5 0.5 1.4\n"
                   "define shape sphere a 0 0 -5 4 glass\ndefine shape sphere b 0 0 -5 3 glass\n"
                   "define shape sphere c 0 0 -5 2 glass\ndefine shape box w -30 -30 -40 30 30 -30 glass\n"
                   "define light l 0 20 20 1 1 1 1\ndefine camera eye 40 0 0 10 0 0 0\n")
    hs = pt.HostScene(str(sdf))
    s = port.load(str(sdf))
    small = pt.DeviceScene(hs, 0, 1)        # clamps to the minimum queue memory
    big = pt.DeviceScene(hs, 0, 2 << 30)
    for w, h, aa, nb in ((96, 96, 2, 6), (48, 48, 1, 11)):
        ref = port.render(s, w, h, aa, nb, counters=True)
        for ds in (small, big):
            for pipeline in PIPELINES:
                r = ds.render_image(w, h, aa, nb, pipeline=pipeline)
                assert_image_close(r["color"], ref["color"], r["rgb"], ref["rgb"])
                assert r["stats"]["trace_rays"] > 3 * w * h * aa * aa
    # the two strategies and the two pipelines do the same work
    for pipeline in PIPELINES:
        a = small.render_image(96, 96, 2, 6, pipeline=pipeline)["stats"]
        b = big.render_image(96, 96, 2, 6, pipeline=pipeline)["stats"]
        assert a["trace_rays"] == b["trace_rays"] and a["shadow_rays"] == b["shadow_rays"]
        assert a["launches"] > b["launches"]
    f, s = big.render_image(96, 96, 2, 6, pipeline="fused")["stats"], big.render_image(96, 96, 2, 6, pipeline="staged")["stats"]
    assert (f["trace_rays"], f["shadow_rays"], f["shaded_hits"]) == (s["trace_rays"], s["shadow_rays"], s["shaded_hits"])


def test_general_specular_exponent_and_many_lights(pt, port, tmp_path):
    """Specular exponents that are not small whole numbers take the general pow kernels (the
    scenes of the reference only use whole numbers); three lights give three shadow rays per hit."""
    sdf = tmp_path / "shine.sdf"
    sdf.write_text("define ambient amb 1 1 1 0.1\n"
                   "define material a 1 0 0 1 0 0 1 1 1 12.5 0 1 1\n"
                   "define material b 0 1 0 0 1 0 1 1 1 0.5 0.3 1 1.3\n"
                   "define material c 0 0 1 0 0 1 1 1 1 100000 0 0.5 1.5\n"
                   "define material d 1 1 0 1 1 0 1 1 0 0 0 1 1\n"
                   "define shape sphere s1 -2 0 -8 1.5 a\ndefine shape sphere s2 2 0 -8 1.5 b\n"
                   "define shape box b1 -1 -3 -9 1 -2 -7 c\ndefine shape box floor -10 -5 -20 10 -4 0 d\n"
                   "define light l1 0 20 20 .6 .6 .6 1\ndefine light l2 -10 5 0 .3 .2 .2 1\ndefine light l3 10 5 0 .2 .2 .3 1\n"
                   "define camera eye 50 0 0 4 0 0 0\n")
    hs = pt.HostScene(str(sdf))
    ds = pt.DeviceScene(hs, 0, 1 << 28)
    s = port.load(str(sdf))
    for w, h, aa, nb in ((160, 100, 2, 4), (33, 17, 3, 0)):
        ref = port.render(s, w, h, aa, nb, counters=True)
        for pipeline in PIPELINES:
            r = ds.render_image(w, h, aa, nb, pipeline=pipeline)
            assert_image_close(r["color"], ref["color"], r["rgb"], ref["rgb"])
            assert r["stats"]["shadow_rays"] > r["stats"]["shaded_hits"]   # more than one light per hit on average


def test_many_shapes_fixture(pt, port):
    """fixtures/sdf/many_shapes.sdf: 60 frame-0 leaves of all kinds under the reference's root (not yet split), 40
    materials, two lights, specular exponents that are not whole numbers - the scene class that takes the shading
    kernels without LDS tables and with the general pow, and the traversal kernels with a BVH over frame 0."""
    from oracle_lib import random_rays
    path = f"{pt.SDF_DIR}/many_shapes.sdf"
    hs = pt.HostScene(path)
    ds, s = pt.DeviceScene(hs, 0, 1 << 28), port.load(path)
    o1, d1 = port.primary_rays(s, 160, 90, 1)
    o2, d2 = random_rays(30000, 17, lo=(-8, -4, -29), hi=(8, 8, 3))
    o, d = np.concatenate([o1, o2]), np.concatenate([d1, d2])
    a, b = ds.intersect(o, d), port.intersect(s, o, d)
    assert np.array_equal(a["prim"], b["rank"]) and b["hit"].sum() > 40000
    for k in ("t", "pos", "raydir", "normal"):
        assert beq(a[k][b["hit"]], b[k][b["hit"]]), k
    ref = port.render(s, 160, 90, 2, 5)
    for pipeline in PIPELINES:
        r = ds.render_image(160, 90, 2, 5, pipeline=pipeline)
        assert_image_close(r["color"], ref["color"], r["rgb"], ref["rgb"])
    ds.close()


def test_full_size_properties(pt, scenes):
    """BASELINE config 2 at its full size (example.sdf 1920x1080, 1024 spp, 5 bounces), where
    the CPU oracle would take minutes: the two pipelines agree with each other to the image
    bar, do exactly the same work, and the image is the sum of its sample ranges."""
    import torch
    ds = pt.DeviceScene(scenes.host_scene("example"), 0, 8 << 30)
    w, h, aa, nb = 1920, 1080, 32, 5
    a = ds.render_image(w, h, aa, nb, pipeline="staged")
    b = ds.render_image(w, h, aa, nb, pipeline="fused")
    assert_image_close(a["color"], b["color"], a["rgb"], b["rgb"])
    sa, sb = a["stats"], b["stats"]
    assert sa["samples"] == w * h * aa * aa == sa["per_depth_rays"][0]
    for k in ("trace_rays", "shadow_rays", "shaded_hits"):
        assert sa[k] == sb[k], k
    assert sa["per_depth_rays"][:nb + 1] == sb["per_depth_rays"][:nb + 1]
    # sample ranges as eight ranks would take them, accumulated into one buffer
    dist = __import__("importlib").import_module("path-tracer_amd.dist")
    acc = torch.zeros(h * w * 3, dtype=torch.float32, device="cuda:0")
    stream = torch.cuda.current_stream().cuda_stream
    for rank in range(8):
        s0, s1 = dist.sample_range(rank, 8, aa * aa)
        ds.render_accumulate(pt.render_params(w, h, aa, nb, s0, s1), acc.data_ptr(), stream, check_overflow=True)
    color = torch.empty(h * w * 3, dtype=torch.float32, device="cuda:0")
    rgb = torch.empty(h * w * 3, dtype=torch.uint8, device="cuda:0")
    ds.resolve(acc.data_ptr(), w, h, aa * aa, color.data_ptr(), rgb.data_ptr(), stream)
    torch.cuda.synchronize()
    assert_image_close(color.cpu().numpy().reshape(h, w, 3), a["color"], rgb.cpu().numpy().reshape(h, w, 3), a["rgb"])
    # the background (no hit at any depth) is exactly black, and nothing is NaN
    assert np.isfinite(a["color"]).all() and (a["color"] >= 0).all()


def test_jittered_sampling(pt, port, scenes):
    """jitter_seed != 0 (an option of this library; the reference only has the regular grid):
    the kernels draw the same per-sample offsets as ptx_jitter() publishes, the image equals
    the CPU restatement of that sampling to the usual bar, stays within 2/255 RMSE of the
    grid image at equal spp (64 spp here; the two converge to the same image), and is the sum
    of its sample ranges whoever renders them."""
    import ctypes
    import torch
    L = pt.ptx_lib()
    dx, dy = ctypes.c_float(), ctypes.c_float()
    seen = set()
    for pix, smp in ((0, 0), (0, 1), (1, 0), (123456, 789)):
        L.ptx_jitter(7, pix, smp, ctypes.byref(dx), ctypes.byref(dy))
        assert 0.0 <= dx.value < 1.0 and 0.0 <= dy.value < 1.0
        seen.add((dx.value, dy.value))
    assert len(seen) == 4
    L.ptx_jitter(0, 5, 5, ctypes.byref(dx), ctypes.byref(dy))
    assert dx.value == 0.0 and dy.value == 0.0
    w, h, aa, nb = 96, 60, 8, 5
    ds, s = scenes.device_scene("cornell"), scenes.oracle_scene("cornell")
    grid = ds.render_image(w, h, aa, nb)
    ref = port.render(s, w, h, aa, nb, jitter_seed=7)
    for pipeline in PIPELINES:
        r = ds.render_image(w, h, aa, nb, pipeline=pipeline, jitter_seed=7)
        rmse, mx, frac, worst = image_errors(r["color"], ref["color"], r["rgb"], ref["rgb"])
        # (flat walls: a wall whose value sits on a rounding boundary of the 8-bit quantisation flips
        #  as a whole with the last bit of the float sum, so the share of differing bytes is not a
        #  bar here; no byte differs by more than 1 and the floats agree to 1e-5)
        assert rmse <= RMSE_MAX and mx <= ABS_MAX and worst <= 1, (rmse, mx, frac, worst)
    other = ds.render_image(w, h, aa, nb, jitter_seed=8)
    assert not np.array_equal(other["color"], r["color"])
    rmse = float(np.sqrt(np.mean((r["color"].astype(np.float64) - grid["color"]) ** 2)))
    assert 0 < rmse <= 2.0 / 255.0, rmse
    acc = torch.zeros(h * w * 3, dtype=torch.float32, device="cuda:0")
    stream = torch.cuda.current_stream().cuda_stream
    for b, e in ((0, 3), (3, 41), (41, 64)):
        ds.render_accumulate(pt.render_params(w, h, aa, nb, b, e, jitter_seed=7), acc.data_ptr(), stream, check_overflow=True)
    color = torch.empty(h * w * 3, dtype=torch.float32, device="cuda:0")
    ds.resolve(acc.data_ptr(), w, h, aa * aa, color.data_ptr(), None, stream)
    torch.cuda.synchronize()
    assert float(np.abs(color.cpu().numpy().reshape(h, w, 3) - r["color"]).max()) <= ABS_MAX


def test_many_meshes(pt, port, tmp_path):
    """36 mesh frames (more than the 31 whose bounds tests fit the closest-hit kernel's bit
    mask), overlapping bounds, a mirror sphere to send incoherent rays through them."""
    import shutil
    for k in range(36):
        shutil.copy(pt.SDF_DIR + "/sword.obj", tmp_path / f"m{k:02d}.obj")
    shutil.copy(pt.SDF_DIR + "/sword.mtl", tmp_path / "sword.mtl")
    lines = ["define material mirror 0 0 0 0 0 0 1 1 1 50 0.9 1 1", "define material wall .8 .8 .8 .8 .8 .8 0 0 0 0 0 1 1",
             "define shape sphere ball 0 1 -2 1.2 mirror", "define shape box floor -20 -3 -30 20 -2.5 10 wall"]
    for k in range(36):
        lines.append(f"define shape obj m{k:02d}")
    lines += ["define light sun 5 20 10 .8 .8 .8 1", "define ambient amb 1 1 1 0.2", "define camera eye 60 0 1 8 0 -5 0"]
    for k in range(36):
        x, z = (k % 6 - 2.5) * 1.3, -(k // 6) * 1.5
        lines.append(f"transform m{k:02d} translate {x} {0.2 * (k % 3)} {z}")
        if k % 4 == 1:
            lines.append(f"transform m{k:02d} rotate {10 * k} 20 0")
        if k % 5 == 2:
            lines.append(f"transform m{k:02d} scale 1.5 0.7 1.1")
    sdf = tmp_path / "swords.sdf"
    sdf.write_text("\n".join(lines) + "\n")
    hs = pt.HostScene(str(sdf), str(tmp_path) + "/")
    ds = pt.DeviceScene(hs, 0, 1 << 28)
    assert ds.info()["n_frames"] == 37
    s = port.load(str(sdf), str(tmp_path) + "/")
    from oracle_lib import random_rays
    o, d = port.primary_rays(s, 160, 100, 1)
    o2, d2 = random_rays(20000, 7)
    o, d = np.concatenate([o, o2]), np.concatenate([d, d2])
    a, b = ds.intersect(o, d), port.intersect(s, o, d)
    assert np.array_equal(a["prim"], b["rank"])
    m = b["hit"]
    assert m.sum() > 5000
    for k in ("t", "pos", "normal", "raydir"):
        assert beq(a[k][m], b[k][m]), k
    ref = port.render(s, 160, 100, 2, 4)
    for pipeline in PIPELINES:
        r = ds.render_image(160, 100, 2, 4, pipeline=pipeline)
        assert_image_close(r["color"], ref["color"], r["rgb"], ref["rgb"])


def test_deepest_ray_trees(pt, port, tmp_path):
    """max_ray_bounces at the library's limit (63) in a hall of mirrors (one child per hit), with
    the smallest queue memory: 64 queues share it, every depth is still traced, nothing is dropped."""
    sdf = tmp_path / "mirrors.sdf"
    sdf.write_text("define material mirror 0.1 0.1 0.1 0.2 0.2 0.2 0.95 0.95 0.95 20 1 1 1\n"
                   "define material red 1 0 0 1 0 0 1 1 1 10 0 1 1\n"
                   "define shape box left -6 -5 -30 -5 5 10 mirror\ndefine shape box right 5 -5 -30 6 5 10 mirror\n"
                   "define shape box back -6 -5 -31 6 5 -30 mirror\ndefine shape box front -6 -5 10 6 5 11 mirror\n"
                   "define shape box floor -6 -6 -31 6 -5 11 mirror\ndefine shape box ceil -6 5 -31 6 6 11 mirror\n"
                   "define shape sphere ball 1 0 -8 1.5 red\n"
                   "define light l 0 4 0 .8 .8 .8 1\ndefine ambient amb 1 1 1 0.1\ndefine camera eye 70 0 0 5 20 0 0\n")
    hs = pt.HostScene(str(sdf))
    s = port.load(str(sdf))
    small = pt.DeviceScene(hs, 0, 1)
    w, h, aa, nb = 40, 24, 2, 63
    ref = port.render(s, w, h, aa, nb, counters=True)
    for pipeline in PIPELINES:
        r = small.render_image(w, h, aa, nb, pipeline=pipeline)
        assert_image_close(r["color"], ref["color"], r["rgb"], ref["rgb"])
        assert r["stats"]["trace_rays"] > 20 * w * h * aa * aa      # the trees really are deep
        assert r["stats"]["trace_rays"] <= int(ref["counters"][0])
    with pytest.raises(pt.PtxError):
        small.render_image(w, h, aa, 64)


def test_scene_lifecycle(pt, scenes):
    """Scenes are independent objects: creating, using and destroying many of them, and using two
    alternately, gives the same images every time and leaves no device memory behind."""
    import torch
    hs_a, hs_b = scenes.host_scene("cornell"), scenes.host_scene("example_fixed")
    ref_a = scenes.device_scene("cornell").render_image(64, 40, 2, 3)
    ref_b = scenes.device_scene("example_fixed").render_image(64, 40, 2, 3)
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info()[0]
    for k in range(12):
        a = pt.DeviceScene(hs_a, 0, 1 << 27)
        b = pt.DeviceScene(hs_b, 0, 1 << 27)
        ra = a.render_image(64, 40, 2, 3, pipeline=PIPELINES[k % 2])
        rb = b.render_image(64, 40, 2, 3, pipeline=PIPELINES[(k + 1) % 2])
        ra2 = a.render_image(64, 40, 2, 3)
        assert_image_close(ra["color"], ref_a["color"], ra["rgb"], ref_a["rgb"])
        assert_image_close(rb["color"], ref_b["color"], rb["rgb"], ref_b["rgb"])
        assert_image_close(ra2["color"], ref_a["color"], ra2["rgb"], ref_a["rgb"])
        a.close()
        b.close()
    torch.cuda.synchronize()
    assert torch.cuda.mem_get_info()[0] >= free0 - (64 << 20)


def test_example_program(pt, tmp_path):
    """The reference's demo main rebuilt against this library (csrc/host/example.cpp): a C++
    program with no Python in it loads cornell.sdf, renders, writes the reference's PPM."""
    import os
    import subprocess
    exe = os.path.join(os.path.dirname(pt.__file__), "pt_example")
    assert os.path.exists(exe), "build it with make -C path-tracer_amd/csrc"
    g = load_golden("cornell")
    w, h, aa, nb = (int(v) for v in g["params"])
    out = tmp_path / "img.ppm"
    r = subprocess.run([exe, pt.SDF_DIR + "/cornell.sdf", str(out), str(w), str(h), str(aa), str(nb)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "shapes" in r.stdout and "s rendering" in r.stdout
    text = out.read_bytes()
    vals = np.array(text.split()[4:], dtype=np.int32).astype(np.uint8).reshape(h, w, 3)
    bd = np.abs(vals.astype(int) - g["rgb"].astype(int))
    assert (bd > 0).mean() <= 1e-3 and bd.max() <= 1
    r = subprocess.run([exe, str(tmp_path / "missing.sdf"), str(out)], capture_output=True, text=True, timeout=60)
    assert r.returncode == 1 and "Exception" in r.stdout


def test_many_root_shapes(pt, port, tmp_path):
    """A frame 0 with hundreds of shapes of all three kinds (some glass, some mirrors; the
    triangles rotated, scaled, translated): past 16 leaves the library walks a BVH of their
    bounding boxes instead of the list; hits stay bit-identical to the reference, images within
    the bar, whichever pipeline is asked for.  With more than 64 children the reference's root is
    split into octants (composite.cpp:120-161); the description lists them and the library answers
    through them (a split root always takes the staged kernels).  Spheres and boxes stay
    untransformed here, so that every shape lies inside its octants;
    test_split_root_is_answered_through_its_octants (test_gpu_shapes.py) moves them."""
    rng = np.random.default_rng(5)
    lines = ["define material a 1 0 0 1 0 0 1 1 1 20 0 1 1", "define material b 0 1 0 0 1 0 1 1 1 8 0.4 1 1.3",
             "define material c 0 0 1 0 0 1 1 1 1 50 0.1 0.3 1.5", "define material d .8 .8 .8 .8 .8 .8 0 0 0 0 0 1 1"]
    mats = "abcd"
    tf = []
    for k in range(300):
        x, y, z = rng.uniform(-12, 12), rng.uniform(-6, 6), rng.uniform(-40, -8)
        m = mats[k % 4]
        kind = k % 3
        if kind == 0:
            lines.append(f"define shape sphere s{k:03d} {x:.3f} {y:.3f} {z:.3f} {rng.uniform(0.2, 0.9):.3f} {m}")
        elif kind == 1:
            e = rng.uniform(0.2, 0.9, 3)
            lines.append(f"define shape box b{k:03d} {x:.3f} {y:.3f} {z:.3f} {x + e[0]:.3f} {y + e[1]:.3f} {z + e[2]:.3f} {m}")
        else:
            p = rng.uniform(-0.8, 0.8, 9)
            lines.append("define shape triangle t%03d " % k + " ".join(f"{v:.3f}" for v in (p[0] + x, p[1] + y, p[2] + z, p[3] + x, p[4] + y, p[5] + z, p[6] + x, p[7] + y, p[8] + z)) + f" {m}")
        name = "sbt"[kind] + f"{k:03d}"
        if kind != 2:
            continue
        if k % 5 == 1:
            tf.append(f"transform {name} rotate {rng.uniform(0, 90):.1f} {rng.uniform(0, 90):.1f} 0")
        if k % 7 == 2:
            tf.append(f"transform {name} scale 1.3 0.8 1.1")
        if k % 11 == 3:
            tf.append(f"transform {name} translate 0.5 -0.25 1")
    lines += ["define light sun 5 20 10 .8 .8 .8 1", "define light fill -10 5 5 .2 .2 .3 1", "define ambient amb 1 1 1 0.2",
              "define camera eye 60 0 0 5 0 0 0"] + tf
    sdf = tmp_path / "many.sdf"
    sdf.write_text("\n".join(lines) + "\n")
    hs = pt.HostScene(str(sdf))
    ds = pt.DeviceScene(hs, 0, 1 << 28)
    assert ds.info()["n_bvh_nodes"] > 50          # the root BVH is there
    s = port.load(str(sdf))
    from oracle_lib import random_rays
    o, d = port.primary_rays(s, 200, 120, 1)
    o2, d2 = random_rays(30000, 11)
    o2 = o2 + np.array([0, 0, -20], np.float32)
    o, d = np.concatenate([o, o2]), np.concatenate([d, d2])
    a, b = ds.intersect(o, d), port.intersect(s, o, d)
    assert np.array_equal(a["prim"], b["rank"])
    m = b["hit"]
    assert m.sum() > 8000
    for k in ("t", "pos", "normal", "raydir"):
        assert beq(a[k][m], b[k][m]), k
    ref = port.render(s, 160, 100, 2, 4)
    for pipeline in PIPELINES:
        r = ds.render_image(160, 100, 2, 4, pipeline=pipeline)
        assert_image_close(r["color"], ref["color"], r["rgb"], ref["rgb"])


def test_reference_unit_test_cases_on_device(pt, port, tmp_path):
    """The cases of the reference's own source/tests.cpp as one-shape scenes on the device (the
    shape sits under the root Composite, which only re-normalises the normal), with the values
    the code gives today: every intersector subtracts EPSILON = 0.001 from t."""
    eps = np.float32(0.001)
    mat = "define material m 1 1 1 1 1 1 0 0 0 0 0 1 1\n"

    def scene(text):
        f = tmp_path / ("s%d.sdf" % abs(hash(text)))
        f.write_text(mat + text)
        return pt.DeviceScene(pt.HostScene(str(f)), 0, 1 << 27), port.load(str(f))

    def both(ds_s, o, d):
        ds, s = ds_s
        a, b = ds.intersect(o, d), port.intersect(s, o, d)
        assert np.array_equal(a["prim"], b["rank"])
        m = b["hit"]
        for k in ("t", "pos", "normal", "raydir"):
            assert beq(a[k][m], b[k][m]), k
        return a

    # tests.cpp:76-96 intersect_ray_sphere2: hit at 8 - eps, miss, tangent
    for cx, expect in ((0, 7.999), (4, None), (2, 9.999)):
        a = both(scene(f"define shape sphere s {cx} 0 10 2 m\n"), [[0, 0, 0]], [[0, 0, 1]])
        if expect is None:
            assert a["prim"][0] == -1
        else:
            assert a["prim"][0] == 0 and abs(float(a["t"][0]) - expect) < 1e-3
    a = both(scene("define shape sphere s 0 0 10 2 m\n"), [[0, 0, 0]], [[0, 0, 1]])
    assert a["t"][0] == np.float32(8.0) - eps and np.allclose(a["pos"][0], [0, 0, 7.999], atol=1e-6)
    # tests.cpp:112-142 box_ray_intersection
    a = both(scene("define shape box b -10 -10 -10 10 10 10 m\n"),
             [[5, 5, -20], [20, 20, 0], [-20, -20, -20], [10, 30, 10], [0, 0, 20]], [[0, 0, 1], [0, 0, 1], [1, 20, 1], [-1, -1, -1], [0, 0, 1]])
    assert list(a["prim"] >= 0) == [True, False, False, True, False]
    assert a["t"][0] == np.float32(10.0) - eps and a["t"][3] == np.float32(20.0) - eps
    assert np.allclose(a["pos"][0], [5, 5, -10], atol=1.1e-3) and np.allclose(a["pos"][3], [-10, 10, -10], atol=1.1e-3)
    # tests.cpp:144-158 triangle_ray_intersection
    a = both(scene("define shape triangle t -1 0 -1 0 0 1 1 0 -1 m\n"), [[0, 10, 0], [1, 10, 1], [10, 0, 0]], [[0, -1, 0], [0, -1, 0], [-1, 0, 0]])
    assert list(a["prim"] >= 0) == [True, False, False]
    # tests.cpp:160-176 composite_ray_intersection: the nearer child wins
    ds_s = scene("define shape sphere back 0 0 -1 1 m\ndefine shape box front -1 -1 0 1 1 2 m\n")
    a = both(ds_s, [[0, 10, 1], [0, 10, -1]], [[0, -1, 0], [0, -1, 0]])
    assert a["prim"][0] != a["prim"][1] and (a["prim"] >= 0).all()
    # tests.cpp:252-259 scale sphere: position (0,0,2.001), normal (0,0,1)
    a = both(scene("define shape sphere s 0 0 0 1 m\ntransform s scale 2 2 2\n"), [[0, 0, 10]], [[0, 0, -1]])
    assert a["prim"][0] == 0 and np.allclose(a["pos"][0], [0, 0, 2.001], atol=1e-5) and np.allclose(a["normal"][0], [0, 0, 1], atol=1e-6)
