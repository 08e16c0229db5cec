"""Device parity on generated scenes (tests/fuzz_scenes.py), through the C ABI.

Eighteen scenes: 3 to 130 spheres, boxes and triangles under the root (the leaf list, the BVH over
the frame-0 leaves, roots the reference splits into octants), random and quarter-turn rotations,
even and uneven scales, meshes with turned and unevenly scaled transforms, with and without an
enclosing room, none to five lights, every shading case, four in fourteen scaled by 100, 10^4, 10^-2 or 10^-3; in seven of them parts of
the mesh are moved after loading (transformed Composites inside a transformed Composite; in two of the seven below a
root the reference splits into octants), in four a Composite of the caller's own with spheres and boxes in it is added
and moved.  The CPU oracle is checked against the
real reference on these very scenes in tests/test_oracle.py
(test_oracle_equals_reference_on_generated_scenes); here the device must give the oracle's
answers: hit identity exact, t / position / normal / ray_direction bit-identical - on primary
rays, random rays and rays with zero direction components (also negative zeros) - and images to
the usual bar in both pipelines."""
import os

import numpy as np
import pytest

from conftest import beq
from test_gpu_parity import PIPELINES, assert_image_close

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed", range(int(os.environ.get("PTX_FUZZ_FIRST", "0")), int(os.environ.get("PTX_FUZZ_FIRST", "0")) + int(os.environ.get("PTX_FUZZ_SEEDS", "18"))))   # (PTX_FUZZ_FIRST / PTX_FUZZ_SEEDS: a longer sweep)
def test_generated_scene(pt, port, tmp_path, seed):
    from fuzz_scenes import apply_moves, rays, same_bits_or_both_nan, write_scene
    sdf, obj_dir = write_scene(seed, tmp_path, pt.SDF_DIR)
    hs = pt.HostScene(sdf, obj_dir)
    apply_moves(hs, None, seed)     # (some scenes: parts of the mesh moved after loading, a Composite of the caller's own)
    ds = pt.DeviceScene(hs, 0, 1 << 28)
    s = port.load(sdf, obj_dir)
    apply_moves(port, s, seed)
    assert hs.prim_paths() == port.leaf_paths(s)
    o1, d1 = port.primary_rays(s, 120, 80, 1)
    o2, d2 = rays(seed)
    o, d = np.concatenate([o1, o2]), np.concatenate([d1, d2])
    a, b = ds.intersect(o, d), port.intersect(s, o, d)
    assert np.array_equal(a["prim"], b["rank"])
    m = b["hit"]
    assert m.sum() > 300
    # bit-identical; a NaN (the normal of a box hit that matches no face, box.cpp:167-182; everything about the
    # hit of a ray with a NaN direction) is a NaN in both
    for k in ("t", "pos", "raydir", "normal"):
        assert same_bits_or_both_nan(a[k][m], b[k][m]), k
    assert m[-16:].all() or not m[-16:].any()       # the NaN rays all get one answer
    w, h, aa, nb = (50, 77, 2, 5) if seed % 8 == 5 else ((96, 64, 2, 4), (97, 61, 3, 6), (64, 48, 1, 8))[seed % 3]      # (odd sizes, odd sample grids, deeper trees, upright)
    ref = port.render(s, w, h, aa, nb, counters=True)
    for pipeline in PIPELINES:
        r = ds.render_image(w, h, aa, nb, pipeline=pipeline)
        # NaN pixels (NaN normals) are the reference's.
        finite = np.isfinite(ref["color"]).all(axis=2)
        if not np.array_equal(np.isfinite(r["color"]).all(axis=2), finite):
            # the one case in which they are not: a child ray of weight exactly zero (not traced unless asked for)
            # whose subtree holds a NaN, and 0 * NaN is a NaN in the reference's sum
            assert np.isfinite(r["color"]).all(axis=2)[finite].all()
            r = ds.render_image(w, h, aa, nb, pipeline=pipeline, zero_weight=True)
            assert r["stats"]["trace_rays"] <= int(ref["counters"][0])      # (fewer where the reference reflects twice at one hit - total reflection inside glossy glass - and the kernels once with the summed weight)
        odd = np.isfinite(r["color"]).all(axis=2) != finite
        if odd.any():
            # and the float range: in scenes the size of the reference's EPSILON rays bounce between surfaces that
            # touch, ray directions (never normalised) and specular lobes grow with every bounce, and sample sums
            # reach 1e36 after eight.  The reference's products - light * ks * pow() at the leaf, then factor by
            # factor up the tree - pass 3.4e38 in between (an infinity; times a zero coefficient or minus another a NaN
            # in every channel) where the kernels' - the weight down the tree first - do not yet.  Such a pixel holds
            # a sum beyond 2^24 (exactly 1 after tone mapping) in the kernels' image.
            assert odd.sum() <= 2 and not finite[odd].any()
            assert (np.abs(r["color"][odd]) == 1).any(axis=1).all()
        assert finite.mean() > 0.3      # (ten thousand times larger, a box hit rarely matches a face within the reference's absolute tolerance: NaN normals)
        # Materials with random coefficients give samples of any sign and size (factors like 1 - R below zero, a
        # specular lobe of 40 000): a pixel whose samples cancel is ill-conditioned under the different order of
        # the sums (the reference adds up the tree, the kernels add weighted leaves), and tone mapping c / (c + 1)
        # has a pole at -1.  The image bar therefore applies to the well-conditioned pixels - colour sums in
        # [0, 1) after tone mapping, all but at most three in a thousand of them - and the others must agree before
        # tone mapping to 1e-3 of the sum.
        ca, cb = r["color"].astype(np.float64), ref["color"].astype(np.float64)
        ok = finite & (cb >= 0).all(axis=2) & (cb < 1).all(axis=2)
        assert ok.sum() > 0.5 * finite.sum()
        err = np.where(ok[..., None], np.abs(ca - cb), 0).max(axis=2)
        good = ok & (err <= 1e-5)
        assert good.sum() >= 0.997 * ok.sum()
        rows = good[::-1, :, None]      # (the 8-bit rows run top-down as in the PPM, the colour buffer bottom-up)
        assert_image_close(np.where(good[..., None], r["color"], 0), np.where(good[..., None], ref["color"], 0),
                           np.where(rows, r["rgb"], 0), np.where(rows, ref["rgb"], 0))
        rest = finite & ~good
        with np.errstate(divide="ignore", invalid="ignore"):
            ia, ib = ca[rest] / (1 - ca[rest]), cb[rest] / (1 - cb[rest])
        close = (np.isclose(ia, ib, rtol=1e-3, atol=2e-6) | np.isclose(ca[rest], cb[rest], rtol=0, atol=2e-7)).all(axis=1)      # (sums in the millions map to 1 - one unit in the last place)
        # what is left: terms of opposite sign and hundreds in size that cancel, between samples or within one (then in
        # every channel: the coefficients differ, the tree is the same) - measured against the largest sample sum
        ys, xs = np.nonzero(rest)
        assert (~close).sum() <= 5
        for y, x, da, db in zip(ys[~close], xs[~close], ia[~close], ib[~close]):
            size = np.abs(port.sample_colors(s, w, h, aa, nb, int(x), int(y), 1, 1)[0, 0].astype(np.float64)).sum(axis=0) / (aa * aa)
            assert (np.abs(da - db) <= 2e-5 * size.max() + 2e-6).all(), (x, y, da, db, size)
    if seed % 3 == 2:
        # a larger image of the deepest trees in the smallest work memory (96 MiB: a launch takes a slice of every
        # segment and the rest of a depth is traced in further launches) against the same image in ample memory
        W, H, AA = 640, 400, 2
        tiny = pt.DeviceScene(hs, 0, 1)
        big = ds.render_image(W, H, AA, nb, pipeline="staged")
        fin = np.isfinite(big["color"]).all(axis=2)
        for pipeline in PIPELINES:
            q = tiny.render_image(W, H, AA, nb, pipeline=pipeline)
            assert q["stats"]["trace_rays"] == big["stats"]["trace_rays"] and q["stats"]["shadow_rays"] == big["stats"]["shadow_rays"]
            odd = np.isfinite(q["color"]).all(axis=2) != fin
            if odd.any():       # (sums at the end of the float range, see above: infinite in one order of adding and not in another)
                assert odd.sum() <= 4
                assert (np.abs(np.where(fin[..., None], big["color"], q["color"])[odd]) == 1).any(axis=1).all()
            close = np.isclose(q["color"], big["color"], rtol=0, atol=1e-5).all(axis=2) | ~fin | odd
            with np.errstate(divide="ignore", invalid="ignore"):    # (sums near the pole of the tone mapping: compared before it)
                qa, qb = q["color"].astype(np.float64), big["color"].astype(np.float64)
                close |= np.isclose(qa / (1 - qa), qb / (1 - qb), rtol=1e-5, atol=0).all(axis=2)
            assert close.mean() >= 0.997        # (ill-conditioned pixels aside, see above)
        assert tiny.render_image(W, H, AA, nb, pipeline="staged")["stats"]["launches"] > big["stats"]["launches"]
        tiny.close()
    ds.close()


@pytest.mark.parametrize("seed", (4, 10, 17, 23))      # (two meshes; a split root with a mesh; five lights; deep trees)
def test_generated_scene_at_full_size(pt, tmp_path, seed):
    """1920x1080, 16 spp: both pipelines, ample and smallest work memory - the same counts per depth, the
    same image (scripts/fuzz_big.py sweeps more scenes the same way)."""
    from fuzz_scenes import full_size_findings
    assert full_size_findings(pt, seed, tmp_path) == []
