"""The CPU oracle (oracle/cpu_ref.cpp) against the golden vectors the real reference produced
(tests/golden/, written by oracle/gen_golden.py from the reference's own sources).
Everything here is bit-exact: the oracle restates the reference's fp32 operation order."""
import hashlib

import os

import numpy as np
import pytest

from conftest import SCENES, beq, load_golden


@pytest.mark.parametrize("name", SCENES)
def test_oracle_image_equals_reference(port, scenes, name):
    g = load_golden(name)
    w, h, aa, nb = (int(v) for v in g["params"])
    r = port.render(scenes.oracle_scene(name), w, h, aa, nb)
    assert beq(r["color"], g["color"]), "Renderer::color_buffer() differs from the reference"
    assert np.array_equal(r["rgb"], g["rgb"])
    assert hashlib.md5(port.ppm_text(r["rgb"])).hexdigest() == str(g["ppm_md5"])


def test_oracle_config1_md5(port, scenes):
    """BASELINE config 1: cornell.sdf 480x320, 16 spp, 5 bounces; md5 recorded from the reference."""
    g = load_golden("config1_cornell_480x320_aa4")
    w, h, aa, nb = (int(v) for v in g["params"])
    r = port.render(scenes.oracle_scene("cornell"), w, h, aa, nb)
    assert np.array_equal(r["rgb"], g["rgb"])
    assert hashlib.md5(port.ppm_text(r["rgb"])).hexdigest() == str(g["ppm_md5"]) == "08e2d136a052bafcf2550b038edb801b"


@pytest.mark.parametrize("name", SCENES)
def test_oracle_hits_equal_reference(port, scenes, name):
    g = load_golden(name)
    s = scenes.oracle_scene(name)
    leaf = port.leaf_names(s)
    table = [str(x) for x in g["names"]]
    hw, hh = int(g["hit_w"]), int(g["hit_h"])
    o, d = port.primary_rays(s, hw, hh, 1)
    a = port.intersect(s, o, d)
    names = [leaf[r] if r >= 0 else "null" for r in a["rank"]]
    assert names == [table[i] for i in g["primary_name"]]
    assert beq(a["t"][a["hit"]], g["primary_t"][a["hit"]])
    a = port.intersect(s, g["rays_o"], g["rays_d"])
    names = [leaf[r] if r >= 0 else "null" for r in a["rank"]]
    assert names == [table[i] for i in g["rays_name"]]
    m = a["hit"]
    for k in ("t", "pos", "normal", "raydir"):
        assert beq(a[k][m], g["rays_" + k][m]), k


def test_oracle_primitives_equal_reference(port):
    g = load_golden("primitives")
    o, d = g["rays_o"], g["rays_d"]
    for i in range(int(g["n_prims"])):
        a = port.prim_intersect(int(g[f"kind{i}"]), g[f"p{i}"], g[f"xf{i}"], o, d)
        assert np.array_equal(a["hit"], g[f"hit{i}"])
        m = a["hit"]
        assert m.sum() > 20
        for k in ("t", "pos", "normal", "raydir"):
            assert beq(a[k][m], g[f"{k}{i}"][m]), (i, k)


def test_reference_unit_test_cases(port, tmp_path):
    """The cases of the reference's source/tests.cpp (same shapes, same rays), with the values the
    code gives today: every intersector subtracts EPSILON = 0.001 from t, so the file's own
    expectations 8.0 (tests.cpp:83) and 10 (tests.cpp:118) read 7.999 and 9.999."""
    eps = np.float32(0.001)
    ray = ([[0, 0, 0]], [[0, 0, 1]])
    # tests.cpp:76-96 intersect_ray_sphere2
    s0 = port.prim_intersect(0, [0, 0, 10, 2], [], *ray)
    assert s0["hit"][0] and s0["t"][0] == np.float32(8.0) - eps and np.allclose(s0["pos"][0], [0, 0, 7.999], atol=1e-6)
    s1 = port.prim_intersect(0, [4, 0, 10, 2], [], *ray)
    assert not s1["hit"][0]
    s2 = port.prim_intersect(0, [2, 0, 10, 2], [], *ray)
    assert s2["hit"][0] and abs(float(s2["t"][0]) - 9.999) < 1e-3
    g = load_golden("primitives")  # the reference's answers for the same sphere KAT layout
    a = port.prim_intersect(0, [0, 0, 0, 2], [], g["kat_o"], g["kat_d"])
    assert np.array_equal(a["hit"], g["kat_hit"]) and beq(a["t"], g["kat_t"])
    # tests.cpp:112-142 box_ray_intersection
    box = [-10, -10, -10, 10, 10, 10]
    b = port.prim_intersect(1, box, [], [[5, 5, -20], [20, 20, 0], [-20, -20, -20], [10, 30, 10], [0, 0, 20]],
                            [[0, 0, 1], [0, 0, 1], [1, 20, 1], [-1, -1, -1], [0, 0, 1]])
    assert list(b["hit"]) == [True, False, False, True, False]
    assert b["t"][0] == np.float32(10.0) - eps and np.allclose(b["pos"][0], [5, 5, -10], atol=1.1e-3)
    assert b["t"][3] == np.float32(20.0) - eps and np.allclose(b["pos"][3], [-10, 10, -10], atol=1.1e-3)
    # tests.cpp:144-158 triangle_ray_intersection
    t = port.prim_intersect(2, [-1, 0, -1, 0, 0, 1, 1, 0, -1], [], [[0, 10, 0], [1, 10, 1], [10, 0, 0]], [[0, -1, 0], [0, -1, 0], [-1, 0, 0]])
    assert list(t["hit"]) == [True, False, False]
    # tests.cpp:160-176 composite_ray_intersection: the nearer child wins
    sdf = tmp_path / "comp.sdf"
    sdf.write_text("define material m 1 1 1 1 1 1 0 0 0 0 0 1 1\n"
                   "define shape sphere back 0 0 -1 1 m\n"
                   "define shape box front -1 -1 0 1 1 2 m\n")
    sc = port.load(str(sdf))
    c = port.intersect(sc, [[0, 10, 1], [0, 10, -1]], [[0, -1, 0], [0, -1, 0]])
    names = port.leaf_names(sc)
    assert [names[r] for r in c["rank"]] == ["front", "back"]
    # tests.cpp:235-250 transform box: Box{{0,0,0},{1,1,3}}.rotate(PI/2, 0, 0)
    mn, mx = port.box_bounds([0, 0, 0, 1, 1, 3], [[2, np.float32(3.14159265) / 2, 0, 0]])
    assert np.allclose(mn, [0, 0, -1], atol=1e-6) and np.allclose(mx, [3, 1, 0], atol=1e-6)
    # tests.cpp:252-259 scale sphere: prints position (0,0,2.001), normal (0,0,1)
    s = port.prim_intersect(0, [0, 0, 0, 1], [[3, 2, 2, 2]], [[0, 0, 10]], [[0, 0, -1]])
    assert s["hit"][0] and np.allclose(s["pos"][0], [0, 0, 2.001], atol=1e-5) and np.allclose(s["normal"][0], [0, 0, 1], atol=1e-6)
    # tests.cpp:33-45 / box.cpp:14-16: min > max throws
    with pytest.raises(ValueError):
        port.prim_intersect(1, [1, 0, 0, 0, 1, 1], [], [[0, 0, 0]], [[0, 0, 1]])


def test_reference_side_flatten_equals_the_mirror():
    """The scene hand-over written against the reference's own classes (oracle/ref_flatten.cpp,
    compiled with the reference) and the product's host mirror give byte-identical
    PtxSceneDesc arrays for every fixture scene (oracle/check_flatten.py).  Needs the reference
    built into oracle/_ref/libref.so, which exists where the reference checkout was present."""
    import subprocess
    import sys
    from conftest import REPO
    from oracle_lib import Reference
    if not Reference.available():
        pytest.skip("oracle/_ref/libref.so absent")
    r = subprocess.run([sys.executable, os.path.join(REPO, "oracle", "check_flatten.py")], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "IDENTICAL" in r.stdout, r.stdout + r.stderr


@pytest.mark.parametrize("seed", range(int(os.environ.get("PTX_FUZZ_FIRST", "0")), int(os.environ.get("PTX_FUZZ_FIRST", "0")) + int(os.environ.get("PTX_FUZZ_SEEDS", "18"))))   # (PTX_FUZZ_FIRST / PTX_FUZZ_SEEDS: a longer sweep)
def test_oracle_equals_reference_on_generated_scenes(port, tmp_path, seed):
    """The generated scenes of the device parity tests (tests/fuzz_scenes.py: leaf lists of 3 to 130
    shapes with random and quarter-turn transforms, split roots, meshes, every material case) through
    the real reference and through the oracle: the same hit object, bit-identical distance,
    position, normal and ray direction on primary rays, random rays and rays with zero direction
    components, and a bit-identical small image.  Container-only (needs oracle/_ref/libref.so); it
    is what makes the oracle a pinned checker on these scenes too."""
    from fuzz_scenes import apply_moves, mesh_prefix, rays, same_bits_or_both_nan, write_scene
    from oracle_lib import Reference
    if not Reference.available():
        pytest.skip("oracle/_ref/libref.so absent")
    ref = Reference()
    from oracle_lib import SDF_DIR
    sdf, obj_dir = write_scene(seed, tmp_path, SDF_DIR)
    s, r = port.load(sdf, obj_dir), ref.load(sdf)
    where = mesh_prefix(port, s, seed)
    apply_moves(port, s, seed)      # (some scenes: parts of the mesh moved after loading, a Composite of the caller's own)
    apply_moves(ref, r, seed, where)
    leaf = port.leaf_names(s)
    o1, d1 = port.primary_rays(s, 60, 40, 1)
    o2, d2 = rays(seed, 4000, 1500)
    o, d = np.concatenate([o1, o2]), np.concatenate([d1, d2])
    a, b = port.intersect(s, o, d), ref.intersect(r, o, d)
    assert np.array_equal(a["hit"], b["hit"])
    m = a["hit"]
    assert m.sum() > 100
    assert [leaf[k] for k in a["rank"][m]] == [n for n, h in zip(b["names"], m) if h]
    for k in ("t", "pos", "raydir", "normal"):      # a NaN (box hit that matches no face, box.cpp:167-182; rays with a NaN direction) is a NaN in both
        assert same_bits_or_both_nan(a[k][m], b[k][m]), k
    assert m[-16:].all()       # the reference answers a ray whose direction is NaN with a hit
    ia, ib = port.render(s, 72, 48, 2, 4), ref.render(r, 72, 48, 2, 4, ppm_path=str(tmp_path / "ref.ppm"))
    assert np.array_equal(ia["rgb"], ib["rgb"])
    ca, cb = ia["color"], ib["color"]
    assert ((ca.view(np.uint32) == cb.view(np.uint32)) | (np.isnan(ca) & np.isnan(cb))).all()


@pytest.mark.parametrize("name", ["odd_syntax", "odd_mesh"])
def test_oracle_equals_reference_on_hand_written_fixtures(port, tmp_path, name):
    """fixtures/sdf/odd_syntax.sdf (number formats and layout the reference's stream parsing accepts, a
    duplicate name, a negative radius) and odd_mesh.sdf (an OBJ with faces in all four index forms, faces
    before the first object, a repeated object name, a quad; an MTL with illum 1 / 2 / 7, d, Ni): the
    reference's loader and the oracle's build the same scene - the same hits bit for bit, the same image.
    Container-only (needs oracle/_ref/libref.so)."""
    from fuzz_scenes import same_bits_or_both_nan
    from oracle_lib import Reference, SDF_DIR, random_rays
    if not Reference.available():
        pytest.skip("oracle/_ref/libref.so absent")
    ref = Reference()
    path = f"{SDF_DIR}/{name}.sdf"
    s, r = port.load(path), ref.load(path)
    leaf = port.leaf_names(s)
    o1, d1 = port.primary_rays(s, 90, 60, 1)
    o2, d2 = random_rays(12000, 3, lo=(-5, -3, -14), hi=(5, 5, 4))
    o, d = np.concatenate([o1, o2]), np.concatenate([d1, d2])
    a, b = port.intersect(s, o, d), ref.intersect(r, o, d)
    m = a["hit"]
    assert np.array_equal(m, b["hit"]) and m.sum() > 300
    assert [leaf[k] for k in a["rank"][m]] == [n for n, h in zip(b["names"], m) if h]
    for k in ("t", "pos", "raydir", "normal"):
        assert same_bits_or_both_nan(a[k][m], b[k][m]), k
    ia, ib = port.render(s, 90, 60, 2, 4), ref.render(r, 90, 60, 2, 4, ppm_path=str(tmp_path / "ref.ppm"))
    assert np.array_equal(ia["rgb"], ib["rgb"]) and same_bits_or_both_nan(ia["color"], ib["color"])


@pytest.mark.parametrize("grouped", [False, True, "split_root", "deep"])
def test_oracle_equals_reference_on_moved_mesh_parts(port, grouped):
    """nested.sdf with parts of its two loaded meshes moved after load_scene (tests/nested_ops.py:
    root->find_child(..)->find_child(..)->translate / rotate / scale with the reference's own methods):
    transformed Composites inside transformed Composites.  Reference and oracle: the same hit object,
    bit-identical distance, position, normal and ray direction on primary and random rays, a bit-identical image;
    and the scene hand-over written against the reference's classes equals the host mirror's byte for byte, with
    the chain of frames in it.  grouped: in addition a Composite of the caller's own with spheres and boxes in it, moved as
    a whole and member by member, and single faces of a moved mesh part moved once more (leaves that are not plain
    triangles inside transformed Composites).  split_root: the same kinds of moves on split_root.sdf, where the mesh is
    reached through the octants of a root with more than 64 children.  deep: twelve Composites of the caller's own one
    inside the other, each moved, a sphere or a box in each.  Container-only (needs oracle/_ref/libref.so)."""
    import importlib
    import sys
    from conftest import REPO
    from nested_ops import apply_deep, apply_group, apply_ops, apply_split
    if grouped:
        apply_ops = {"split_root": apply_split, "deep": apply_deep}.get(grouped, apply_group)     # noqa: F811
    from oracle_lib import Reference, SDF_DIR, random_rays
    if not Reference.available():
        pytest.skip("oracle/_ref/libref.so absent")
    sys.path.insert(0, os.path.join(REPO, "oracle"))
    import check_flatten as cf
    ref = Reference()
    path = os.path.join(SDF_DIR, "split_root.sdf" if grouped == "split_root" else "nested.sdf")
    s, r = port.load(path), ref.load(path)
    apply_ops(port, s)
    apply_ops(ref, r)
    leaf = port.leaf_names(s)
    o1, d1 = port.primary_rays(s, 160, 100, 1)
    o2, d2 = random_rays(40000, 7, lo=(-8, -1, -20), hi=(8, 10, 4))
    if grouped == "split_root":
        o2, d2 = random_rays(40000, 7, lo=(-12, -8, -32), hi=(12, 6, 4))
    o, d = np.concatenate([o1, o2]), np.concatenate([d1, d2])
    a, b = port.intersect(s, o, d), ref.intersect(r, o, d)
    assert np.array_equal(a["hit"], b["hit"])
    m = a["hit"]
    assert m.sum() > (5000 if grouped == "split_root" else 20000)
    assert [leaf[k] for k in a["rank"][m]] == [n for n, h in zip(b["names"], m) if h]
    for k in ("t", "pos", "raydir", "normal"):
        assert np.array_equal(a[k][m].view(np.uint32), b[k][m].view(np.uint32)), k
    ia, ib = port.render(s, 96, 64, 2, 5), ref.render(r, 96, 64, 2, 5)
    assert np.array_equal(ia["color"].view(np.uint32), ib["color"].view(np.uint32)) and np.array_equal(ia["rgb"], ib["rgb"])
    pt = importlib.import_module("path-tracer_amd")
    hs = pt.HostScene(path)
    apply_ops(hs, None)
    rp, rn = cf.ref_parts(ref, r)
    mp, mn = cf.mirror_parts(pt, path, hs)
    for part in cf.PARTS:
        assert rp[part] == mp[part], part
    assert rn == mn
    d = hs.desc.contents
    if not grouped:
        assert sorted(d.frames[f].parent for f in range(d.n_frames)) == [-1, 0, 0, 1, 1, 4, 4]
    elif grouped == "deep":
        assert max(d.frames[f].parent for f in range(d.n_frames)) >= 12
    elif grouped == "split_root":   # (the sword with three of its parts as frames inside it, and the group; all below a split root)
        assert d.n_guards >= 8 and sorted(d.frames[f].parent for f in range(d.n_frames)) == [-1, 0, 0, 1, 1, 1]
    else:   # (five more frames: the two untouched parts of the sword with a moved face each, a leg of the other mesh and the octant in it that holds the moved face ... and the group)
        assert d.n_frames >= 12 and max(d.frames[f].parent for f in range(d.n_frames)) > 5
