"""Device parity for single shapes and for transformed shapes at the root, through the C ABI.

* the reference's own answers for six primitives (tests/golden/primitives.npz: Sphere / Box /
  Triangle::intersect of the real reference on 1 500 rays each, three of them translated,
  rotated and scaled) replayed on the device as one-shape scenes;
* a root with rotated, scaled and rotated+scaled boxes and spheres, rays from outside and from
  inside the shapes, against the CPU oracle;
* BASELINE configs[2] (cornell + buster_sword mesh, 1920x1080, 1024 spp) at its full size through
  size-independent properties;
* a closed room of mirrors around a mesh with the smallest queue memory: nothing overflows;
* a root of more than 64 children, which the reference splits into octants by flawed bounds.

Bars as in test_gpu_parity.py: hit identity exact, t / position / normal / ray_direction
bit-identical, images RMSE <= 1e-6 and max abs <= 1e-5 on the tone-mapped floats.
"""
import numpy as np
import pytest

from conftest import beq, load_golden
from test_gpu_parity import PIPELINES, assert_image_close

pytestmark = pytest.mark.gpu
f32 = np.float32


def glm_normalize(v):
    """v * (1 / sqrt(dot(v, v))) in fp32 with GLM's evaluation order (func_geometric.inl:252-261),
    which is what the root Composite applies once more to a leaf's normal (composite.cpp:100)"""
    v = np.asarray(v, f32)
    x, y, z = v[:, 0], v[:, 1], v[:, 2]
    dot = (x * x + y * y).astype(f32) + (z * z).astype(f32)
    inv = (f32(1.0) / np.sqrt(dot.astype(f32))).astype(f32)
    return (v * inv[:, None]).astype(f32)


def test_reference_primitive_answers_on_device(pt):
    """Every case of primitives.npz as a one-shape scene: the shape under an untransformed root
    whose bounds contain everything (so that only the shape's own intersect decides), matrices
    from the host mirror's Shape::translate / rotate / scale.  hit / t / position / ray_direction
    must equal the reference's HitPoint bit for bit; the normal equals the reference's after the
    one normalize() the root Composite adds (composite.cpp:100)."""
    g = load_golden("primitives")
    o, d = g["rays_o"], g["rays_d"]
    tested = 0
    for i in range(int(g["n_prims"])):
        kind, p, xf = int(g[f"kind{i}"]), g[f"p{i}"], g[f"xf{i}"]
        w, wi = pt.shape_matrices(xf)
        prim = dict(kind=kind, p=p, world=w, world_inv=wi)
        if kind == 2:   # Triangle's n_ = normalize(cross(v1 - v0, v2 - v0)), framework/triangle.cpp:11
            e1, e2 = (p[3:6] - p[0:3]).astype(f32), (p[6:9] - p[0:3]).astype(f32)
            c = np.array([e1[1] * e2[2] - e2[1] * e1[2], e1[2] * e2[0] - e2[2] * e1[0], e1[0] * e2[1] - e2[0] * e1[1]], f32)
            prim["normal"] = glm_normalize(c[None])[0]
        ds = pt.DeviceScene(pt.DescScene([prim]), 0, 1 << 26)
        a = ds.intersect(o, d)
        hit = g[f"hit{i}"]
        assert np.array_equal(a["hit"], hit), i
        assert hit.sum() > 20
        for k in ("t", "pos", "raydir"):
            assert beq(a[k][hit], g[f"{k}{i}"][hit]), (i, k)
        assert beq(a["normal"][hit], glm_normalize(g[f"normal{i}"][hit])), i
        tested += int(hit.sum())
        ds.close()
    assert tested > 400
    # the hand-checkable sphere cases (hit / miss / tangent) with the reference's values
    ds = pt.DeviceScene(pt.DescScene([dict(kind=0, p=[0, 0, 0, 2])]), 0, 1 << 26)
    a = ds.intersect(g["kat_o"], g["kat_d"])
    assert np.array_equal(a["hit"], g["kat_hit"]) and beq(a["t"][g["kat_hit"]], g["kat_t"][g["kat_hit"]])


def transformed_root_scene(tmp_path):
    rng = np.random.default_rng(17)
    lines = ["define material a 1 0 0 1 0 0 1 1 1 20 0 1 1", "define material b 0 1 0 0 1 0 1 1 1 8 0.4 1 1.3",
             "define material c 0 0 1 0 0 1 1 1 1 50 0.1 0.3 1.5", "define material d .8 .8 .8 .8 .8 .8 0 0 0 0 0 1 1"]
    tf, names = [], []
    for k in range(40):     # <= 64 shapes under the root: the reference keeps them in one list
        x, y, z = rng.uniform(-6, 6), rng.uniform(-3, 3), rng.uniform(-20, -6)
        m = "abcd"[k % 4]
        if k % 2 == 0:
            e = rng.uniform(0.8, 2.5, 3)
            name = f"b{k:02d}"
            lines.append(f"define shape box {name} {-e[0] / 2:.3f} {-e[1] / 2:.3f} {-e[2] / 2:.3f} {e[0] / 2:.3f} {e[1] / 2:.3f} {e[2] / 2:.3f} {m}")
        else:
            name = f"s{k:02d}"
            lines.append(f"define shape sphere {name} 0 0 0 {rng.uniform(0.5, 1.5):.3f} {m}")
        names.append(name)
        tf.append(f"transform {name} translate {x:.3f} {y:.3f} {z:.3f}")
        mode = k % 8 // 2       # 0 rotate, 1 scale, 2 rotate + scale, 3 scale + rotate
        rot = f"transform {name} rotate {rng.uniform(-180, 180):.1f} {rng.uniform(-90, 90):.1f} {rng.uniform(-180, 180):.1f}"
        scl = f"transform {name} scale {rng.uniform(0.4, 2.0):.2f} {rng.uniform(0.4, 2.0):.2f} {rng.uniform(0.4, 2.0):.2f}"
        tf += {0: [rot], 1: [scl], 2: [rot, scl], 3: [scl, rot]}[mode]
    lines += ["define shape box room -9 -5 -24 9 5 2 d", "define light sun 0 4.5 -8 .8 .8 .8 1", "define light fill -6 2 0 .2 .2 .3 1",
              "define ambient amb 1 1 1 0.2", "define camera eye 60 0 0 1 0 0 0"] + tf
    sdf = tmp_path / "turned.sdf"
    sdf.write_text("\n".join(lines) + "\n")
    return str(sdf)


def test_rotated_and_scaled_root_shapes(pt, port, tmp_path):
    """Rotated, scaled, rotated+scaled and scaled+rotated boxes and spheres directly under the root
    (the branch of the root-leaf test that carries the ray through the shape's full inverse matrix,
    and the box / sphere branches of the surface rebuild under a rotation), inside one big box so
    that every ray ends somewhere.  Rays: primary rays, random rays from outside, and rays that
    start INSIDE each shape (the far root of a sphere, t_max of a box)."""
    from oracle_lib import random_rays
    sdf = transformed_root_scene(tmp_path)
    hs = pt.HostScene(sdf)
    ds = pt.DeviceScene(hs, 0, 1 << 28)
    assert ds.info()["n_bvh_nodes"] > 0 or hs.n_prims <= 16
    s = port.load(sdf)
    o1, d1 = port.primary_rays(s, 240, 160, 1)
    o2, d2 = random_rays(40000, 3, lo=(-8, -4, -22), hi=(8, 4, 1))
    # rays from inside the shapes: origins at the world-space centres (the translation of each transform)
    centres = []
    for line in open(sdf):
        q = line.split()
        if len(q) == 6 and q[0] == "transform" and q[2] == "translate":
            centres.append([float(v) for v in q[3:6]])
    centres = np.array(centres, f32)
    assert len(centres) == 40
    _, d3 = random_rays(40 * 200, 4)
    o3 = np.repeat(centres, 200, axis=0)
    o, d = np.concatenate([o1, o2, o3]), np.concatenate([d1, d2, d3])
    a, b = ds.intersect(o, d), port.intersect(s, o, d)
    assert np.array_equal(a["prim"], b["rank"])
    m = b["hit"]
    assert m.sum() > 0.9 * len(o)
    names = hs.prim_names()
    inside = a["prim"][len(o1) + len(o2):]
    own = sum(names[p] == "bs"[k // 200 % 2] + f"{k // 200:02d}" for k, p in enumerate(inside) if p >= 0)
    assert own > 0.6 * len(inside)          # most rays from a shape's centre hit that shape from the inside (the shapes overlap a little)
    for k in ("t", "pos", "normal", "raydir"):
        assert beq(a[k][m], b[k][m]), k
    hit_kinds = {names[p][0] for p in np.unique(a["prim"][m])}
    assert hit_kinds >= {"b", "s", "r"}
    ref = port.render(s, 160, 100, 2, 4)
    for pipeline in PIPELINES:
        r = ds.render_image(160, 100, 2, 4, pipeline=pipeline)
        assert_image_close(r["color"], ref["color"], r["rgb"], ref["rgb"])
    # the list variant of the closest-hit kernels (at most 16 root leaves) on a subset of the same shapes
    few = [l for l in open(sdf) if not any(f" {n} " in l or l.rstrip().endswith(" " + n) for n in [f"b{k:02d}" for k in range(12, 40, 2)] + [f"s{k:02d}" for k in range(13, 40, 2)])]
    sdf2 = tmp_path / "turned_few.sdf"
    sdf2.write_text("".join(few))
    hs2 = pt.HostScene(str(sdf2))
    assert hs2.n_prims == 13
    ds2, s2 = pt.DeviceScene(hs2, 0, 1 << 28), port.load(str(sdf2))
    assert ds2.info()["n_bvh_nodes"] == 0
    a, b = ds2.intersect(o, d), port.intersect(s2, o, d)
    assert np.array_equal(a["prim"], b["rank"])
    m = b["hit"]
    for k in ("t", "pos", "normal", "raydir"):
        assert beq(a[k][m], b[k][m]), k
    ref = port.render(s2, 160, 100, 2, 4)
    for pipeline in PIPELINES:
        r = ds2.render_image(160, 100, 2, 4, pipeline=pipeline)
        assert_image_close(r["color"], ref["color"], r["rgb"], ref["rgb"])


def test_full_size_properties_mesh(pt, port, scenes):
    """BASELINE configs[2] at its full size (cornell + buster_sword, 16 008 triangles, 1920x1080,
    1024 spp, 5 bounces: 2.1 G samples, 9 G rays; the CPU oracle would need hours).  The staged
    pipeline with the library's default queue memory (8 GiB: a hundred chunks of primary samples per image, the
    parked mesh walks under full load), the staged pipeline in 2 GiB (hundreds of chunks) and the
    fused pipeline agree to the image bar and count exactly the same rays, shadow rays and hits
    at every depth; the image is the sum of eight ranks' sample ranges; two windows of it equal the
    oracle's pixels at the full sample count."""
    import importlib
    import torch
    hs = scenes.host_scene("cornell_buster")
    w, h, aa, nb = 1920, 1080, 32, 5
    ds = pt.DeviceScene(hs, 0, 0)
    a = ds.render_image(w, h, aa, nb, pipeline="staged")
    b = ds.render_image(w, h, aa, nb, pipeline="fused")
    assert_image_close(a["color"], b["color"], a["rgb"], b["rgb"])
    sa, sb = a["stats"], b["stats"]
    assert sa["samples"] == w * h * aa * aa == sa["per_depth_rays"][0]
    for k in ("trace_rays", "shadow_rays", "shaded_hits"):
        assert sa[k] == sb[k], k
    for k in ("per_depth_rays", "per_depth_shadow", "per_depth_hits"):
        assert sa[k][:nb + 1] == sb[k][:nb + 1], k
    assert sa["per_depth_hits"][0] == sa["samples"]      # a closed room: every primary ray hits something
    dist = importlib.import_module("path-tracer_amd.dist")
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
    assert np.isfinite(a["color"]).all() and (a["color"] >= 0).all()
    # two windows of the full-size image against the oracle at the full 1024 samples per pixel: one on the mesh
    # and the glass box (deep trees), one on a wall
    s = scenes.oracle_scene("cornell_buster")
    for x0, y0 in ((1180, 420), (300, 800)):
        nx, ny = 6, 4
        sc = port.sample_colors(s, w, h, aa, nb, x0, y0, nx, ny)        # (ny, nx, spp, 3), un-tone-mapped
        mean = sc.astype(np.float64).sum(axis=2) / (aa * aa)
        want = (mean / (mean + 1)).astype(np.float32)                    # tone_map_color, renderer.cpp:286-291
        got = a["color"][y0:y0 + ny, x0:x0 + nx]
        assert float(np.sqrt(np.mean((got - want) ** 2))) <= 1e-6 and float(np.abs(got - want).max()) <= 1e-5
    ds.close()
    small = pt.DeviceScene(hs, 0, 2 << 30)
    c = small.render_image(w, h, aa, nb, pipeline="staged")
    assert_image_close(c["color"], a["color"], c["rgb"], a["rgb"])
    for k in ("trace_rays", "shadow_rays", "shaded_hits"):
        assert c["stats"][k] == sa[k], k
    assert c["stats"]["launches"] > 4 * sa["launches"]
    small.close()


def test_mirror_room_around_a_mesh_in_small_queues(pt, port, tmp_path):
    """A closed room whose six walls are mirrors, a mesh inside it, deep ray trees, the smallest
    queue memory: every ray hits something at every depth, and the rays that reach the mesh are
    parked and walked later by the closest-hit kernel, i.e. the hit list of a launch receives more
    batches of records than the launch has steps.  Nothing may overflow (the call checks), both
    pipelines give the reference's image and count the same rays."""
    import shutil
    shutil.copy(pt.SDF_DIR + "/sword.obj", tmp_path / "sword.obj")
    shutil.copy(pt.SDF_DIR + "/sword.mtl", tmp_path / "sword.mtl")
    sdf = tmp_path / "mirror_room.sdf"
    sdf.write_text("define material mirror 0.1 0.1 0.1 0.2 0.2 0.2 0.9 0.9 0.9 20 1 1 1\n"
                   "define shape box left -4 -3 -12 -3.5 3 4 mirror\ndefine shape box right 3.5 -3 -12 4 3 4 mirror\n"
                   "define shape box back -4 -3 -12.5 4 3 -12 mirror\ndefine shape box front -4 -3 4 4 3 4.5 mirror\n"
                   "define shape box floor -4 -3.5 -12.5 4 -3 4.5 mirror\ndefine shape box ceil -4 3 -12.5 4 3.5 4.5 mirror\n"
                   "define shape obj sword\n"
                   "define light l 0 2.5 -2 .8 .8 .8 1\ndefine ambient amb 1 1 1 0.1\ndefine camera eye 70 0 0 3 10 0 0\n"
                   "transform sword translate 0 -1 -6\ntransform sword rotate 30 20 0\ntransform sword scale 2 2 2\n")
    hs = pt.HostScene(str(sdf), str(tmp_path) + "/")
    s = port.load(str(sdf), str(tmp_path) + "/")
    small = pt.DeviceScene(hs, 0, 1)
    big = pt.DeviceScene(hs, 0, 1 << 30)
    w, h, aa, nb = 64, 40, 2, 24
    ref = port.render(s, w, h, aa, nb, counters=True)
    stats = []
    for ds in (small, big):
        for pipeline in PIPELINES:
            r = ds.render_image(w, h, aa, nb, pipeline=pipeline)     # (render_image checks the overflow flag)
            assert_image_close(r["color"], ref["color"], r["rgb"], ref["rgb"])
            st = r["stats"]
            assert st["per_depth_hits"][:10] == st["per_depth_rays"][:10]      # a closed room: every ray hits
            stats.append((st["trace_rays"], st["shadow_rays"], st["shaded_hits"]))
    assert len(set(stats)) == 1 and stats[0][0] > 10 * w * h * aa * aa


def test_split_root_is_answered_through_its_octants(pt, port, tmp_path):
    """More than 64 children directly under the root, some of them MOVED spheres and rotated boxes,
    and a mesh: the reference sorts its root's children into octants by bounds that ignore a sphere's
    transform and use two corners of a box (composite.cpp:120-161, sphere.cpp:22-28,
    box.cpp:42-66), and so loses a moved shape for every ray that does not cross an octant it was
    filed under; a child found through a deeper octant has its normal re-normalised once more per
    level (composite.cpp:100); and among equal distances the child met first wins.  The scene
    description lists the octants and the entries under them (PtxGuard / PtxOccurrence), and the
    library answers through them: hit identity, distance, position, normal and ray direction are
    those of the reference bit for bit, on primary rays and on random rays, for the fixture
    scene (120 shapes + a mesh that lies in several octants) and for a second, generated scene;
    images at the usual bar (shadow rays obey the octants as well).  The same description with the
    octants left out disagrees with the reference on some rays - the check has teeth."""
    import ctypes as C
    from oracle_lib import random_rays
    rng = np.random.default_rng(5)
    lines = ["define material a 1 0 0 1 0 0 1 1 1 20 0.3 1 1", "define material d .8 .8 .8 .8 .8 .8 0 0 0 0 0 1 1"]
    tf = []
    for k in range(150):
        x, y, z = rng.uniform(-10, 10), rng.uniform(-5, 5), rng.uniform(-30, -8)
        if k % 2 == 0:
            lines.append(f"define shape sphere s{k:03d} {x:.3f} {y:.3f} {z:.3f} {rng.uniform(0.3, 0.9):.3f} a")
            if k % 4 == 0:
                tf.append(f"transform s{k:03d} translate {rng.uniform(-6, 6):.2f} {rng.uniform(-3, 3):.2f} {rng.uniform(-6, 6):.2f}")
        else:
            e = rng.uniform(0.3, 0.9, 3)
            lines.append(f"define shape box b{k:03d} {x:.3f} {y:.3f} {z:.3f} {x + e[0]:.3f} {y + e[1]:.3f} {z + e[2]:.3f} d")
            if k % 4 == 1:
                tf.append(f"transform b{k:03d} rotate {rng.uniform(20, 160):.1f} {rng.uniform(10, 80):.1f} 0")
    gen = tmp_path / "generated.sdf"
    gen.write_text("\n".join(lines + ["define light sun 5 20 10 .8 .8 .8 1", "define camera eye 60 0 0 5 0 0 0"] + tf) + "\n")
    o2, d2 = random_rays(40000, 5, lo=(-12, -6, -32), hi=(12, 6, 4))
    for path in (pt.SDF_DIR + "/split_root.sdf", str(gen)):
        hs = pt.HostScene(path)
        desc = hs.desc.contents
        assert desc.n_guards >= 8 and desc.n_occurrences > desc.n_frames - 1 + sum(1 for i in range(desc.n_prims) if desc.prims[i].frame == 0)
        ds, s = pt.DeviceScene(hs, 0, 1 << 27), port.load(path)
        o1, d1 = port.primary_rays(s, 240, 160, 1)
        o, d = np.concatenate([o1, o2]), np.concatenate([d1, d2])
        a, b = ds.intersect(o, d), port.intersect(s, o, d)
        assert np.array_equal(a["prim"], b["rank"])
        m = b["hit"]
        assert m.sum() > 0.1 * len(o)
        for k in ("t", "pos", "normal", "raydir"):
            assert beq(a[k][m], b[k][m]), k
        ref = port.render(s, 120, 80, 2, 3)
        for pipeline in PIPELINES:          # (a split root always takes the staged kernels)
            r = ds.render_image(120, 80, 2, 3, pipeline=pipeline)
            assert_image_close(r["color"], ref["color"], r["rgb"], ref["rgb"])
        ds.close()
        # the same leaves without the octants: every shape tested wherever it really is
        bare = C.pointer(type(desc).from_buffer_copy(desc))
        bare.contents.n_guards = 0

        class Bare:
            pass
        hb = Bare()
        hb.desc = bare
        db = pt.DeviceScene(hb, 0, 1 << 27)
        c = db.intersect(o, d)
        differ = c["prim"] != b["rank"]
        assert 0 < differ.sum() < 0.05 * len(o)
        nearer = differ & (c["prim"] >= 0)
        assert (~b["hit"][nearer] | (c["t"][nearer] < b["t"][nearer])).all()       # what the octants hide lies in front of what they show
        db.close()


@pytest.mark.parametrize("name", ["odd_syntax", "odd_mesh"])
def test_hand_written_fixtures(pt, port, name):
    """The two hand-written fixtures (oddly formatted scene text; an OBJ with faces in every index form, triangles
    directly under the mesh and in sub-objects, MTL materials of every illumination model the loader knows) on
    the device against the oracle, which equals the reference on them (tests/test_oracle.py)."""
    from fuzz_scenes import same_bits_or_both_nan
    from oracle_lib import random_rays
    path = f"{pt.SDF_DIR}/{name}.sdf"
    hs = pt.HostScene(path)
    ds, s = pt.DeviceScene(hs, 0, 1 << 27), port.load(path)
    assert hs.prim_paths() == port.leaf_paths(s)
    o1, d1 = port.primary_rays(s, 150, 100, 1)
    o2, d2 = random_rays(30000, 3, lo=(-5, -3, -14), hi=(5, 5, 4))
    o, d = np.concatenate([o1, o2]), np.concatenate([d1, d2])
    a, b = ds.intersect(o, d), port.intersect(s, o, d)
    assert np.array_equal(a["prim"], b["rank"])
    m = b["hit"]
    assert m.sum() > 1000
    for k in ("t", "pos", "raydir", "normal"):
        assert same_bits_or_both_nan(a[k][m], b[k][m]), k
    ref = port.render(s, 150, 100, 2, 5)
    for pipeline in PIPELINES:
        r = ds.render_image(150, 100, 2, 5, pipeline=pipeline)
        assert_image_close(r["color"], ref["color"], r["rgb"], ref["rgb"])
    ds.close()


def test_full_size_4k_configs(pt, port, scenes):
    """BASELINE configs[3] (superhot + coloured buster, 3840x2160, 4096 spp: 34 G samples) in full on
    this one GPU, and of configs[4] (cornell + buster_sword, 3840x2160, 10 000 spp over 8 GPUs) the
    share of one rank (1 250 of every pixel's samples: 10.4 G samples).  Size-independent
    properties: the work counted equals the work of the parts (the image as four quarter ranges of
    samples, accumulated one after the other), primary rays = samples, a closed room leaves no
    pixel black and the open scene's background exactly black, nothing is NaN; the full image equals
    the sum of the parts to the image bar."""
    import importlib
    import torch
    dist = importlib.import_module("path-tracer_amd.dist")
    w, h, nb = 3840, 2160, 5
    stream = torch.cuda.current_stream().cuda_stream
    acc = torch.zeros(h * w * 3, dtype=torch.float32, device="cuda:0")
    color = torch.empty(h * w * 3, dtype=torch.float32, device="cuda:0")
    rgb = torch.empty(h * w * 3, dtype=torch.uint8, device="cuda:0")
    # ---- configs[3]
    aa = 64
    ds = pt.DeviceScene(scenes.host_scene("superhot_colored"), 0, 32 << 30)
    full = ds.render_image(w, h, aa, nb)
    st = full["stats"]
    assert st["samples"] == w * h * aa * aa == st["per_depth_rays"][0]
    parts = []
    for q in range(4):
        s0, s1 = dist.sample_range(q, 4, aa * aa)
        parts.append(ds.render_accumulate(pt.render_params(w, h, aa, nb, s0, s1), acc.data_ptr(), stream, check_overflow=True, want_stats=True))
    ds.resolve(acc.data_ptr(), w, h, aa * aa, color.data_ptr(), rgb.data_ptr(), stream)
    torch.cuda.synchronize()
    for k in ("samples", "trace_rays", "shadow_rays", "shaded_hits"):
        assert sum(p[k] for p in parts) == st[k], k
    assert_image_close(color.cpu().numpy().reshape(h, w, 3), full["color"], rgb.cpu().numpy().reshape(h, w, 3), full["rgb"])
    # (this scene has samples the REFERENCE computes as NaN - 0.5 % of the pixels of its own 320x180 image at
    # 144 spp hold one, the oracle check below - and a pixel with one such sample is NaN: such pixels are NaN
    # in both renders, which assert_image_close has just required, and stay a fraction of a percent)
    nan = np.isnan(full["color"]).any(axis=2)
    assert nan.mean() < 0.01 and not np.isinf(full["color"]).any() and (full["color"][~nan] >= 0).all()
    ref = port.render(scenes.oracle_scene("superhot_colored"), 320, 180, 12, nb)
    low = ds.render_image(320, 180, 12, nb)
    assert np.isnan(ref["color"]).any() and np.array_equal(np.isnan(ref["color"]), np.isnan(low["color"]))
    assert_image_close(low["color"], ref["color"], low["rgb"], ref["rgb"])
    assert (full["color"].reshape(-1, 3).max(axis=1) == 0).mean() > 0.05       # open scene: part of the image is background, exactly black
    print("configs[3]: NaN pixels", int(nan.sum()), "of", nan.size)
    ds.close()
    # ---- configs[4], rank 3 of 8
    aa = 100
    s0, s1 = dist.sample_range(3, 8, aa * aa)
    assert s1 - s0 == 1250
    ds = pt.DeviceScene(scenes.host_scene("cornell_buster"), 0, 32 << 30)
    acc.zero_()
    a = ds.render_accumulate(pt.render_params(w, h, aa, nb, s0, s1), acc.data_ptr(), stream, check_overflow=True, want_stats=True)
    assert a["samples"] == w * h * 1250 == a["per_depth_rays"][0] == a["per_depth_hits"][0]     # closed room: every primary ray hits
    ds.resolve(acc.data_ptr(), w, h, 1250, color.data_ptr(), rgb.data_ptr(), stream)
    torch.cuda.synchronize()
    share = color.cpu().numpy().reshape(h, w, 3)
    assert np.isfinite(share).all() and (share.reshape(-1, 3).max(axis=1) > 0).all()
    # the rank's share is already the image to a few 1e-3: the 1080p image at 1024 spp, every second pixel of it, is the same view
    lo = ds.render_image(1920, 1080, 32, nb)["color"]
    assert float(np.abs(share[::2, ::2] - lo).mean()) < 5e-3
    # two windows of the rank's share against the oracle's samples s0 .. s1 of the same pixels (glass box + mesh, wall)
    so = scenes.oracle_scene("cornell_buster")
    for x0, y0 in ((2360, 840), (600, 1600)):
        assert_window_equals_oracle_samples(port, so, share, w, h, aa, nb, x0, y0, s0, s1)
    ds.close()


def assert_window_equals_oracle_samples(port, s, color, w, h, aa, nb, x0, y0, s0, s1, nx=4, ny=3):
    """color[y0:y0+ny, x0:x0+nx] (the tone-mapped average of samples s0 .. s1 of every pixel) against the oracle's
    colours of those very samples, to the image bar.  Two references: the exact average of the samples (float64),
    and the average the reference's own loop forms - one fp32 accumulator per pixel, samples added in grid order
    (renderer.cpp:74-86) - which at 10^4 samples carries the rounding of 10^4 additions; the device's sums (wave
    reductions, then float atomics) must meet the bar against the exact average and stay within the reference's own
    rounding of the reference's."""
    sc = port.sample_colors(s, w, h, aa, nb, x0, y0, nx, ny)[:, :, s0:s1]      # (ny, nx, samples, 3), un-tone-mapped
    n = s1 - s0
    assert np.isfinite(sc).all()
    mean = sc.astype(np.float64).sum(axis=2) / n
    exact = (mean / (mean + 1)).astype(f32)                                    # tone_map_color, renderer.cpp:286-291
    seq = np.cumsum(sc, axis=2, dtype=f32)[:, :, -1] * (f32(1.0) / f32(n))     # traced_color += ...; *= 1.0f / n
    seq = (seq / (seq + f32(1.0))).astype(f32)
    got = color[y0:y0 + ny, x0:x0 + nx]
    rmse, mx = float(np.sqrt(np.mean((got.astype(np.float64) - exact) ** 2))), float(np.abs(got - exact).max())
    ref_own = float(np.abs(seq.astype(np.float64) - exact).max())              # what the reference's summation order costs it
    print(f"window ({x0},{y0}) {n} samples: device vs exact average rmse {rmse:.2e} max {mx:.2e}; reference's own sum vs exact max {ref_own:.2e};"
          f" device vs reference's sum max {float(np.abs(got - seq).max()):.2e}")
    assert rmse <= 1e-6 and mx <= 1e-5, (rmse, mx)
    assert float(np.abs(got - seq).max()) <= 1e-5 + ref_own


def test_ten_thousand_samples_per_pixel_against_the_oracle(pt, port, scenes):
    """The target of the north star: cornell + buster_sword at 1920x1080 with aa_steps = 100 (10 000 samples per
    pixel, 20.7 G samples), rendered in full; three windows (glass box in front of the mesh, a wall, the glass
    sphere) against the oracle's 10 000 samples of each of their pixels.  What is pinned here is the fp32
    accumulation: every pixel receives some 10^5 float additions in an order the reference does not have."""
    w, h, aa, nb = 1920, 1080, 100, 5
    ds = pt.DeviceScene(scenes.host_scene("cornell_buster"), 0, 32 << 30)
    a = ds.render_image(w, h, aa, nb)
    st = a["stats"]
    assert st["samples"] == w * h * aa * aa == st["per_depth_rays"][0] == st["per_depth_hits"][0]
    assert np.isfinite(a["color"]).all() and (a["color"].reshape(-1, 3).max(axis=1) > 0).all()
    names = scenes.host_scene("cornell_buster").prim_names()
    prim, _ = ds.primary_hits(w, h, 1, 0)
    prim = prim.reshape(h, w)
    s = scenes.oracle_scene("cornell_buster")
    seen = set()
    for x0, y0 in ((1180, 420), (300, 800), (760, 150)):
        seen |= {names[i] for i in np.unique(prim[y0:y0 + 3, x0:x0 + 4]) if i >= 0}
        assert_window_equals_oracle_samples(port, s, a["color"], w, h, aa, nb, x0, y0, 0, aa * aa)
    assert {"box1", "ball1"} <= seen and len(seen) >= 3, seen
    ds.close()


def test_moved_parts_of_loaded_meshes(pt, port):
    """Transformed Composites inside transformed Composites: nested.sdf loads two meshes (each a transformed
    Composite of sub-objects), and then parts of them are moved the way a C++ caller of the reference does it,
    root->find_child("sword")->find_child("Blade_Cube")->rotate(...) (tests/nested_ops.py).  Composite::intersect
    recurses with one bounds test and one change of coordinates per level (composite.cpp:75-103); the parent's
    bounds are NOT rebuilt when a child moves, so a moved part is cut off where it leaves them.  The scene description
    carries the chain (PtxFrame::parent), the library walks it: hit identity exact, distance, position, normal and
    ray direction bit-identical to the oracle (which tests/test_oracle.py pins to the real reference on this very
    scene) on primary and random rays; images at the usual bar, in both pipelines (such a scene always takes the
    staged kernels) and in the smallest work memory."""
    from nested_ops import apply_ops
    from oracle_lib import random_rays
    path = pt.SDF_DIR + "/nested.sdf"
    hs = pt.HostScene(path)
    assert all(hs.desc.contents.frames[f].parent == 0 for f in range(1, hs.desc.contents.n_frames))   # as loaded: nothing nested
    apply_ops(hs, None)
    d = hs.desc.contents
    parents = [d.frames[f].parent for f in range(d.n_frames)]
    assert sum(1 for p in parents if p > 0) == 4 and d.n_guards == 0
    s = port.load(path)
    apply_ops(port, s)
    assert hs.prim_names() == port.leaf_names(s)
    ds = pt.DeviceScene(hs, 0, 1 << 28)
    o1, d1 = port.primary_rays(s, 240, 160, 1)
    o2, d2 = random_rays(60000, 11, lo=(-8, -1, -20), hi=(8, 10, 4))
    o, dd = np.concatenate([o1, o2]), np.concatenate([d1, d2])
    a, b = ds.intersect(o, dd), port.intersect(s, o, dd)
    assert np.array_equal(a["prim"], b["rank"])
    m = b["hit"]
    names = np.array(hs.prim_paths())[b["rank"][m]]
    for part in ("sword/Blade_Cube/", "sword/Guard_Cube.002/", "superhot/left_arm", "superhot/torso", "sword/Grip", "superhot/right_leg"):
        assert sum(n.startswith(part) for n in names) > 100, part        # moved parts and parts left alone are all hit
    for k in ("t", "pos", "normal", "raydir"):
        assert beq(a[k][m], b[k][m]), k
    ref = port.render(s, 160, 100, 2, 4)
    for pipeline in PIPELINES:
        r = ds.render_image(160, 100, 2, 4, pipeline=pipeline)
        assert_image_close(r["color"], ref["color"], r["rgb"], ref["rgb"])
    ds.close()
    small = pt.DeviceScene(hs, 0, 96 << 20)
    r = small.render_image(160, 100, 2, 4)
    assert_image_close(r["color"], ref["color"], r["rgb"], ref["rgb"])
    small.close()
    port.free(s)


def test_shapes_grouped_and_moved_together(pt, port):
    """Leaves that are not plain triangles inside transformed Composites: on top of the moved mesh parts of
    test_moved_parts_of_loaded_meshes, a Composite of the caller's own with two spheres and two boxes is added to the
    root and moved as a whole (translate, rotate, uneven scale), its members are moved once more inside it, and two
    faces of the sword's blade - itself a moved part of a moved mesh - are moved on their own, and so are faces of
    parts that were not moved, one of them inside an octant of a part with more than 64 faces (tests/nested_ops.py;
    an untransformed Composite keeps the bounds it had, and a face that leaves them is cut off there: such a
    Composite is handed over as a frame of its own).
    Every such leaf is reached through the chain of frames above it and its own inverse, and reports position,
    normal and ray direction as Sphere / Box / Triangle::intersect and the Composites above would: hit identity exact,
    distance, position, normal, ray direction bit-identical to the oracle (pinned to the real reference on this scene
    by tests/test_oracle.py), images at the usual bar."""
    from nested_ops import apply_group
    from oracle_lib import random_rays
    path = pt.SDF_DIR + "/nested.sdf"
    hs = pt.HostScene(path)
    apply_group(hs, None)
    s = port.load(path)
    apply_group(port, s)
    assert hs.prim_names() == port.leaf_names(s)
    d = hs.desc.contents
    inside = [i for i in range(d.n_prims) if d.prims[i].frame > 0 and (d.prims[i].kind != 2 or any(d.prims[i].world[k] != (1.0 if k % 5 == 0 else 0.0) for k in range(16)))]
    assert len(inside) == 9 and {d.prims[i].kind for i in inside} == {0, 1, 2}
    assert all(d.prims[i].levels_in == 0 for i in inside)      # (an untransformed Composite that holds such a leaf is handed over as a frame of its own)
    ds = pt.DeviceScene(hs, 0, 1 << 28)
    o1, d1 = port.primary_rays(s, 960, 640, 1)      # (dense: the single moved faces are small)
    o2, d2 = random_rays(60000, 11, lo=(-8, -1, -20), hi=(8, 10, 4))
    o, dd = np.concatenate([o1, o2]), np.concatenate([d1, d2])
    a, b = ds.intersect(o, dd), port.intersect(s, o, dd)
    assert np.array_equal(a["prim"], b["rank"])
    m = b["hit"]
    paths = np.array(hs.prim_paths())[b["rank"][m]]
    for leaf in ("grp/gs1", "grp/gs2", "grp/gb1", "grp/gb2"):
        assert (paths == leaf).sum() > 30, leaf
    assert np.isin(paths, ["sword/Blade_Cube/face60", "sword/Blade_Cube/face61"]).sum() > 20
    for leaf in ("sword/Grip_Cube.004/face3", "sword/Pommel_Cube.003/face38", "superhot/right_leg_Superhot_Enemy_Stl.003/0/face249"):
        assert (paths == leaf).sum() >= 10, leaf
    for k in ("t", "pos", "normal", "raydir"):
        assert beq(a[k][m], b[k][m]), k
    ref = port.render(s, 160, 100, 2, 4)
    for pipeline in PIPELINES:
        r = ds.render_image(160, 100, 2, 4, pipeline=pipeline)
        assert_image_close(r["color"], ref["color"], r["rgb"], ref["rgb"])
    ds.close()
    port.free(s)


def test_moved_parts_below_a_split_root(pt, port):
    """Frames inside frames together with a split root: split_root.sdf (120 shapes and a mesh under a root that sorts them
    into octants), then parts of the mesh and single faces of it are moved, and a Composite of the caller's own with
    spheres and boxes is added beside the octants and moved (tests/nested_ops.py, SPLIT_OPS ...).  A ray reaches the mesh -
    and everything inside it - only through an octant it enters (composite.cpp:75-103), the hit's normal is re-normalised
    once more per octant level on the way up (:100), and among equal distances the child met first wins.  Hit identity
    exact; distance, position, normal, ray direction bit-identical to the oracle (pinned to the real reference on this
    very scene by tests/test_oracle.py); images at the usual bar."""
    from nested_ops import apply_split
    from oracle_lib import random_rays
    path = pt.SDF_DIR + "/split_root.sdf"
    hs = pt.HostScene(path)
    apply_split(hs, None)
    s = port.load(path)
    apply_split(port, s)
    assert hs.prim_names() == port.leaf_names(s)
    d = hs.desc.contents
    assert d.n_guards >= 8 and sorted(d.frames[f].parent for f in range(d.n_frames)) == [-1, 0, 0, 1, 1, 1]
    ds = pt.DeviceScene(hs, 0, 1 << 28)
    o1, d1 = port.primary_rays(s, 720, 480, 1)
    o2, d2 = random_rays(60000, 11, lo=(-12, -8, -32), hi=(12, 6, 4))
    o, dd = np.concatenate([o1, o2]), np.concatenate([d1, d2])
    a, b = ds.intersect(o, dd), port.intersect(s, o, dd)
    assert np.array_equal(a["prim"], b["rank"])
    m = b["hit"]
    paths = np.array(hs.prim_paths())[b["rank"][m]]
    for part in ("4/1/sword/Blade_Cube/", "4/1/sword/Guard_Cube.002/", "4/1/sword/Grip_Cube.004/", "4/1/sword/Pommel_Cube.003/"):
        assert sum(n.startswith(part) for n in paths) > 50, part
    for leaf in ("grp/gs1", "grp/gs2", "grp/gb1", "grp/gb2"):
        assert (paths == leaf).sum() > 30, leaf
    for leaf in ("4/1/sword/Grip_Cube.004/face3", "4/1/sword/Blade_Cube/face60"):
        assert (paths == leaf).sum() >= 10, leaf
    for k in ("t", "pos", "normal", "raydir"):
        assert beq(a[k][m], b[k][m]), k
    ref = port.render(s, 160, 100, 2, 4)
    for pipeline in PIPELINES:
        r = ds.render_image(160, 100, 2, 4, pipeline=pipeline)
        assert_image_close(r["color"], ref["color"], r["rgb"], ref["rgb"])
    ds.close()
    port.free(s)


def test_twelve_composites_one_inside_the_other(pt, port):
    """A chain of frames deeper than the octree of any loaded mesh gives (a face moved at the bottom of the big mesh's
    octree sits seven frames deep): twelve Composites of the caller's own one inside the other under the root of
    nested.sdf, each turned and shifted, a sphere or a box in each (tests/nested_ops.py, apply_deep; the reference and
    the oracle agree on this scene bit for bit, tests/test_oracle.py).  Every level is hit, every hit is the oracle's:
    identity exact, distance, position, normal and ray direction bit-identical; images at the usual bar."""
    from nested_ops import DEEP_LEVELS, apply_deep
    from oracle_lib import random_rays
    path = pt.SDF_DIR + "/nested.sdf"
    hs = pt.HostScene(path)
    apply_deep(hs, None)
    s = port.load(path)
    apply_deep(port, s)
    assert hs.prim_names() == port.leaf_names(s)
    d = hs.desc.contents
    depth = [0] * d.n_frames
    for f in range(1, d.n_frames):
        depth[f] = depth[d.frames[f].parent] + 1
    assert max(depth) == DEEP_LEVELS
    ds = pt.DeviceScene(hs, 0, 1 << 28)
    o1, d1 = port.primary_rays(s, 480, 320, 1)
    o2, d2 = random_rays(40000, 7, lo=(-8, -1, -20), hi=(8, 10, 4))
    o, dd = np.concatenate([o1, o2]), np.concatenate([d1, d2])
    a, b = ds.intersect(o, dd), port.intersect(s, o, dd)
    assert np.array_equal(a["prim"], b["rank"])
    m = b["hit"]
    names = np.array(hs.prim_names())[b["rank"][m]]
    for k in range(DEEP_LEVELS):
        assert (names == ("ds%d" if k % 2 == 0 else "db%d") % k).sum() > 50, k
    for k in ("t", "pos", "normal", "raydir"):
        assert beq(a[k][m], b[k][m]), k
    ref = port.render(s, 160, 100, 2, 4)
    for pipeline in PIPELINES:
        r = ds.render_image(160, 100, 2, 4, pipeline=pipeline)
        assert_image_close(r["color"], ref["color"], r["rgb"], ref["rgb"])
    ds.close()
    port.free(s)
