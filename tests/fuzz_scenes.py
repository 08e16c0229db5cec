"""Generated scenes for the parity tests: every seed gives one scene file (SDF text) with a random
mix of spheres, boxes and triangles directly under the root - few (the leaf list), some dozens (the
BVH over the frame-0 leaves) or more than 64 (a root the reference splits into octants) - half of
them translated, rotated (also by exact quarter turns, whose matrices carry exact and negative
zeros) and scaled (also unevenly), sometimes a mesh from the fixtures, an enclosing room or not,
none to five lights, materials of every shading case (Phong only, mirror, glass, both).
The same text goes to the reference build, to the CPU oracle and to the product.  Some scenes are changed after
loading the way a C++ caller changes them (scene_moves: parts of a mesh moved, a Composite of the caller's own)."""
import numpy as np

f32 = np.float32
N_SHAPES = (3, 9, 16, 17, 30, 64, 65, 90, 130)
MESHES = ("sword", "superhot")


def scene_scale(seed):
    """two scenes in seven are a hundred or ten thousand times larger, or a hundred or a thousand times smaller - down
    to where the reference's absolute EPSILON of 0.001 is the size of the shapes (the conservative filters in front
    of the exact tests - padded boxes, fused slab tests - must hold at any size)"""
    return {5: 100.0, 6: 0.01, 12: 1.0e4, 13: 1.0e-3}.get(seed % 14, 1.0)


def scene_text(seed):
    rng = np.random.default_rng(1000 + seed)
    S = scene_scale(seed)

    def L(*vals):       # lengths and positions, in the scene's scale
        return " ".join(f"{v * S:.3f}" if S == 1.0 else f"{v * S:.7g}" for v in vals)
    lines = []
    mats = []
    for k in range(6):
        ka, kd, ks = rng.uniform(0, 1, 3), rng.uniform(0, 1, 3), rng.uniform(0, 1, 3)
        m = [0, 1, 8, 50, 500, 2.5][k] if seed % 2 else [20, 0, 3, 100, 7.25, 1][k]
        if seed % 3 == 0:
            m = int(m)      # whole-number exponents only: the kernels' square-and-multiply power instead of pow()
        glossy = [0, 0.3, 1, 0, 0.5, 0][k]
        opacity = [1, 1, 1, 0.4, 0.2, 1][k]
        ior = [1, 1, 1, 1.3, 1.5, 1][k]
        name = f"m{k}"
        mats.append(name)
        lines.append("define material %s %s %g %g %g %g" % (name, " ".join(f"{v:.3f}" for v in (*ka, *kd, *ks)), m, glossy, opacity, ior))
    n = N_SHAPES[seed % len(N_SHAPES)]
    tf = []
    for k in range(n):
        x, y, z = rng.uniform(-8, 8), rng.uniform(-4, 4), rng.uniform(-25, -6)
        mat = mats[int(rng.integers(0, len(mats)))]
        kind = int(rng.integers(0, 3))
        name = "sbt"[kind] + f"{k:03d}"
        local = rng.random() < 0.5      # defined around the origin and moved by its transform, or defined in place
        cx, cy, cz = (0.0, 0.0, 0.0) if local else (x, y, z)
        if kind == 0:
            lines.append(f"define shape sphere {name} {L(cx, cy, cz, rng.uniform(0.3, 1.4))} {mat}")
        elif kind == 1:
            e = rng.uniform(0.3, 1.6, 3)
            lines.append(f"define shape box {name} {L(cx - e[0], cy - e[1], cz - e[2], cx + e[0], cy + e[1], cz + e[2])} {mat}")
        else:
            p = rng.uniform(-1.5, 1.5, 9)
            lines.append(f"define shape triangle {name} " + L(p[0] + cx, p[1] + cy, p[2] + cz, p[3] + cx, p[4] + cy, p[5] + cz, p[6] + cx, p[7] + cy, p[8] + cz) + f" {mat}")
        if local:
            tf.append(f"transform {name} translate {L(x, y, z)}")
        r = rng.random()
        if r < 0.25:
            tf.append(f"transform {name} rotate {rng.uniform(-180, 180):.1f} {rng.uniform(-90, 90):.1f} {rng.uniform(-180, 180):.1f}")
        elif r < 0.35:
            q = rng.integers(-2, 3, 3) * 90
            tf.append(f"transform {name} rotate {q[0]} {q[1]} {q[2]}")
        r = rng.random()
        if r < 0.2:
            tf.append(f"transform {name} scale {rng.uniform(0.4, 1.8):.2f} {rng.uniform(0.4, 1.8):.2f} {rng.uniform(0.4, 1.8):.2f}")
        elif r < 0.3:
            s = rng.uniform(0.5, 1.5)
            tf.append(f"transform {name} scale {s:.2f} {s:.2f} {s:.2f}")
        if not local and rng.random() < 0.15:
            tf.append(f"transform {name} translate {L(rng.uniform(-2, 2), rng.uniform(-1, 1), rng.uniform(-2, 2))}")
    if seed % 5 == 0:
        # ties: boxes on a lattice that share whole faces, and exact copies of a box and a triangle under later names
        # (equal distances: the reference keeps the child it met first, composite.cpp:93)
        for k in range(4):
            x0 = -4 + 2 * k
            lines.append(f"define shape box lat{k} {L(x0, -1, -12, x0 + 2, 1, -10)} {mats[k % len(mats)]}")
        lines.append(f"define shape box twin_a {L(-7, 2, -15, -5, 3, -13)} {mats[1]}")
        lines.append(f"define shape box twin_b {L(-7, 2, -15, -5, 3, -13)} {mats[2]}")
        lines.append(f"define shape triangle twin_c {L(3, 2, -9, 6, 2, -9, 4, 4, -9)} {mats[3]}")
        lines.append(f"define shape triangle twin_d {L(3, 2, -9, 6, 2, -9, 4, 4, -9)} {mats[4]}")
    if seed % 3 != 2:
        lines.append(f"define shape box room {L(-11, -6, -30, 11, 6, 8)} {mats[0]}")
    if seed % 2 == 0:
        mesh = MESHES[(seed // 2) % len(MESHES)]
        lines.append(f"define shape obj {mesh}")
        tf.append(f"transform {mesh} translate {L(rng.uniform(-3, 3), rng.uniform(-4, -1), rng.uniform(-16, -9))}")
        if seed % 4 == 0:
            tf.append(f"transform {mesh} rotate {rng.uniform(-90, 90):.1f} {rng.uniform(-40, 40):.1f} 0")
        else:
            tf.append(f"transform {mesh} rotate 90 0 -90")
        s = rng.uniform(0.5, 1.1)
        tf.append(f"transform {mesh} scale {s * S:.7g} {s * S:.7g} {s * S * (1.0 if seed % 8 else 1.4):.7g}")
    # a second, large mesh with its own materials (16 008 faces, 11 sub-objects, deep octrees); not ten thousand times
    # larger, where NaN normals abound and every ray with a NaN direction visits every leaf in the reference: minutes
    # of CPU time for one small image
    if seed % 6 == 4 and S < 1.0e4:
        lines.append("define shape obj colored-buster")
        tf.append(f"transform colored-buster translate {L(rng.uniform(-4, 4), rng.uniform(-3, 0), rng.uniform(-20, -12))}")
        tf.append(f"transform colored-buster rotate {rng.uniform(-180, 180):.1f} {rng.uniform(-60, 60):.1f} {rng.uniform(-90, 90):.1f}")
        tf.append(f"transform colored-buster scale {.14 * S:.7g} {.14 * S:.7g} {.14 * S:.7g}")
    for k in range({7: 0, 9: 5}.get(seed % 11, 1 + seed % 3)):      # (one to three lights; now and then none, or five)
        lines.append(f"define light l{k} {L(rng.uniform(-8, 8), rng.uniform(2, 5.5), rng.uniform(-20, 4))} {rng.uniform(.3, 1):.2f} {rng.uniform(.3, 1):.2f} {rng.uniform(.3, 1):.2f} {rng.uniform(.5, 1.5):.2f}")
    lines.append("define ambient amb 1 1 1 0.15")
    if seed % 4 == 1:     # a camera that looks along no axis, rolled
        lines.append(f"define camera eye {rng.uniform(40, 90):.1f} {L(rng.uniform(-3, 3), rng.uniform(-2, 3), rng.uniform(3, 7))} "
                     f"{rng.uniform(-25, 25):.1f} {rng.uniform(-15, 15):.1f} {rng.uniform(-30, 30):.1f}")
    elif seed % 4 == 3:   # quarter turns: camera matrices with exact and negative zeros
        lines.append(f"define camera eye 60 {L(0, 0, 5)} 0 0 {(seed // 4 % 4) * 90}")
    else:
        lines.append(f"define camera eye 60 {L(0, 0, 5)} 0 0 0")
    return "\n".join(lines + tf) + "\n"


PARTS = {"sword": ("Grip_Cube.004", "Guard_Cube.002", "Pommel_Cube.003", "Blade_Cube"),
         "superhot": ("torso_Superhot_Enemy_Stl.002", "right_leg_Superhot_Enemy_Stl.003", "left_leg_Superhot_Enemy_Stl.004",
                      "left_arm_Superhot_Enemy_Stl.005", "right_arm_Superhot_Enemy_Stl.006", "head_Superhot_Enemy_Stl.007")}


def scene_moves(seed):
    """What a C++ caller does to some of the scenes after load_scene (None for the others): parts of the loaded mesh moved
    through find_child(..)->find_child(..)->translate / rotate / scale - transformed Composites inside a transformed
    Composite -, in some scenes a face of a moved part moved once more and a Composite of the caller's own with spheres
    and boxes in it, moved as a whole and member by member.  Returns (ops, group, group_ops): ops as
    (path, 'translate' | 'rotate' | 'scale', a, b, c), group as (name, [(kind, name, numbers, material)])."""
    if seed % 2 or not (seed % 4 == 0 or seed % 8 == 2):
        return None     # no mesh, or left as loaded
    rng = np.random.default_rng(5000 + seed)
    S = scene_scale(seed)
    mesh = MESHES[(seed // 2) % len(MESHES)]
    parts = list(PARTS[mesh])
    ops = []
    for k in rng.choice(len(parts), size=2 if mesh == "sword" else 3, replace=False):
        path = f"{mesh}/{parts[int(k)]}"
        for _ in range(int(rng.integers(1, 4))):
            op = ("translate", "rotate", "scale")[int(rng.integers(0, 3))]
            if op == "translate":
                ops.append((path, op, *(float(v) for v in rng.uniform(-0.6, 0.6, 3))))
            elif op == "rotate":
                ops.append((path, op, *(float(v) for v in rng.uniform(-0.7, 0.7, 3))))
            else:
                ops.append((path, op, *(float(v) for v in rng.uniform(0.6, 1.6, 3))))
    if mesh == "sword" and any(o[0].endswith("Blade_Cube") for o in ops):
        ops.append(("sword/Blade_Cube/face60", "translate", 0.2, 0.1, 0.0))
        ops.append(("sword/Blade_Cube/face55", "rotate", 0.0, 0.3, 0.2))
    if mesh == "sword":     # a face of a part that stays where it is (the part keeps its bounds: the face is cut off where it leaves them)
        first_face = {"Grip_Cube.004": 0, "Guard_Cube.002": 12, "Pommel_Cube.003": 32, "Blade_Cube": 52}
        still = [q for q in parts if not any(o[0] == f"sword/{q}" for o in ops)]
        q = still[int(rng.integers(0, len(still)))]
        ops.append((f"sword/{q}/face{first_face[q] + int(rng.integers(0, 12))}", "translate", *(float(v) for v in rng.uniform(-0.3, 0.3, 3))))
    group, group_ops = None, []
    if seed % 8 in (2, 4):
        kids = []
        for k in range(int(rng.integers(2, 5))):
            c = rng.uniform(-1.5, 1.5, 3) * S
            if rng.random() < 0.5:
                kids.append(("sphere", f"gs{k}", (float(c[0]), float(c[1]), float(c[2]), float(rng.uniform(0.3, 0.9) * S)), f"m{int(rng.integers(0, 6))}"))
            else:
                e = rng.uniform(0.2, 0.8, 3) * S
                kids.append(("box", f"gb{k}", tuple(float(v) for v in (*(c - e), *(c + e))), f"m{int(rng.integers(0, 6))}"))
        group = ("grp", kids)
        group_ops.append((f"grp/{kids[0][1]}", "translate", *(float(v) for v in rng.uniform(-0.5, 0.5, 3) * S)))
        group_ops.append((f"grp/{kids[-1][1]}", "rotate", *(float(v) for v in rng.uniform(-0.8, 0.8, 3))))
        group_ops.append(("grp", "translate", float(rng.uniform(-4, 4) * S), float(rng.uniform(-2, 2) * S), float(rng.uniform(-14, -7) * S)))
        group_ops.append(("grp", "rotate", *(float(v) for v in rng.uniform(-1.0, 1.0, 3))))
        group_ops.append(("grp", "scale", *(float(v) for v in rng.uniform(0.7, 1.5, 3))))
    return ops, group, group_ops


def mesh_prefix(impl, scene, seed):
    """Where a caller finds the mesh of scene `seed`: "" when it is a child of the root, the names of the octants above it
    ("4/1/") when the root has more than 64 children and was split (find_child looks at one level, composite.cpp:112-114;
    the first octant in visiting order that holds the mesh - the others hold the same object).  impl: oracle_lib.Port
    (scene = its handle) or a path-tracer_amd HostScene (scene = None)."""
    mesh = MESHES[(seed // 2) % len(MESHES)]
    for path in (impl.prim_paths() if scene is None else impl.leaf_paths(scene)):
        names = path.split("/")[:-1]
        if mesh in names:
            return "".join(n + "/" for n in names[:names.index(mesh)])
    return ""


def apply_moves(impl, scene, seed, prefix=None):
    """scene_moves(seed) applied through oracle_lib.Port / Reference (scene = their handle) or a path-tracer_amd HostScene
    (scene = None); returns whether the scene has any.  prefix: mesh_prefix() of the scene (looked up here when the
    implementation can list its leaves' paths; the Reference shim cannot and needs it passed in)"""
    moves = scene_moves(seed)
    if moves is None:
        return False
    ops, group, group_ops = moves
    args = () if scene is None else (scene,)
    if prefix is None:
        prefix = mesh_prefix(impl, scene, seed)
    for path, op, a, b, c in ops:
        impl.transform_child(*args, prefix + path, op, a, b, c)
    if group:
        impl.add_group(*args, *group)
    for path, op, a, b, c in group_ops:
        impl.transform_child(*args, path, op, a, b, c)
    return True


def rays(seed, n_random=12000, n_axis=3000):
    """Random rays through the region of the shapes, and rays with one or two direction components
    exactly zero (the branch of the box test for rays parallel to a slab, the divisions by zero
    everywhere else), some of them starting exactly on a lattice of round coordinates."""
    rng = np.random.default_rng(7000 + seed)
    o = rng.uniform((-10, -5, -28), (10, 5, 6), size=(n_random, 3)).astype(f32)
    d = rng.normal(size=(n_random, 3)).astype(f32)
    d *= (rng.uniform(0.5, 2.0, size=(n_random, 1)) / np.linalg.norm(d, axis=1, keepdims=True)).astype(f32)
    oa = rng.uniform((-10, -5, -28), (10, 5, 6), size=(n_axis, 3)).astype(f32)
    oa[::3] = np.round(oa[::3])
    da = rng.normal(size=(n_axis, 3)).astype(f32)
    zero = rng.integers(0, 3, n_axis)
    da[np.arange(n_axis), zero] = 0.0
    two = rng.random(n_axis) < 0.5
    da[np.arange(n_axis)[two], (zero[two] + 1) % 3] = 0.0
    da[::7] *= -1.0     # (negative zeros among the zero components)
    # rays whose direction is NaN (the children of a hit with a NaN normal): the reference answers them all alike
    on = rng.uniform((-10, -5, -28), (10, 5, 6), size=(16, 3)).astype(f32)
    dn = np.full((16, 3), np.nan, f32)
    S = f32(scene_scale(seed))
    return (np.concatenate([o, oa, on]) * S).astype(f32), np.concatenate([d, da, dn]).astype(f32)


def same_bits_or_both_nan(a, b):
    a, b = np.asarray(a, f32), np.asarray(b, f32)
    return bool(((a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b))).all())


def write_scene(seed, folder, fixtures_sdf):
    """The scene of `seed` as <folder>/sdf/gen<seed>.sdf with the fixture meshes linked beside it (the
    reference opens OBJ files at ../../sdf/ relative to a working directory below the folder that
    holds sdf/; the oracle and the product take the folder as an argument).  Returns (file, obj_dir)."""
    import os
    d = os.path.join(str(folder), "sdf")
    os.makedirs(d, exist_ok=True)
    for f in os.listdir(fixtures_sdf):
        if f.endswith((".obj", ".mtl")) and not os.path.exists(os.path.join(d, f)):
            os.symlink(os.path.join(fixtures_sdf, f), os.path.join(d, f))
    path = os.path.join(d, f"gen{seed}.sdf")
    with open(path, "w") as fh:
        fh.write(scene_text(seed))
    return path, d + "/"


def full_size_findings(pt, seed, folder, W=1920, H=1080, AA=4):
    """The scene of `seed` at a full image size: the staged pipeline in ample work memory against the
    fused one and against both in the smallest work memory - the same rays, shadow rays and hits per
    depth, the same image to the bar of the parity tests (pixels near the pole of the tone mapping
    compared before it, at most eight at the end of the float range).  No oracle: a consistency check
    of the queue driving under load.  Returns the findings as text lines."""
    sdf, obj_dir = write_scene(seed, folder, pt.SDF_DIR)
    hs = pt.HostScene(sdf, obj_dir)
    nested = apply_moves(hs, None, seed)
    nb = (4, 6, 8)[seed % 3]
    big, tiny = pt.DeviceScene(hs, 0, 1 << 32), pt.DeviceScene(hs, 0, 1)
    found = []
    try:
        ref = big.render_image(W, H, AA, nb, pipeline="staged")
        fin = np.isfinite(ref["color"]).all(axis=2)
        rb = ref["color"].astype(np.float64)
        for ds, name, pipeline in ((big, "ample", "fused"), (tiny, "smallest", "staged"), (tiny, "smallest", "fused")):
            if pipeline == "fused" and (hs.desc.contents.n_guards or nested):
                continue        # (a split root always takes the staged kernels, and so do frames inside frames)
            q = ds.render_image(W, H, AA, nb, pipeline=pipeline)
            same = all(q["stats"][k] == ref["stats"][k] for k in ("trace_rays", "shadow_rays", "shaded_hits", "per_depth_rays"))
            odd = np.isfinite(q["color"]).all(axis=2) != fin
            qa = q["color"].astype(np.float64)
            with np.errstate(divide="ignore", invalid="ignore"):
                close = (np.isclose(qa, rb, rtol=0, atol=1e-5).all(axis=2) | ~fin | odd
                         | np.isclose(qa / (1 - qa), rb / (1 - rb), rtol=1e-5, atol=0).all(axis=2))
            if not (same and odd.sum() <= 8 and close.mean() >= 0.997):
                found.append(f"seed {seed}, {name} work memory, {pipeline}: counts equal {same}, finite masks differ in {odd.sum()} pixels, "
                             f"{close.mean():.5f} of the pixels agree")
    finally:
        big.close()
        tiny.close()
    return found
