"""What a C++ caller of the reference does to nested.sdf after load_scene: parts of the loaded meshes moved through
find_child(...)->find_child(...)->translate / rotate / scale (framework/composite.cpp:112-114, :168-181).  The same list is
applied to the reference's own Scene, to the oracle's and to the host mirror's."""
NESTED_OPS = [
    ("sword/Blade_Cube", "rotate", 0.0, 0.0, 0.12),
    ("sword/Blade_Cube", "translate", 0.15, 0.3, 0.0),
    ("sword/Guard_Cube.002", "scale", 1.5, 1.0, 2.0),
    ("superhot/left_arm_Superhot_Enemy_Stl.005", "translate", 0.0, 0.4, 0.1),
    ("superhot/left_arm_Superhot_Enemy_Stl.005", "rotate", 0.25, 0.0, 0.0),
    ("superhot/torso_Superhot_Enemy_Stl.002", "scale", 1.15, 1.0, 1.3),
]


def apply_ops(impl, scene, ops=NESTED_OPS):
    """impl: oracle_lib.Port / Reference (scene = their handle) or a path-tracer_amd HostScene (scene = None)"""
    for path, op, a, b, c in ops:
        if scene is None:
            impl.transform_child(path, op, a, b, c)
        else:
            impl.transform_child(scene, path, op, a, b, c)


# A Composite of the caller's own (spheres and boxes in it) added to the root and moved as a whole, members of it moved
# once more inside it, and single faces of a mesh part that was itself moved: leaves that are not plain triangles inside
# transformed Composites.
GROUP = ("grp", [("sphere", "gs1", (0.0, 0.0, 0.0, 0.8), "glass"), ("box", "gb1", (-0.5, -0.5, -0.5, 0.5, 0.5, 0.5), "red"),
                 ("sphere", "gs2", (1.5, 0.3, 0.0, 0.5), "mirror"), ("box", "gb2", (1.0, -0.8, -0.4, 2.2, -0.3, 0.4), "white")])
GROUP_OPS = [
    ("grp/gs2", "translate", 0.0, 0.5, 0.2),
    ("grp/gb1", "rotate", 0.5, 0.3, 0.0),
    ("grp/gb2", "scale", 1.0, 1.5, 1.0),
    ("grp", "translate", 0.5, 4.5, -10.0),
    ("grp", "rotate", 0.4, 0.2, 0.1),
    ("grp", "scale", 1.2, 0.9, 1.1),
    ("sword/Blade_Cube/face60", "translate", 0.3, 0.0, 0.0),
    ("sword/Blade_Cube/face61", "rotate", 0.0, 0.0, 0.3),
    # faces of parts that were NOT moved: the part (an untransformed Composite inside the moved mesh) keeps the bounds it
    # had, and a face that leaves them is cut off there; and a face inside an octant of a part with more than 64 faces
    ("sword/Grip_Cube.004/face3", "translate", 0.0, 0.0, 0.25),
    ("sword/Pommel_Cube.003/face38", "translate", 0.0, 0.15, 0.3),
    ("superhot/right_leg_Superhot_Enemy_Stl.003/0/face249", "translate", 0.1, 0.0, 0.1),
]


def apply_group(impl, scene):
    """NESTED_OPS, then the group and GROUP_OPS"""
    apply_ops(impl, scene)
    if scene is None:
        impl.add_group(*GROUP)
    else:
        impl.add_group(scene, *GROUP)
    apply_ops(impl, scene, GROUP_OPS)


# The same kinds of moves on split_root.sdf, whose root has more than 64 children and sorts them - the mesh among them -
# into octants: frames inside frames below a child that rays reach only through an octant they enter.  A caller finds the
# mesh through the octants (root->find_child("4")->find_child("1")->find_child("sword"): find_child looks at one level,
# composite.cpp:112-114); the other octants the mesh lies in hold the same object.
SPLIT_OPS = [
    ("4/1/sword/Blade_Cube", "rotate", 0.0, 0.0, 0.12),
    ("4/1/sword/Blade_Cube", "translate", 0.15, 0.3, 0.0),
    ("4/1/sword/Guard_Cube.002", "scale", 1.5, 1.0, 2.0),
    ("4/1/sword/Grip_Cube.004/face3", "translate", 0.0, 0.0, 0.25),
    ("4/1/sword/Blade_Cube/face60", "rotate", 0.0, 0.0, 0.3),
]
SPLIT_GROUP = ("grp", [("sphere", "gs1", (0.0, 0.0, 0.0, 0.8), "g"), ("box", "gb1", (-0.5, -0.5, -0.5, 0.5, 0.5, 0.5), "a"),
                       ("sphere", "gs2", (1.5, 0.3, 0.0, 0.5), "m"), ("box", "gb2", (1.0, -0.8, -0.4, 2.2, -0.3, 0.4), "d")])
SPLIT_GROUP_OPS = [
    ("grp/gs2", "translate", 0.0, 0.5, 0.2),
    ("grp/gb1", "rotate", 0.5, 0.3, 0.0),
    ("grp", "translate", -2.0, 1.0, -10.0),
    ("grp", "rotate", 0.4, 0.2, 0.1),
    ("grp", "scale", 1.2, 0.9, 1.1),
]


def apply_split(impl, scene):
    """SPLIT_OPS, then the group and SPLIT_GROUP_OPS (on split_root.sdf)"""
    apply_ops(impl, scene, SPLIT_OPS)
    if scene is None:
        impl.add_group(*SPLIT_GROUP)
    else:
        impl.add_group(scene, *SPLIT_GROUP)
    apply_ops(impl, scene, SPLIT_GROUP_OPS)


# Twelve Composites of the caller's own one inside the other under the root of nested.sdf, a sphere or a box in each,
# every one of them turned and shifted a little: a chain of frames deeper than any octree of a loaded mesh gives.
DEEP_LEVELS = 12


def apply_deep(impl, scene):
    """NESTED_OPS, then the chain t0/t1/.../t11, every level moved as soon as it is there (the Composites above get
    their bounds from what they hold when the next one is added; the moves after the last one do not reach them)"""
    apply_ops(impl, scene)
    args = () if scene is None else (scene,)
    path = ""
    for k in range(DEEP_LEVELS):
        path = f"{path}/t{k}" if path else "t0"
        leaf = ("sphere", f"ds{k}", (0.35 * k, 0.0, 0.0, 0.3), ("glass", "red", "mirror", "white")[k % 4]) if k % 2 == 0 else \
               ("box", f"db{k}", (0.35 * k - 0.25, -0.25, -0.25, 0.35 * k + 0.25, 0.25, 0.25), ("glass", "red", "mirror", "white")[k % 4])
        impl.add_group(*args, path, [leaf])
        impl.transform_child(*args, path, "rotate", 0.0, 0.0, 0.12)
        impl.transform_child(*args, path, "translate", 0.05, 0.1, 0.0)
    impl.transform_child(*args, "t0", "translate", -3.0, 1.0, -9.0)
    impl.transform_child(*args, "t0", "scale", 1.2, 1.0, 1.1)
