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
