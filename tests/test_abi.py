"""The C-ABI library loads and exports every symbol include/ptx.h declares; without a GPU
the device entry points fail loudly instead of falling back to anything."""
import ctypes
import os
import re

import pytest

from conftest import REPO


def declared_symbols():
    text = open(os.path.join(REPO, "include", "ptx.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ptx_[a-z_0-9]+)\s*\(", text)))


def test_header_and_library_agree(pt):
    L = pt.ptx_lib()
    names = declared_symbols()
    assert sorted(pt.PTX_SYMBOLS) == names
    for n in names:
        assert hasattr(L, n), f"libptx.so does not export {n}"
    assert b"gfx950" in L.ptx_version()


def test_library_holds_gfx950_code_only(pt):
    so = os.path.join(REPO, "path-tracer_amd", "libptx.so")
    blob = open(so, "rb").read()
    assert b"amdgcn-amd-amdhsa--gfx950" in blob
    for other in (b"gfx90a", b"gfx942", b"sm_90"):
        assert other not in blob


def test_struct_sizes_match_the_header(pt, tmp_path):
    """The ctypes mirrors have the sizes a C compiler gives the structs of include/ptx.h (the
    header is compiled as plain C here, which also shows that it is one)."""
    import subprocess
    names = ["PtxMaterial", "PtxLight", "PtxCamera", "PtxFrame", "PtxPrim", "PtxGuard", "PtxOccurrence", "PtxSceneDesc", "PtxRenderParams", "PtxStats"]
    src = tmp_path / "sizes.c"
    src.write_text('#include <stdio.h>\n#include "ptx.h"\nint main(void) {\n' +
                   "".join(f'printf("{n} %zu\\n", sizeof({n}));\n' for n in names) + "return 0; }\n")
    exe = tmp_path / "sizes"
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(REPO, "include"), str(src), "-o", str(exe)], check=True)
    out = dict(line.split() for line in subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.splitlines())
    for n in names:
        assert ctypes.sizeof(getattr(pt, n)) == int(out[n]), n
    assert ctypes.sizeof(pt.PtxMaterial) == 13 * 4 and ctypes.sizeof(pt.PtxPrim) == (4 + 9 + 3 + 32) * 4


def test_no_device_means_loud_failure(pt, scenes):
    if pt.device_count() > 0:
        pytest.skip("a GPU is present: the failure path is for boxes without one")
    with pytest.raises(pt.PtxError) as e:
        pt.DeviceScene(scenes.host_scene("cornell"))
    assert e.value.status == 3 and "no CPU path" in str(e.value)


def test_bad_arguments_are_rejected_without_a_device(pt):
    L = pt.ptx_lib()
    out = ctypes.c_void_p()
    assert L.ptx_scene_create(None, 0, 0, ctypes.byref(out)) == 1
    assert b"NULL" in L.ptx_last_error()
    assert L.ptx_write_ppm(None, None, 1, 1) == 1
    assert L.ptx_scene_set_profiling(None, 1) == 1


def test_malformed_octant_lists_are_rejected_without_a_device(pt):
    """The octant lists of a split root (PtxGuard / PtxOccurrence) are checked before anything
    touches a device: an octant must come after the octant above it, its level count must follow
    from that one, and an occurrence must name an existing octant and a frame-0 leaf or a mesh."""
    L = pt.ptx_lib()
    hs = pt.HostScene(pt.SDF_DIR + "/split_root.sdf")
    good = hs.desc.contents
    assert good.n_guards > 0

    def attempt(change):
        guards = (pt.PtxGuard * good.n_guards)(*[good.guards[i] for i in range(good.n_guards)])
        occ = (pt.PtxOccurrence * good.n_occurrences)(*[good.occurrences[i] for i in range(good.n_occurrences)])
        d = type(good).from_buffer_copy(good)
        d.guards, d.occurrences = guards, occ
        change(d, guards, occ)
        out = ctypes.c_void_p()
        st = L.ptx_scene_create(ctypes.byref(d), 0, 0, ctypes.byref(out))
        return st, L.ptx_last_error().decode()

    def parent_after(d, g, o):
        g[0].parent = 3
    def wrong_levels(d, g, o):
        g[1].levels += 1
    def no_such_octant(d, g, o):
        o[5].guard = d.n_guards
    def neither_leaf_nor_mesh(d, g, o):
        o[2].prim, o[2].frame = -1, 0
    def missing_list(d, g, o):
        d.occurrences = None
    for change, word in ((parent_after, "guard 0"), (wrong_levels, "guard 1"), (no_such_octant, "occurrence 5"), (neither_leaf_nor_mesh, "occurrence 2"),
                         (missing_list, "NULL")):
        st, msg = attempt(change)
        assert st == 1 and word in msg, (change.__name__, st, msg)
    st, msg = attempt(lambda d, g, o: None)           # the untouched description gets as far as the device
    assert st == (0 if pt.device_count() > 0 else 3), msg
