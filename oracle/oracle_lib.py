"""ctypes bindings for the CPU oracle (TEST INFRASTRUCTURE, never imported by the product).

Two libraries can sit behind the same Python interface:

* ``libcpu_ref.so``  - this repository's CPU restatement (oracle/cpu_ref.cpp), kind "port";
* ``_ref/libref.so`` - the unmodified reference sources compiled with oracle/ref_shim.cpp,
  kind "reference".  It only exists where oracle/Makefile found the reference checkout
  at build time (the binary travels to the GPU box, the sources do not).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module.
"""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
SDF_DIR = os.path.join(REPO, "fixtures", "sdf")

f32 = np.float32


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


OPS = {"translate": 1, "rotate": 2, "scale": 3}


def group_arrays(children):
    """the arrays the three add_group entry points take (oracle, reference shim, host mirror)"""
    kinds = np.array([0 if c[0] == "sphere" else 1 for c in children], np.int32)
    params = np.zeros((len(children), 7), np.float32)
    for i, c in enumerate(children):
        params[i, :len(c[2])] = c[2]
    return kinds, params, ";".join(c[1] for c in children).encode(), ";".join(c[3] for c in children).encode()


def _add_group(fn, s, name, children):
    kinds, params, names, mats = group_arrays(children)
    fn.argtypes = [C.c_void_p, C.c_char_p, C.c_int, C.c_void_p, C.c_void_p, C.c_char_p, C.c_char_p]
    if fn(s, name.encode(), len(children), _p(kinds), _p(params), names, mats) != 0:
        raise RuntimeError("add_group failed")


class Port:
    """oracle/cpu_ref.cpp"""

    kind = "port"

    def __init__(self, path=None):
        path = path or os.path.join(HERE, "libcpu_ref.so")
        if not os.path.exists(path):
            raise FileNotFoundError(f"{path} missing: run `make -C oracle` (or __graft_entry__.build())")
        L = self.L = C.CDLL(path)
        L.oracle_load_scene.restype = C.c_void_p
        L.oracle_load_scene.argtypes = [C.c_char_p, C.c_char_p]
        L.oracle_free_scene.argtypes = [C.c_void_p]
        L.oracle_scene_error.restype = C.c_char_p
        L.oracle_scene_error.argtypes = [C.c_void_p]
        L.oracle_leaf_count.argtypes = [C.c_void_p]
        L.oracle_leaf_name.restype = C.c_char_p
        L.oracle_leaf_name.argtypes = [C.c_void_p, C.c_int]
        L.oracle_leaf_path.restype = C.c_char_p
        L.oracle_leaf_path.argtypes = [C.c_void_p, C.c_int]
        L.oracle_render.restype = C.c_double
        L.oracle_render.argtypes = [C.c_void_p, C.c_uint, C.c_uint, C.c_uint, C.c_uint, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.oracle_render_jittered.restype = C.c_double
        L.oracle_render_jittered.argtypes = L.oracle_render.argtypes + [C.c_uint32]
        L.oracle_sample_colors.argtypes = [C.c_void_p] + [C.c_uint] * 8 + [C.c_void_p]
        L.oracle_ppm_text.restype = C.c_long
        L.oracle_ppm_text.argtypes = [C.c_void_p, C.c_uint, C.c_uint, C.c_void_p, C.c_long]
        L.oracle_primary_rays.argtypes = [C.c_void_p, C.c_uint, C.c_uint, C.c_uint, C.c_void_p, C.c_void_p]
        L.oracle_intersect.argtypes = [C.c_void_p, C.c_long] + [C.c_void_p] * 7
        L.oracle_prim_intersect.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_long] + [C.c_void_p] * 7
        L.oracle_box_bounds.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.oracle_scene_info.argtypes = [C.c_void_p] + [C.c_void_p] * 6

    def load(self, sdf_path, obj_dir=None):
        obj_dir = obj_dir or (os.path.dirname(os.path.abspath(sdf_path)) + "/")
        s = self.L.oracle_load_scene(sdf_path.encode(), obj_dir.encode())
        err = self.L.oracle_scene_error(s).decode()
        if err:
            raise RuntimeError(err)
        return s

    def free(self, s):
        self.L.oracle_free_scene(s)

    def add_group(self, s, name, children):
        """a Composite of the caller's own under the root; children: (kind 'sphere' | 'box', name, 4 or 6 floats, material name)"""
        _add_group(self.L.oracle_add_group, s, name, children)

    def transform_child(self, s, path, op, a, b, c):
        """root->find_child(..)->...->translate / rotate / scale; op 'translate' | 'rotate' (radians) | 'scale'"""
        self.L.oracle_transform_child.argtypes = [C.c_void_p, C.c_char_p, C.c_int, C.c_float, C.c_float, C.c_float]
        if self.L.oracle_transform_child(s, path.encode(), OPS[op], a, b, c) != 0:
            raise RuntimeError("no child " + path)

    def leaf_names(self, s):
        n = self.L.oracle_leaf_count(s)
        return [self.L.oracle_leaf_name(s, i).decode() for i in range(n)]

    def leaf_paths(self, s):
        n = self.L.oracle_leaf_count(s)
        return [self.L.oracle_leaf_path(s, i).decode() for i in range(n)]

    def render(self, s, w, h, aa, bounces, threads=0, counters=False, jitter_seed=0):
        cb = np.zeros((h, w, 3), f32)
        rgb = np.zeros((h, w, 3), np.uint8)
        cnt = np.zeros(7, np.uint64) if counters else None
        if jitter_seed:  # the product's optional jittered sampling (nothing in the reference corresponds to it)
            sec = self.L.oracle_render_jittered(s, w, h, aa, bounces, threads, _p(cb), _p(rgb), _p(cnt), jitter_seed)
        else:
            sec = self.L.oracle_render(s, w, h, aa, bounces, threads, _p(cb), _p(rgb), _p(cnt))
        return dict(color=cb, rgb=rgb, seconds=sec, counters=cnt)

    def sample_colors(self, s, w, h, aa, bounces, x0, y0, nx, ny):
        out = np.zeros((ny, nx, aa * aa, 3), f32)
        self.L.oracle_sample_colors(s, w, h, aa, bounces, x0, y0, nx, ny, _p(out))
        return out

    def ppm_text(self, rgb):
        h, w, _ = rgb.shape
        rgb = np.ascontiguousarray(rgb, np.uint8)
        cap = w * h * 3 * 4 + 64
        buf = np.zeros(cap, np.uint8)
        n = self.L.oracle_ppm_text(_p(rgb), w, h, _p(buf), cap)
        return buf[:n].tobytes()

    def primary_rays(self, s, w, h, aa=1):
        o = np.zeros((h * w * aa * aa, 3), f32)
        d = np.zeros_like(o)
        self.L.oracle_primary_rays(s, w, h, aa, _p(o), _p(d))
        return o, d

    def intersect(self, s, o, d):
        o = np.ascontiguousarray(o, f32)
        d = np.ascontiguousarray(d, f32)
        n = o.shape[0]
        rank = np.zeros(n, np.int32)
        t = np.zeros(n, f32)
        pos = np.zeros((n, 3), f32)
        nrm = np.zeros((n, 3), f32)
        rd = np.zeros((n, 3), f32)
        self.L.oracle_intersect(s, n, _p(o), _p(d), _p(rank), _p(t), _p(pos), _p(nrm), _p(rd))
        return dict(rank=rank, hit=(rank >= 0), t=t, pos=pos, normal=nrm, raydir=rd)

    def prim_intersect(self, kind, params, xforms, o, d):
        p = np.zeros(9, f32)
        p[: len(params)] = params
        xf = np.ascontiguousarray(np.array(xforms, f32).reshape(-1, 4)) if len(xforms) else np.zeros((0, 4), f32)
        o = np.ascontiguousarray(o, f32).reshape(-1, 3)
        d = np.ascontiguousarray(d, f32).reshape(-1, 3)
        n = o.shape[0]
        hit = np.zeros(n, np.int32)
        t = np.zeros(n, f32)
        pos = np.zeros((n, 3), f32)
        nrm = np.zeros((n, 3), f32)
        rd = np.zeros((n, 3), f32)
        rc = self.L.oracle_prim_intersect(kind, _p(p), _p(xf), xf.shape[0], n, _p(o), _p(d), _p(hit), _p(t), _p(pos), _p(nrm), _p(rd))
        if rc != 0:
            raise ValueError("Box minimum cannot be greater than maximum")
        return dict(hit=hit.astype(bool), t=t, pos=pos, normal=nrm, raydir=rd)

    def box_bounds(self, params, xforms):
        p = np.array(params, f32)
        xf = np.ascontiguousarray(np.array(xforms, f32).reshape(-1, 4)) if len(xforms) else np.zeros((0, 4), f32)
        mn = np.zeros(3, f32)
        mx = np.zeros(3, f32)
        self.L.oracle_box_bounds(_p(p), _p(xf), xf.shape[0], _p(mn), _p(mx))
        return mn, mx

    def scene_info(self, s):
        cam = np.zeros(10, f32)
        amb = np.zeros(3, f32)
        nl = C.c_int(0)
        lights = np.zeros((16, 6), f32)
        mn = np.zeros(3, f32)
        mx = np.zeros(3, f32)
        self.L.oracle_scene_info(s, _p(cam), _p(amb), C.byref(nl), _p(lights), _p(mn), _p(mx))
        return dict(camera=cam, ambient=amb, lights=lights[: nl.value].copy(), root_min=mn, root_max=mx)


class Reference:
    """The real reference (oracle/_ref/libref.so).  The scene file must live in a folder
    called ``sdf`` because the reference opens OBJ files at ``../../sdf/`` (scene.cpp:326)."""

    kind = "reference"

    def __init__(self, path=None):
        path = path or os.path.join(HERE, "_ref", "libref.so")
        if not os.path.exists(path):
            raise FileNotFoundError(path)
        L = self.L = C.CDLL(path)
        L.ref_load_scene.restype = C.c_void_p
        L.ref_load_scene.argtypes = [C.c_char_p]
        L.ref_free_scene.argtypes = [C.c_void_p]
        L.ref_render.restype = C.c_double
        L.ref_render.argtypes = [C.c_void_p, C.c_uint, C.c_uint, C.c_uint, C.c_uint, C.c_char_p, C.c_void_p]
        L.ref_primary_rays.argtypes = [C.c_void_p, C.c_uint, C.c_uint, C.c_uint, C.c_void_p, C.c_void_p]
        L.ref_intersect.argtypes = [C.c_void_p, C.c_long] + [C.c_void_p] * 8
        L.ref_prim_intersect.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_long] + [C.c_void_p] * 7
        L.ref_scene_info.argtypes = [C.c_void_p] + [C.c_void_p] * 6

    @staticmethod
    def available():
        return os.path.exists(os.path.join(HERE, "_ref", "libref.so"))

    def load(self, sdf_path, obj_dir=None):
        s = self.L.ref_load_scene(os.path.abspath(sdf_path).encode())
        if not s:
            raise RuntimeError("reference could not load " + sdf_path)
        return s

    def free(self, s):
        self.L.ref_free_scene(s)

    def add_group(self, s, name, children):
        _add_group(self.L.ref_add_group, s, name, children)

    def transform_child(self, s, path, op, a, b, c):
        self.L.ref_transform_child.argtypes = [C.c_void_p, C.c_char_p, C.c_int, C.c_float, C.c_float, C.c_float]
        if self.L.ref_transform_child(s, path.encode(), OPS[op], a, b, c) != 0:
            raise RuntimeError("no child " + path)

    def render(self, s, w, h, aa, bounces, ppm_path="/tmp/_ref_out.ppm", **_):
        cb = np.zeros((h, w, 3), f32)
        sec = self.L.ref_render(s, w, h, aa, bounces, ppm_path.encode(), _p(cb))
        with open(ppm_path, "rb") as f:
            text = f.read()
        vals = np.array(text.split()[4:], dtype=np.int32).astype(np.uint8).reshape(h, w, 3)
        return dict(color=cb, rgb=vals, seconds=sec, ppm_text=text, counters=None)

    def primary_rays(self, s, w, h, aa=1):
        o = np.zeros((h * w * aa * aa, 3), f32)
        d = np.zeros_like(o)
        self.L.ref_primary_rays(s, w, h, aa, _p(o), _p(d))
        return o, d

    def intersect(self, s, o, d):
        o = np.ascontiguousarray(o, f32)
        d = np.ascontiguousarray(d, f32)
        n = o.shape[0]
        hit = np.zeros(n, np.int32)
        t = np.zeros(n, f32)
        pos = np.zeros((n, 3), f32)
        nrm = np.zeros((n, 3), f32)
        rd = np.zeros((n, 3), f32)
        names = np.zeros((n, 48), np.uint8)
        self.L.ref_intersect(s, n, _p(o), _p(d), _p(hit), _p(t), _p(pos), _p(nrm), _p(rd), _p(names))
        nm = [bytes(r).split(b"\0", 1)[0].decode() for r in names]
        return dict(hit=hit.astype(bool), t=t, pos=pos, normal=nrm, raydir=rd, names=nm)

    def prim_intersect(self, kind, params, xforms, o, d):
        p = np.zeros(9, f32)
        p[: len(params)] = params
        xf = np.ascontiguousarray(np.array(xforms, f32).reshape(-1, 4)) if len(xforms) else np.zeros((0, 4), f32)
        o = np.ascontiguousarray(o, f32).reshape(-1, 3)
        d = np.ascontiguousarray(d, f32).reshape(-1, 3)
        n = o.shape[0]
        hit = np.zeros(n, np.int32)
        t = np.zeros(n, f32)
        pos = np.zeros((n, 3), f32)
        nrm = np.zeros((n, 3), f32)
        rd = np.zeros((n, 3), f32)
        self.L.ref_prim_intersect(kind, _p(p), _p(xf), xf.shape[0], n, _p(o), _p(d), _p(hit), _p(t), _p(pos), _p(nrm), _p(rd))
        return dict(hit=hit.astype(bool), t=t, pos=pos, normal=nrm, raydir=rd)

    def scene_info(self, s):
        cam = np.zeros(10, f32)
        amb = np.zeros(3, f32)
        nl = C.c_int(0)
        lights = np.zeros((16, 6), f32)
        mn = np.zeros(3, f32)
        mx = np.zeros(3, f32)
        self.L.ref_scene_info(s, _p(cam), _p(amb), C.byref(nl), _p(lights), _p(mn), _p(mx))
        return dict(camera=cam, ambient=amb, lights=lights[: nl.value].copy(), root_min=mn, root_max=mx)


def random_rays(n, seed, lo=(-8, -2, -8), hi=(8, 12, 16)):
    """Incoherent test rays: origins in a box around the scenes, directions on the sphere
    (NOT normalised exactly - the path takes any direction length)."""
    rng = np.random.default_rng(seed)
    o = rng.uniform(lo, hi, size=(n, 3)).astype(f32)
    d = rng.normal(size=(n, 3)).astype(f32)
    d *= rng.uniform(0.5, 2.0, size=(n, 1)).astype(f32) / np.linalg.norm(d, axis=1, keepdims=True).astype(f32)
    return o, d.astype(f32)
