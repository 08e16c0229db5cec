"""Python face of the MI355X render path (ctypes over the C ABI of include/ptx.h).

The work is done by two in-tree shared libraries built by ``make -C path-tracer_amd/csrc``
(or ``__graft_entry__.build()``):

* ``libptx.so``      - HIP kernels for gfx950 + the C ABI (the drop-in boundary);
* ``libpt_host.so``  - host-side C++ mirror of the reference's Scene / Shape / Composite /
  Renderer API (scene loading, octree, flattening).

There is no CPU implementation behind this module: if the libraries are missing, or no
gfx950 device is present, the calls raise.

A process that also uses torch should import torch BEFORE the first call into this module:
torch's wheel brings a HIP runtime of its own with the system runtime's soname, and whichever
is loaded first serves both; loaded the other way round the process holds two runtimes.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(_HERE)
SDF_DIR = os.path.join(REPO, "fixtures", "sdf")

PTX_OK = 0
STATUS_NAMES = {0: "PTX_OK", 1: "PTX_ERR_INVALID", 2: "PTX_ERR_UNSUPPORTED", 3: "PTX_ERR_DEVICE", 4: "PTX_ERR_OVERFLOW", 5: "PTX_ERR_NOMEM"}

# every symbol include/ptx.h declares (checked by tests/test_abi.py)
PTX_SYMBOLS = [
    "ptx_last_error", "ptx_version", "ptx_device_count", "ptx_scene_create", "ptx_scene_destroy", "ptx_render_accumulate",
    "ptx_resolve", "ptx_render_image", "ptx_intersect", "ptx_primary_hits", "ptx_write_ppm", "ptx_write_ppm_binary", "ptx_write_pfm", "ptx_scene_info", "ptx_jitter",
    "ptx_scene_set_profiling", "ptx_render_accumulate_static", "ptx_render_image_multi", "ptx_write_image_async", "ptx_write_wait",
    "ptx_accum_save", "ptx_accum_load", "ptx_trim", "ptx_reduce_accum",
]


class PtxError(RuntimeError):
    def __init__(self, status, message):
        super().__init__(f"{STATUS_NAMES.get(status, status)}: {message}")
        self.status = status


class PtxMaterial(C.Structure):
    _fields_ = [("ka", C.c_float * 3), ("kd", C.c_float * 3), ("ks", C.c_float * 3), ("m", C.c_float), ("glossy", C.c_float),
                ("opacity", C.c_float), ("ior", C.c_float)]


class PtxLight(C.Structure):
    _fields_ = [("position", C.c_float * 3), ("intensity", C.c_float * 3)]


class PtxCamera(C.Structure):
    _fields_ = [("fov_x", C.c_float), ("position", C.c_float * 3), ("direction", C.c_float * 3), ("up", C.c_float * 3)]


class PtxFrame(C.Structure):
    _fields_ = [("world", C.c_float * 16), ("world_inv", C.c_float * 16), ("bounds_min", C.c_float * 3), ("bounds_max", C.c_float * 3),
                ("parent", C.c_int32), ("levels_up", C.c_int32)]


class PtxPrim(C.Structure):
    _fields_ = [("kind", C.c_int32), ("material", C.c_int32), ("frame", C.c_int32), ("levels_in", C.c_int32), ("p", C.c_float * 9),
                ("normal", C.c_float * 3), ("world", C.c_float * 16), ("world_inv", C.c_float * 16)]


class PtxGuard(C.Structure):
    _fields_ = [("bounds_min", C.c_float * 3), ("bounds_max", C.c_float * 3), ("parent", C.c_int32), ("levels", C.c_int32)]


class PtxOccurrence(C.Structure):
    _fields_ = [("guard", C.c_int32), ("prim", C.c_int32), ("frame", C.c_int32), ("reserved", C.c_int32)]


class PtxSceneDesc(C.Structure):
    _fields_ = [("frames", C.POINTER(PtxFrame)), ("n_frames", C.c_int32), ("prims", C.POINTER(PtxPrim)), ("n_prims", C.c_int32),
                ("materials", C.POINTER(PtxMaterial)), ("n_materials", C.c_int32), ("lights", C.POINTER(PtxLight)), ("n_lights", C.c_int32),
                ("ambient", C.c_float * 3), ("camera", PtxCamera),
                ("guards", C.POINTER(PtxGuard)), ("n_guards", C.c_int32), ("occurrences", C.POINTER(PtxOccurrence)), ("n_occurrences", C.c_int32)]


class PtxRenderParams(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("aa_steps", C.c_uint32), ("max_ray_bounces", C.c_uint32),
                ("sample_begin", C.c_uint32), ("sample_end", C.c_uint32), ("pipeline", C.c_uint32), ("jitter_seed", C.c_uint32)]


class PtxStats(C.Structure):
    _fields_ = [("samples", C.c_uint64), ("trace_rays", C.c_uint64), ("shadow_rays", C.c_uint64), ("shaded_hits", C.c_uint64),
                ("launches", C.c_uint64), ("per_depth_rays", C.c_uint64 * 16), ("per_depth_shadow", C.c_uint64 * 16),
                ("per_depth_hits", C.c_uint64 * 16), ("kernel_ms", (C.c_double * 16) * 3), ("kernel_launches", (C.c_uint64 * 16) * 3),
                ("trace_tests", (C.c_uint64 * 5) * 16), ("shadow_tests", (C.c_uint64 * 5) * 16), ("tests_counted", C.c_uint64), ("chunks_redone", C.c_uint64)]

    def as_dict(self):
        return dict(samples=self.samples, trace_rays=self.trace_rays, shadow_rays=self.shadow_rays, shaded_hits=self.shaded_hits,
                    launches=self.launches, per_depth_rays=list(self.per_depth_rays), per_depth_shadow=list(self.per_depth_shadow),
                    per_depth_hits=list(self.per_depth_hits), kernel_ms=[list(r) for r in self.kernel_ms],
                    kernel_launches=[list(r) for r in self.kernel_launches], trace_tests=[list(r) for r in self.trace_tests],
                    shadow_tests=[list(r) for r in self.shadow_tests], tests_counted=bool(self.tests_counted), chunks_redone=self.chunks_redone)


_libs = {}
_ptx_path = None


def set_library(path):
    """Load the render library from `path` instead of the in-tree libptx.so (for comparing two
    builds with bench.py --lib); must be called before the first use."""
    global _ptx_path
    if "ptx" in _libs:
        raise RuntimeError("the render library is already loaded")
    _ptx_path = os.path.abspath(path)


def _load(name):
    if name not in _libs:
        path = os.path.join(_HERE, name)
        if not os.path.exists(path):
            raise FileNotFoundError(f"{path} is missing: build it with `make -C path-tracer_amd/csrc` (there is no fallback path)")
        _libs[name] = C.CDLL(path)
    return _libs[name]


def ptx_lib():
    if "ptx" not in _libs:
        if _ptx_path is not None:
            if not os.path.exists(_ptx_path):
                raise FileNotFoundError(_ptx_path)
            _libs["ptx"] = C.CDLL(_ptx_path)
        else:
            _libs["ptx"] = _load("libptx.so")
    L = _libs["ptx"]
    if getattr(L, "_typed", False):
        return L
    vp = C.c_void_p
    L.ptx_last_error.restype = C.c_char_p
    L.ptx_version.restype = C.c_char_p
    L.ptx_device_count.restype = C.c_int
    L.ptx_scene_create.argtypes = [C.POINTER(PtxSceneDesc), C.c_int, C.c_uint64, C.POINTER(vp)]
    L.ptx_scene_destroy.argtypes = [vp]
    L.ptx_scene_destroy.restype = None
    L.ptx_render_accumulate.argtypes = [vp, C.POINTER(PtxRenderParams), vp, vp, C.c_int, C.POINTER(PtxStats)]
    L.ptx_render_accumulate_static.argtypes = L.ptx_render_accumulate.argtypes
    L.ptx_render_image_multi.argtypes = [C.POINTER(vp), C.c_int32, C.POINTER(PtxRenderParams), vp, vp, C.POINTER(C.c_double), C.POINTER(PtxStats)]
    L.ptx_write_image_async.argtypes = [C.c_char_p, C.c_int, C.c_int, vp, C.c_uint32, C.c_uint32, vp, C.POINTER(vp)]
    L.ptx_write_wait.argtypes = [vp]
    L.ptx_accum_save.argtypes = [C.c_char_p, C.c_int, vp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, vp]
    L.ptx_accum_load.argtypes = [C.c_char_p, C.c_int, vp, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), vp]
    L.ptx_resolve.argtypes = [C.c_int, vp, C.c_uint32, C.c_uint32, C.c_uint32, vp, vp, vp]
    L.ptx_render_image.argtypes = [vp, C.POINTER(PtxRenderParams), vp, vp, C.POINTER(C.c_double), C.POINTER(PtxStats)]
    L.ptx_intersect.argtypes = [vp, C.c_int64, vp, vp, vp, vp, vp, vp, vp]
    L.ptx_primary_hits.argtypes = [vp, C.POINTER(PtxRenderParams), C.c_uint32, vp, vp]
    L.ptx_write_ppm.argtypes = [C.c_char_p, vp, C.c_uint32, C.c_uint32]
    L.ptx_write_ppm_binary.argtypes = [C.c_char_p, vp, C.c_uint32, C.c_uint32]
    L.ptx_write_pfm.argtypes = [C.c_char_p, vp, C.c_uint32, C.c_uint32]
    L.ptx_jitter.restype = None
    L.ptx_trim.restype = None
    L.ptx_trim.argtypes = []
    L.ptx_reduce_accum.argtypes = [vp, vp, vp, C.c_uint64, C.c_int, vp]
    L.ptx_jitter.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_float), C.POINTER(C.c_float)]
    L.ptx_scene_set_profiling.argtypes = [vp, C.c_int]
    L.ptx_scene_info.argtypes = [vp, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_uint64)]
    L._typed = True
    return L


def host_lib():
    ptx_lib()  # libpt_host.so links against libptx.so
    H = _load("libpt_host.so")
    if getattr(H, "_typed", False):
        return H
    H.pth_last_error.restype = C.c_char_p
    H.pth_load_scene.restype = C.c_void_p
    H.pth_load_scene.argtypes = [C.c_char_p, C.c_char_p]
    H.pth_free_scene.argtypes = [C.c_void_p]
    H.pth_free_scene.restype = None
    H.pth_scene_desc.restype = C.POINTER(PtxSceneDesc)
    H.pth_scene_desc.argtypes = [C.c_void_p]
    H.pth_scene_transform_child.argtypes = [C.c_void_p, C.c_char_p, C.c_int, C.c_float, C.c_float, C.c_float]
    H.pth_prim_name.restype = C.c_char_p
    H.pth_prim_name.argtypes = [C.c_void_p, C.c_int]
    H.pth_prim_path.restype = C.c_char_p
    H.pth_prim_path.argtypes = [C.c_void_p, C.c_int]
    H.pth_tree_stats.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]
    H.pth_render_file.argtypes = [C.c_void_p, C.c_uint, C.c_uint, C.c_uint, C.c_uint, C.c_char_p, C.c_void_p, C.POINTER(C.c_double)]
    H.pth_renderer_create.restype = C.c_void_p
    H.pth_renderer_create.argtypes = [C.c_uint, C.c_uint, C.c_uint, C.c_uint, C.c_char_p]
    H.pth_renderer_destroy.restype = None
    H.pth_renderer_destroy.argtypes = [C.c_void_p]
    H.pth_renderer_options.restype = None
    H.pth_renderer_options.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int]
    H.pth_renderer_render.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    H.pth_select_devices.argtypes = [C.c_int, C.c_int, C.c_int, C.c_ulonglong, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    H.pth_shape_matrices.restype = None
    H.pth_shape_matrices.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    H._typed = True
    return H


def shape_matrices(xforms):
    """World matrix and inverse (column-major float32[16] each) of a Shape after the transforms
    `xforms` = [[op, a, b, c], ...] (1 translate, 2 rotate in radians, 3 scale), formed by the host
    mirror's Shape::translate / rotate / scale."""
    xf = np.ascontiguousarray(np.array(xforms, np.float32).reshape(-1, 4))
    w, wi = np.zeros(16, np.float32), np.zeros(16, np.float32)
    host_lib().pth_shape_matrices(_ptr(xf), xf.shape[0], _ptr(w), _ptr(wi))
    return w, wi


class DescScene:
    """A scene described directly as PtxSceneDesc arrays (no scene file): `prims` is a list of
    dicts(kind, p, world, world_inv[, normal, material, levels_in]) under an untransformed root
    whose bounds are `bounds` (min xyz, max xyz).  Stands where a HostScene does for DeviceScene."""

    def __init__(self, prims, bounds=((-1e30,) * 3, (1e30,) * 3), materials=None, lights=(), ambient=(0, 0, 0), camera=None):
        ident = np.eye(4, dtype=np.float32).reshape(-1)
        self._frames = (PtxFrame * 1)()
        f = self._frames[0]
        f.world[:] = ident
        f.world_inv[:] = ident
        f.bounds_min[:] = bounds[0]
        f.bounds_max[:] = bounds[1]
        f.parent, f.levels_up = -1, 0
        self._prims = (PtxPrim * max(len(prims), 1))()
        for q, d in zip(self._prims, prims):
            q.kind, q.material, q.frame, q.levels_in = d["kind"], d.get("material", 0), 0, d.get("levels_in", 1)
            pp = np.zeros(9, np.float32)
            pp[: len(d["p"])] = d["p"]
            q.p[:] = pp
            q.normal[:] = d.get("normal", (0, 0, 0))
            q.world[:] = d.get("world", ident)
            q.world_inv[:] = d.get("world_inv", ident)
        mats = materials or [dict()]
        self._mats = (PtxMaterial * len(mats))()
        for m, d in zip(self._mats, mats):
            m.ka[:] = d.get("ka", (0.2, 0.2, 0.2))
            m.kd[:] = d.get("kd", (0.5, 0.5, 0.5))
            m.ks[:] = d.get("ks", (0, 0, 0))
            m.m, m.glossy, m.opacity, m.ior = d.get("m", 0), d.get("glossy", 0), d.get("opacity", 1), d.get("ior", 1)
        self._lights = (PtxLight * max(len(lights), 1))()
        for l, (pos, inten) in zip(self._lights, lights):
            l.position[:] = pos
            l.intensity[:] = inten
        cam = PtxCamera(60.0, (C.c_float * 3)(0, 0, 0), (C.c_float * 3)(0, 0, -1), (C.c_float * 3)(0, 1, 0)) if camera is None else camera
        self._desc = PtxSceneDesc(self._frames, 1, self._prims, len(prims), self._mats, len(mats), self._lights, len(lights),
                                  (C.c_float * 3)(*ambient), cam)
        self.desc = C.pointer(self._desc)


def _check(status):
    if status != PTX_OK:
        raise PtxError(status, ptx_lib().ptx_last_error().decode())


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


PIPELINES = {"default": 0, "fused": 1, "staged": 2}


TRACE_ZERO_WEIGHT = 0x100
REDUCE_RCCL = 0x200


def render_params(width, height, aa_steps, max_ray_bounces, sample_begin=0, sample_end=None, pipeline="default", jitter_seed=0, zero_weight=False):
    """zero_weight: also trace child rays of exactly zero weight, as the reference does (PTX_TRACE_ZERO_WEIGHT)"""
    if sample_end is None:
        sample_end = aa_steps * aa_steps
    return PtxRenderParams(width, height, aa_steps, max_ray_bounces, sample_begin, sample_end,
                           PIPELINES[pipeline] | (TRACE_ZERO_WEIGHT if zero_weight else 0), jitter_seed)


class HostScene:
    """load_scene() + flattening on the host (no GPU needed)."""

    def __init__(self, sdf_path, obj_dir=None):
        H = host_lib()
        self._H = H
        self.handle = H.pth_load_scene(os.fspath(sdf_path).encode(), obj_dir.encode() if obj_dir else None)
        if not self.handle:
            raise RuntimeError("load_scene failed: " + H.pth_last_error().decode())
        self.desc = H.pth_scene_desc(self.handle)

    def close(self):
        if self.handle:
            self._H.pth_free_scene(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def add_group(self, name, children):
        """What a C++ caller does to move several shapes together: a Composite of its own under the root.
        children: (kind 'sphere' | 'box', name, (cx, cy, cz, r) | (min xyz, max xyz), material name)."""
        kinds = np.array([0 if c[0] == "sphere" else 1 for c in children], np.int32)
        params = np.zeros((len(children), 7), np.float32)
        for i, c in enumerate(children):
            params[i, :len(c[2])] = c[2]
        self._H.pth_scene_add_group.argtypes = [C.c_void_p, C.c_char_p, C.c_int, C.c_void_p, C.c_void_p, C.c_char_p, C.c_char_p]
        if self._H.pth_scene_add_group(self.handle, name.encode(), len(children), _ptr(kinds), _ptr(params),
                                       ";".join(c[1] for c in children).encode(), ";".join(c[3] for c in children).encode()) != 0:
            raise RuntimeError(self._H.pth_last_error().decode())
        self.desc = self._H.pth_scene_desc(self.handle)

    def transform_child(self, path, op, a, b, c):
        """What a C++ caller does after load_scene to move a part of a loaded mesh:
        root->find_child(..)->...->translate / rotate (radians) / scale; `path` = names joined by '/'.
        The scene is flattened again (device scenes made from it before keep the old state)."""
        if self._H.pth_scene_transform_child(self.handle, path.encode(), {"translate": 1, "rotate": 2, "scale": 3}[op], a, b, c) != 0:
            raise RuntimeError(self._H.pth_last_error().decode())
        self.desc = self._H.pth_scene_desc(self.handle)

    @property
    def n_prims(self):
        return self.desc.contents.n_prims

    def prim_names(self):
        return [self._H.pth_prim_name(self.handle, i).decode() for i in range(self.n_prims)]

    def prim_paths(self):
        return [self._H.pth_prim_path(self.handle, i).decode() for i in range(self.n_prims)]

    def tree_stats(self):
        a, b, c = C.c_int(), C.c_int(), C.c_int()
        self._H.pth_tree_stats(self.handle, C.byref(a), C.byref(b), C.byref(c))
        return dict(composites=a.value, leaf_entries=b.value, max_depth=c.value)

    def render_file(self, w, h, aa, bounces, ppm_path=None, want_color=True):
        """The reference's call sequence: Renderer{w,h,file,aa,bounces}.render(scene)."""
        color = np.zeros((h, w, 3), np.float32) if want_color else None
        sec = C.c_double(0)
        rc = self._H.pth_render_file(self.handle, w, h, aa, bounces, ppm_path.encode() if ppm_path else None, _ptr(color), C.byref(sec))
        if rc != 0:
            raise RuntimeError(self._H.pth_last_error().decode())
        return color, sec.value


def select_devices(first_device, n_devices, visible, samples):
    """Renderer::select_devices: (first, count) of the GPUs a selection comes to, or None if they are not there."""
    a, b = C.c_int(0), C.c_int(0)
    if host_lib().pth_select_devices(first_device, n_devices, visible, samples, C.byref(a), C.byref(b)) != 0:
        return None
    return a.value, b.value


class HostRenderer:
    """The host-side C++ Renderer (the reference's class, rendering on the GPUs) kept across calls.
    first_device < 0: the default, one GPU (the first); else n_devices GPUs from first_device on (0 = all from there on)."""

    def __init__(self, w, h, aa, bounces, ppm_path=None, first_device=-1, n_devices=0, keep_scene=False):
        self._H = host_lib()
        self.w, self.h = w, h
        self.handle = self._H.pth_renderer_create(w, h, aa, bounces, os.fspath(ppm_path).encode() if ppm_path else None)
        self._H.pth_renderer_options(self.handle, first_device, n_devices, int(keep_scene))

    def render(self, host_scene):
        color = np.zeros((self.h, self.w, 3), np.float32)
        dev, setup = C.c_double(0), C.c_double(0)
        if self._H.pth_renderer_render(self.handle, host_scene.handle, _ptr(color), C.byref(dev), C.byref(setup)) != 0:
            raise RuntimeError(self._H.pth_last_error().decode())
        return dict(color=color, seconds=dev.value, setup_seconds=setup.value)

    def close(self):
        if self.handle:
            self._H.pth_renderer_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class DeviceScene:
    """A scene resident on one GPU (ptx_scene_create)."""

    def __init__(self, host_scene, device=0, queue_bytes=0):
        self.L = ptx_lib()
        self.host = host_scene
        self.device = device
        h = C.c_void_p()
        _check(self.L.ptx_scene_create(host_scene.desc, device, queue_bytes, C.byref(h)))
        self.handle = h

    def close(self):
        if self.handle:
            self.L.ptx_scene_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_profiling(self, level=1):
        """0 off, 1 HIP events around every launch, 2 events + test counting (slower kernels)"""
        _check(self.L.ptx_scene_set_profiling(self.handle, int(level)))

    def info(self):
        a, b, c, d, e = C.c_int32(), C.c_int32(), C.c_int32(), C.c_int32(), C.c_uint64()
        _check(self.L.ptx_scene_info(self.handle, C.byref(a), C.byref(b), C.byref(c), C.byref(d), C.byref(e)))
        return dict(n_frames=a.value, n_prims=b.value, n_bvh_nodes=c.value, max_bvh_depth=d.value, device_bytes=e.value)

    def render_image(self, w, h, aa, bounces, sample_begin=0, sample_end=None, pipeline="default", jitter_seed=0, zero_weight=False):
        rp = render_params(w, h, aa, bounces, sample_begin, sample_end, pipeline, jitter_seed, zero_weight)
        color = np.zeros((h, w, 3), np.float32)
        rgb = np.zeros((h, w, 3), np.uint8)
        sec = C.c_double(0)
        st = PtxStats()
        _check(self.L.ptx_render_image(self.handle, C.byref(rp), _ptr(color), _ptr(rgb), C.byref(sec), C.byref(st)))
        return dict(color=color, rgb=rgb, seconds=sec.value, stats=st.as_dict())

    def render_accumulate(self, rp, d_accum_ptr, stream=None, check_overflow=False, want_stats=False, static=False):
        """static=True: ptx_render_accumulate_static (work memory inside the library, nothing allocated)"""
        st = PtxStats() if want_stats else None
        fn = self.L.ptx_render_accumulate_static if static else self.L.ptx_render_accumulate
        _check(fn(self.handle, C.byref(rp), C.c_void_p(d_accum_ptr), C.c_void_p(stream or 0), int(check_overflow),
                                            C.byref(st) if st is not None else None))
        return st.as_dict() if st is not None else None

    def resolve(self, d_accum_ptr, w, h, spp, d_color_ptr=None, d_rgb_ptr=None, stream=None):
        _check(self.L.ptx_resolve(self.device, C.c_void_p(d_accum_ptr), w, h, spp, C.c_void_p(d_color_ptr or 0), C.c_void_p(d_rgb_ptr or 0),
                                  C.c_void_p(stream or 0)))

    def intersect(self, origins, dirs):
        o = np.ascontiguousarray(origins, np.float32).reshape(-1, 3)
        d = np.ascontiguousarray(dirs, np.float32).reshape(-1, 3)
        n = o.shape[0]
        prim = np.zeros(n, np.int32)
        t = np.zeros(n, np.float32)
        pos = np.zeros((n, 3), np.float32)
        nrm = np.zeros((n, 3), np.float32)
        rd = np.zeros((n, 3), np.float32)
        _check(self.L.ptx_intersect(self.handle, n, _ptr(o), _ptr(d), _ptr(prim), _ptr(t), _ptr(pos), _ptr(nrm), _ptr(rd)))
        return dict(prim=prim, hit=prim >= 0, t=t, pos=pos, normal=nrm, raydir=rd)

    def primary_hits(self, w, h, aa=1, sample=0):
        rp = render_params(w, h, aa, 0)
        prim = np.zeros(w * h, np.int32)
        t = np.zeros(w * h, np.float32)
        _check(self.L.ptx_primary_hits(self.handle, C.byref(rp), sample, _ptr(prim), _ptr(t)))
        return prim, t


IMAGE_FORMATS = {"ppm": 0, "ppm_binary": 1, "pfm": 2}


def render_image_multi(scenes, w, h, aa, bounces, sample_begin=0, sample_end=None, pipeline="default", jitter_seed=0, rccl=False):
    """ptx_render_image_multi over a list of DeviceScene objects (one per GPU).  rccl: sum the buffers with one RCCL
    reduce (PTX_REDUCE_RCCL) instead of the first GPU reading its peers' memory."""
    L = ptx_lib()
    rp = render_params(w, h, aa, bounces, sample_begin, sample_end, pipeline, jitter_seed)
    if rccl:
        rp.pipeline |= REDUCE_RCCL
    color = np.zeros((h, w, 3), np.float32)
    rgb = np.zeros((h, w, 3), np.uint8)
    sec = C.c_double(0)
    stats = (PtxStats * len(scenes))()
    handles = (C.c_void_p * len(scenes))(*[s.handle for s in scenes])
    _check(L.ptx_render_image_multi(handles, len(scenes), C.byref(rp), _ptr(color), _ptr(rgb), C.byref(sec), stats))
    return dict(color=color, rgb=rgb, seconds=sec.value, stats=[s.as_dict() for s in stats])


class ImageWrite:
    """ptx_write_image_async: the file is written by a worker thread of the library; wait() joins it."""

    def __init__(self, path, fmt, device, d_pixels_ptr, w, h, stream=None):
        self.L = ptx_lib()
        self.job = C.c_void_p()
        _check(self.L.ptx_write_image_async(os.fspath(path).encode(), IMAGE_FORMATS[fmt], device, C.c_void_p(d_pixels_ptr), w, h,
                                            C.c_void_p(stream or 0), C.byref(self.job)))

    def wait(self):
        job, self.job = self.job, None
        if job:
            _check(self.L.ptx_write_wait(job))


def accum_save(path, device, d_accum_ptr, w, h, aa, sample_begin, sample_end, stream=None):
    _check(ptx_lib().ptx_accum_save(os.fspath(path).encode(), device, C.c_void_p(d_accum_ptr), w, h, aa, sample_begin, sample_end, C.c_void_p(stream or 0)))


def accum_load(path, device, d_accum_ptr, w, h, aa, stream=None):
    b, e = C.c_uint32(0), C.c_uint32(0)
    _check(ptx_lib().ptx_accum_load(os.fspath(path).encode(), device, C.c_void_p(d_accum_ptr), w, h, aa, C.byref(b), C.byref(e), C.c_void_p(stream or 0)))
    return b.value, e.value


def write_ppm(path, rgb):
    h, w, _ = rgb.shape
    rgb = np.ascontiguousarray(rgb, np.uint8)
    _check(ptx_lib().ptx_write_ppm(os.fspath(path).encode(), _ptr(rgb), w, h))


def write_ppm_binary(path, rgb):
    h, w, _ = rgb.shape
    rgb = np.ascontiguousarray(rgb, np.uint8)
    _check(ptx_lib().ptx_write_ppm_binary(os.fspath(path).encode(), _ptr(rgb), w, h))


def write_pfm(path, color):
    h, w, _ = color.shape
    color = np.ascontiguousarray(color, np.float32)
    _check(ptx_lib().ptx_write_pfm(os.fspath(path).encode(), _ptr(color), w, h))


def device_count():
    return ptx_lib().ptx_device_count()


def trim():
    """ptx_trim: return the queue blocks the library kept from destroyed scenes to their devices."""
    ptx_lib().ptx_trim()


def reduce_accum(comm, d_send_ptr, d_recv_ptr, n_floats, root=0, stream=None):
    """ptx_reduce_accum: one RCCL sum-reduce of an accumulation buffer onto rank `root` of the communicator `comm`
    (an ncclComm_t as an integer / c_void_p)."""
    _check(ptx_lib().ptx_reduce_accum(C.c_void_p(comm), C.c_void_p(d_send_ptr), C.c_void_p(d_recv_ptr or 0), n_floats, root, C.c_void_p(stream or 0)))
