"""functracer_amd — Python face of the MI355X-native FuncTracer render loop.

The product is `lib/libfunctracer_hip.so` (hand-written gfx950 kernels behind the C ABI of
include/functracer_hip.h) plus `lib/libfunctracer_host.so` (the host side that is F# in the
reference: SceneParser / PlyParser / Program).  This package only binds them; nothing here
computes pixels, and nothing here touches the CPU oracle under oracle/.  If the HIP library or a
GPU is missing, `Context()` raises — there is no fallback path.
"""
import ctypes as C
import os

import numpy as np

from . import _capi
from ._capi import (CIRCLE, CONE, CUBE, CYLINDER, EXCLUDE, INTERSECT, PLANE, SOLID_CYLINDER, SPHERE, SQUARE, SUBTRACT, UNION,
                    FtError, SceneBuilder, make_camera)

__all__ = ["Context", "PinnedArray", "ParsedScene", "parse_scene", "parse_scene_file", "jitter_pattern", "quantise_rgba8", "write_png",
           "parse_colour", "parse_ply", "FtError", "make_camera", "HIP_LIB", "HOST_LIB", "DEFAULT_SEED"]

HIP_LIB = os.environ.get("FT_HIP_LIB") or os.path.join(_capi.LIB_DIR, "libfunctracer_hip.so")   # FT_HIP_LIB: experimental builds
HOST_LIB = os.environ.get("FT_HOST_LIB") or os.path.join(_capi.LIB_DIR, "libfunctracer_host.so")
DEFAULT_SEED = 20260104          # jitter pattern seed of the BASELINE configs
MAX_DEPTH = 8                    # Shading.fs:142

_hip = None
_host = None


def hip_lib():
    global _hip
    if _hip is None:
        lib = _capi.load_library(HIP_LIB)
        lib.ft_create.argtypes = [_capi.c_int32_p, C.c_int32, C.POINTER(C.c_void_p)]
        lib.ft_create_host_only.argtypes = [C.POINTER(C.c_void_p)]
        lib.ft_destroy.argtypes = [C.c_void_p]
        lib.ft_destroy.restype = None
        lib.ft_set_option.argtypes = [C.c_void_p, C.c_char_p, C.c_int64]
        lib.ft_render.argtypes = [C.c_void_p, C.POINTER(_capi.ft_camera), C.c_int32, C.c_int32, C.c_int32, _capi.c_double_p, C.c_int32,
                                  C.c_uint64, C.POINTER(_capi.ft_rect), C.c_int32, _capi.c_double_p, C.POINTER(_capi.ft_stats)]
        lib.ft_fetch_frame.argtypes = [C.c_void_p, _capi.c_double_p]
        lib.ft_render_rgba8.argtypes = [C.c_void_p, C.POINTER(_capi.ft_camera), C.c_int32, C.c_int32, C.c_int32, _capi.c_double_p, C.c_int32,
                                        C.c_uint64, C.POINTER(_capi.ft_rect), C.c_int32, C.POINTER(C.c_uint8), C.POINTER(_capi.ft_stats)]
        lib.ft_fetch_frame_rgba8.argtypes = [C.c_void_p, C.POINTER(C.c_uint8)]
        lib.ft_render_enqueue_rgba8.argtypes = [C.c_void_p, C.POINTER(_capi.ft_camera), C.c_int32, C.c_int32, C.c_int32, _capi.c_double_p, C.c_int32, C.c_uint64, C.POINTER(_capi.ft_rect), C.c_int32]
        lib.ft_host_alloc.restype = C.c_void_p
        lib.ft_host_alloc.argtypes = [C.c_size_t]
        lib.ft_host_free.restype = None
        lib.ft_host_free.argtypes = [C.c_void_p]
        lib.ft_debug_closest.argtypes = [C.c_void_p, _capi.c_double_p, _capi.c_double_p, C.c_int64, _capi.c_int32_p,
                                         _capi.c_double_p, _capi.c_double_p, _capi.c_double_p, _capi.c_double_p]
        lib.ft_debug_blocked.argtypes = [C.c_void_p, _capi.c_double_p, _capi.c_double_p, _capi.c_double_p, C.c_int64, _capi.c_int32_p]
        lib.ft_debug_colour.argtypes = [C.c_void_p, _capi.c_double_p, _capi.c_double_p, C.c_int64, C.c_int32, _capi.c_double_p]
        lib.ft_get_commit_times.argtypes = [C.c_void_p, _capi.c_double_p]
        lib.ft_debug_scene_info.argtypes = [C.c_void_p, C.POINTER(C.c_int64)]
        lib.ft_debug_devices.argtypes = [C.c_void_p, _capi.c_int32_p, C.c_int32]
        lib.ft_debug_slice.argtypes = [_capi.c_double_p] * 4 + [_capi.c_int32_p, _capi.c_double_p, _capi.c_int32_p]
        lib.ft_render_enqueue.argtypes = [C.c_void_p, C.POINTER(_capi.ft_camera), C.c_int32, C.c_int32, C.c_int32, _capi.c_double_p, C.c_int32, C.c_uint64, C.POINTER(_capi.ft_rect), C.c_int32]
        lib.ft_render_wait.argtypes = [C.c_void_p, C.POINTER(_capi.ft_stats)]
        lib.ft_render_enqueue_into.argtypes = [C.c_void_p, C.POINTER(_capi.ft_camera), C.c_int32, C.c_int32, C.c_int32, _capi.c_double_p, C.c_int32, C.c_uint64, C.POINTER(_capi.ft_rect), C.c_int32, C.c_int32, C.c_void_p]
        lib.ft_get_kernel_times.argtypes = [C.c_void_p, _capi.c_double_p, _capi.c_int32_p]
        lib.ft_quantise_rgba8.argtypes = [_capi.c_double_p, C.c_int64, C.POINTER(C.c_uint8)]
        _hip = lib
    return _hip


def host_lib():
    global _host
    if _host is None:
        lib = _capi.load_library(HOST_LIB)
        lib.fth_parse_scene.restype = C.c_void_p
        lib.fth_parse_scene.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_int32]
        lib.fth_parse_scene_file.restype = C.c_void_p
        lib.fth_parse_scene_file.argtypes = [C.c_char_p, C.c_char_p, C.c_int32]
        lib.fth_scene_free.argtypes = [C.c_void_p]
        lib.fth_scene_free.restype = None
        lib.fth_scene_options.argtypes = [C.c_void_p, C.POINTER(_capi.fth_options)]
        lib.fth_scene_counts.argtypes = [C.c_void_p, _capi.c_int32_p, _capi.c_int32_p]
        lib.fth_scene_lower.argtypes = [C.c_void_p, C.POINTER(_capi.fth_builder), C.c_void_p]
        lib.fth_parse_colour.argtypes = [C.c_char_p, _capi.c_double_p]
        lib.fth_parse_ply.restype = C.c_int64
        lib.fth_parse_ply.argtypes = [C.c_char_p, _capi.c_double_p, C.c_int64, C.c_char_p, C.c_int32]
        lib.fth_jitter_pattern.argtypes = [C.c_uint64, C.c_int32, _capi.c_double_p]
        lib.fth_write_png.argtypes = [C.c_char_p, C.POINTER(C.c_uint8), C.c_int32, C.c_int32]
        lib.fth_load_image.restype = C.c_int64
        lib.fth_load_image.argtypes = [C.c_char_p, _capi.c_int32_p, _capi.c_int32_p, C.POINTER(C.c_uint8), C.c_int64, C.c_char_p, C.c_int32]
        _host = lib
    return _host


class ParsedScene:
    """Result of SceneParser.parse (SceneParser.fs:360-366): (SceneOptions, Scene)."""

    def __init__(self, handle):
        self._h = handle
        opt = _capi.fth_options()
        host_lib().fth_scene_options(handle, C.byref(opt))
        self.camera = opt.camera
        self.resolution = (opt.res_h, opt.res_v)
        self.samples = opt.samples
        self.corner = bool(opt.corner)
        n_obj, n_l = C.c_int32(), C.c_int32()
        host_lib().fth_scene_counts(handle, C.byref(n_obj), C.byref(n_l))
        self.n_objects, self.n_lights = n_obj.value, n_l.value

    def lower(self, builder):
        """Hand the scene to anything with a SceneBuilder (`Context`, or the oracle in tests)."""
        rc = host_lib().fth_scene_lower(self._h, C.byref(builder.table), builder._ctx)
        if rc < 0:
            raise FtError(rc, builder.last_error())

    def __del__(self):
        if getattr(self, "_h", None) and _host is not None:
            _host.fth_scene_free(self._h)
            self._h = None


def parse_scene(text, base_dir=""):
    err = C.create_string_buffer(2048)
    h = host_lib().fth_parse_scene(text.encode(), base_dir.encode(), err, len(err))
    if not h:
        raise ValueError(err.value.decode())
    return ParsedScene(h)


def parse_scene_file(path):
    err = C.create_string_buffer(2048)
    h = host_lib().fth_parse_scene_file(os.fspath(path).encode(), err, len(err))
    if not h:
        raise ValueError(err.value.decode())
    return ParsedScene(h)


def parse_colour(text):
    rgb = np.zeros(3)
    if host_lib().fth_parse_colour(text.encode(), _capi.dptr(rgb)) != 0:
        raise ValueError("Parsing failed")
    return tuple(rgb)


def parse_ply(text):
    err = C.create_string_buffer(1024)
    n = host_lib().fth_parse_ply(text.encode(), None, 0, err, len(err))
    if n < 0:
        raise ValueError(err.value.decode())
    out = np.zeros((n, 9))
    host_lib().fth_parse_ply(text.encode(), _capi.dptr(out), n, err, len(err))
    return out


def jitter_pattern(n, seed=DEFAULT_SEED):
    """Jitter.pattern random Jitter.circle n on the documented seeded stream (host_api.h)."""
    out = np.zeros((n, 2))
    host_lib().fth_jitter_pattern(int(seed), int(n), _capi.dptr(out))
    return out


def quantise_rgba8(rgb):
    """Image.write's toByte (Image.fs:36)."""
    rgb = _capi.as_f64(rgb)
    n = rgb.size // 3
    out = np.zeros((n, 4), dtype=np.uint8)
    rc = hip_lib().ft_quantise_rgba8(_capi.dptr(rgb), n, out.ctypes.data_as(C.POINTER(C.c_uint8)))
    if rc < 0:
        raise FtError(rc)
    return out.reshape(rgb.shape[:-1] + (4,))


def write_png(path, rgba):
    rgba = np.ascontiguousarray(rgba, dtype=np.uint8)
    h, w = rgba.shape[:2]
    rc = host_lib().fth_write_png(os.fspath(path).encode(), rgba.ctypes.data_as(C.POINTER(C.c_uint8)), w, h)
    if rc < 0:
        raise FtError(rc, "write_png")


def load_image(path):
    """Image.Load<Rgb24> for local PNG / PPM files (Textures/Image.fs:21-26): (height, width, 3) uint8, row 0 = top."""
    lib = host_lib()
    w, h = C.c_int32(), C.c_int32()
    err = C.create_string_buffer(512)
    n = lib.fth_load_image(os.fspath(path).encode(), C.byref(w), C.byref(h), None, 0, err, 512)
    if n < 0:
        raise FtError(int(n), err.value.decode())
    out = np.empty(n, dtype=np.uint8)
    lib.fth_load_image(os.fspath(path).encode(), C.byref(w), C.byref(h), out.ctypes.data_as(C.POINTER(C.c_uint8)), n, err, 512)
    return out.reshape(h.value, w.value, 3)


class PinnedArray:
    """A numpy array over page-locked host memory from ft_host_alloc: frames copied into it arrive by one DMA at link rate
    (`with ft.PinnedArray((h, w, 3)) as frame: ctx.render(..., out=frame)`)."""

    def __init__(self, shape, dtype=np.float64):
        self.nbytes = int(np.prod(shape)) * np.dtype(dtype).itemsize
        self._p = hip_lib().ft_host_alloc(self.nbytes)
        if not self._p:
            raise MemoryError(f"ft_host_alloc({self.nbytes}) failed")
        self.array = np.frombuffer((C.c_char * self.nbytes).from_address(self._p), dtype=dtype).reshape(shape)

    def __enter__(self):
        return self.array

    def __exit__(self, *exc):
        self.close()

    def close(self):
        if self._p:
            self.array = None
            hip_lib().ft_host_free(self._p)
            self._p = None


class Context(SceneBuilder):
    """One ft_context on one MI355X.  `host_only=True` gives the test hook that can build and flatten
    scenes but never renders."""

    def __init__(self, device=0, host_only=False):
        lib = hip_lib()
        h = C.c_void_p()
        if host_only:
            rc = lib.ft_create_host_only(C.byref(h))
        else:
            ids = [int(d) for d in device] if isinstance(device, (list, tuple)) else [int(device)]
            dev = (C.c_int32 * len(ids))(*ids)                # several ids: one context that tiles every frame over those GPUs
            rc = lib.ft_create(dev, len(ids), C.byref(h))
        if rc < 0:
            raise FtError(rc, "ft_create: no usable HIP device (the HIP path has no CPU fallback)" if rc == -2 else "ft_create")
        self.device = device
        super().__init__(lib, "ft_", h)

    def close(self):
        if self._ctx:
            self._lib.ft_destroy(self._ctx)
            self._ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_option(self, key, value):
        self._check(self._lib.ft_set_option(self._ctx, key.encode(), int(value)))

    def render(self, camera, res_h, res_v, spp, jitter, max_depth=MAX_DEPTH, seed=DEFAULT_SEED, tiles=None, out=None, fetch=True):
        """Program.fs:54-64 on the GPU.  Returns (rgb[res_v, res_h, 3] float64, stats dict).  With
        fetch=False the frame stays in HBM (returns (None, stats)); `fetch_frame` copies it out later."""
        jitter = np.zeros((1, 2)) if spp == 0 else _capi.as_f64(jitter, (spp, 2))   # spp == 0: `samples corner` (Image.fs:125-150)
        if fetch and out is None:
            out = np.zeros((res_v, res_h, 3))
        rects, n_rects = _capi.make_rects(tiles)
        st = _capi.ft_stats()
        rc = self._lib.ft_render(self._ctx, C.byref(camera), res_h, res_v, spp, _capi.dptr(jitter), max_depth, int(seed), rects, n_rects,
                                 _capi.dptr(out) if fetch else None, C.byref(st))
        self._check(rc)
        return (out if fetch else None), st.as_dict()

    def fetch_frame(self, out):
        self._check(self._lib.ft_fetch_frame(self._ctx, _capi.dptr(out)))
        return out

    def render_rgba8(self, camera, res_h, res_v, spp, jitter, max_depth=MAX_DEPTH, seed=DEFAULT_SEED, tiles=None, out=None, fetch=True):
        """ft_render_rgba8: the frame as Image.write consumes it (Image.fs:35-44), quantised on the device.  Returns
        (rgba[res_v, res_h, 4] uint8, stats dict); with fetch=False the bytes stay in HBM until `fetch_frame_rgba8`."""
        jitter = np.zeros((1, 2)) if spp == 0 else _capi.as_f64(jitter, (spp, 2))
        if fetch and out is None:
            out = np.zeros((res_v, res_h, 4), dtype=np.uint8)
        rects, n_rects = _capi.make_rects(tiles)
        st = _capi.ft_stats()
        rc = self._lib.ft_render_rgba8(self._ctx, C.byref(camera), res_h, res_v, spp, _capi.dptr(jitter), max_depth, int(seed), rects, n_rects,
                                       out.ctypes.data_as(C.POINTER(C.c_uint8)) if fetch else None, C.byref(st))
        self._check(rc)
        return (out if fetch else None), st.as_dict()

    def fetch_frame_rgba8(self, out):
        self._check(self._lib.ft_fetch_frame_rgba8(self._ctx, out.ctypes.data_as(C.POINTER(C.c_uint8))))
        return out

    def render_enqueue(self, camera, res_h, res_v, spp, jitter, max_depth=MAX_DEPTH, seed=DEFAULT_SEED, tiles=None, rgba8=False, out=None):
        """ft_render_enqueue[_rgba8]: queue a frame and return; `wait()` retires what is queued.  With `out` (a PinnedArray's array, or any
        C-contiguous array of the frame's shape that outlives the frame) the copy to the host is queued behind the frame: ft_render_enqueue_into."""
        jitter = np.zeros((1, 2)) if spp == 0 else _capi.as_f64(jitter, (spp, 2))
        rects, n_rects = _capi.make_rects(tiles)
        if out is not None:
            assert out.flags["C_CONTIGUOUS"] and out.nbytes == res_h * res_v * (4 if rgba8 else 24)
            self._check(self._lib.ft_render_enqueue_into(self._ctx, C.byref(camera), res_h, res_v, spp, _capi.dptr(jitter), max_depth, int(seed), rects, n_rects,
                                                         1 if rgba8 else 0, out.ctypes.data_as(C.c_void_p)))
            return
        fn = self._lib.ft_render_enqueue_rgba8 if rgba8 else self._lib.ft_render_enqueue
        self._check(fn(self._ctx, C.byref(camera), res_h, res_v, spp, _capi.dptr(jitter), max_depth, int(seed), rects, n_rects))

    def wait(self):
        """ft_render_wait: statistics of the last queued frame; kernel_times() then holds the sums over all frames since the previous wait."""
        st = _capi.ft_stats()
        self._check(self._lib.ft_render_wait(self._ctx, C.byref(st)))
        return st.as_dict()

    def kernel_times(self):
        ms = np.zeros(5)
        n = np.zeros(5, dtype=np.int32)
        self._check(self._lib.ft_get_kernel_times(self._ctx, _capi.dptr(ms), n.ctypes.data_as(_capi.c_int32_p)))
        names = ["other", "closest", "shade", "resolve", "primary"]   # "other": the fill, k_classify (and k_resolve unless "timing" = 2); "shade": k_bounce, all levels; "closest": unused
        return {k: {"ms": float(ms[i]), "launches": int(n[i])} for i, k in enumerate(names)}

    def closest(self, origins, dirs):
        """Scene.intersectScene (Scene.fs:118) for explicit rays, through the device path."""
        o = _capi.as_f64(origins).reshape(-1, 3)
        d = _capi.as_f64(dirs).reshape(-1, 3)
        n = o.shape[0]
        hit = np.zeros(n, dtype=np.int32)
        t, p, nr, col = np.zeros(n), np.zeros((n, 3)), np.zeros((n, 3)), np.zeros((n, 3))
        self._check(self._lib.ft_debug_closest(self._ctx, _capi.dptr(o), _capi.dptr(d), n, hit.ctypes.data_as(_capi.c_int32_p),
                                               _capi.dptr(t), _capi.dptr(p), _capi.dptr(nr), _capi.dptr(col)))
        return hit, t, p, nr, col

    def blocked(self, origins, dirs, max_dist):
        """Scene.lightIsBocked (Scene.fs:119-121) for explicit rays, through the device path."""
        o = _capi.as_f64(origins).reshape(-1, 3)
        d = _capi.as_f64(dirs).reshape(-1, 3)
        m = _capi.as_f64(max_dist).reshape(-1)
        out = np.zeros(o.shape[0], dtype=np.int32)
        self._check(self._lib.ft_debug_blocked(self._ctx, _capi.dptr(o), _capi.dptr(d), _capi.dptr(m), o.shape[0], out.ctypes.data_as(_capi.c_int32_p)))
        return out

    def colour_for_ray(self, origins, dirs, max_depth=MAX_DEPTH):
        """Shading.getColourForRay (Shading.fs:131-139) for explicit rays, through the device path."""
        o = _capi.as_f64(origins).reshape(-1, 3)
        d = _capi.as_f64(dirs).reshape(-1, 3)
        rgb = np.zeros((o.shape[0], 3))
        self._check(self._lib.ft_debug_colour(self._ctx, _capi.dptr(o), _capi.dptr(d), o.shape[0], max_depth, _capi.dptr(rgb)))
        return rgb

    def devices(self):
        """The device ordinals behind this context (several for a multi-device context)."""
        out = (C.c_int32 * 64)()
        n = self._check(self._lib.ft_debug_devices(self._ctx, out, 64))
        return list(out[:n])

    def commit_times(self):
        """ft_get_commit_times of the last commit: host flatten, device BVH builds, uploads (ms) and the tallest device-built tree."""
        ms = np.zeros(4)
        self._check(self._lib.ft_get_commit_times(self._ctx, _capi.dptr(ms)))
        return {"flatten_ms": float(ms[0]), "device_bvh_ms": float(ms[1]), "upload_ms": float(ms[2]), "device_bvh_height": int(ms[3])}

    def scene_info(self):
        out = (C.c_int64 * 12)()
        self._check(self._lib.ft_debug_scene_info(self._ctx, out))
        keys = ["leaves", "program_words", "meshes", "bsp_nodes", "bsp_leaves", "triangles", "csg_capacity", "stack_capacity", "items", "bounded_items",
                "unbounded", "face_directions"]
        return dict(zip(keys, list(out)))


def rays_handled_by(kernel, st):
    """Rays one frame's launches of `kernel` trace, from the frame's ft_stats (bench.py's roofline line)."""
    generated = st["rays_primary"] - st["rays_primary_culled"]
    if kernel == "primary":                                         # fused bounce 0: every generated primary ray + the shadow rays of its hits
        return generated + st["rays_shadow_primary"]
    # k_bounce ("shade"): the reflection rays of every level >= 1 and the shadow rays of their hits
    return st["rays_reflect"] + st["rays_shadow"] - st["rays_shadow_primary"]


def debug_slice(p0, n, tri):
    """Triangle.slice as implemented by the product's BSP builder."""
    above, below = np.zeros(18), np.zeros(18)
    na, nb = C.c_int32(), C.c_int32()
    tri = _capi.as_f64(tri, (9,))
    rc = hip_lib().ft_debug_slice(_capi.dptr(_capi.as_f64(p0)), _capi.dptr(_capi.as_f64(n)), _capi.dptr(tri), _capi.dptr(above), C.byref(na),
                                  _capi.dptr(below), C.byref(nb))
    if rc < 0:
        raise FtError(rc, "slice")
    return above[:9 * na.value].reshape(-1, 3, 3), below[:9 * nb.value].reshape(-1, 3, 3)
