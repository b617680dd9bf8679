"""ctypes binding of oracle/libft_oracle.so — TEST INFRASTRUCTURE ONLY.

Importable only from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.  The
product package (functracer_amd/) never imports this module.
"""
import ctypes as C
import os

import numpy as np

from functracer_amd import _capi

_HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_LIB = os.path.join(_HERE, "libft_oracle.so")


class fto_stats(C.Structure):
    _fields_ = [("rays_primary", C.c_uint64), ("rays_shadow", C.c_uint64), ("rays_reflect", C.c_uint64), ("rays_traced", C.c_uint64),
                ("wall_ms", C.c_double), ("threads", C.c_int32), ("_pad", C.c_int32)]


_lib = None
dp, ip = _capi.c_double_p, _capi.c_int32_p


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(ORACLE_LIB):
            raise RuntimeError(f"{ORACLE_LIB} missing: run `make -C oracle`")
        L = C.CDLL(ORACLE_LIB)
        L.fto_create.argtypes = [C.POINTER(C.c_void_p)]
        L.fto_destroy.argtypes = [C.c_void_p]
        L.fto_destroy.restype = None
        L.fto_render.argtypes = [C.c_void_p, C.POINTER(_capi.ft_camera), C.c_int32, C.c_int32, C.c_int32, dp, C.c_int32, C.c_uint64,
                                 C.POINTER(_capi.ft_rect), C.c_int32, dp, C.c_int32, C.POINTER(fto_stats)]
        L.fto_closest.argtypes = [C.c_void_p, dp, dp, C.c_int64, ip, dp, dp, dp, dp]
        L.fto_all_hits.argtypes = [C.c_void_p, dp, dp, C.c_int64, C.c_int32, ip, dp, dp, dp]
        L.fto_blocked.argtypes = [C.c_void_p, dp, dp, dp, C.c_int64, ip]
        L.fto_colour_for_ray.argtypes = [C.c_void_p, dp, dp, C.c_int64, C.c_int32, dp]
        L.fto_ray_through_pixel.argtypes = [C.POINTER(_capi.ft_camera), C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_double, C.c_double, dp, dp]
        L.fto_image_plane.argtypes = [C.POINTER(_capi.ft_camera), C.c_int32, C.c_int32, dp]
        L.fto_aabb_intersects.argtypes = [dp, dp, dp, dp]
        L.fto_slice.argtypes = [dp, dp, dp, dp, ip, dp, ip]
        L.fto_bsp_stats.argtypes = [dp, C.c_int64, C.c_int32, C.POINTER(C.c_int64)]
        L.fto_quadratic.argtypes = [C.c_double, C.c_double, C.c_double, dp]
        L.fto_quantise_rgba8.argtypes = [dp, C.c_int64, C.POINTER(C.c_uint8)]
        _lib = L
    return _lib


class Oracle(_capi.SceneBuilder):
    """CPU restatement of the reference behind the same builder shape as functracer_amd.Context."""

    def __init__(self):
        L = lib()
        h = C.c_void_p()
        rc = L.fto_create(C.byref(h))
        if rc < 0:
            raise _capi.FtError(rc, "fto_create")
        super().__init__(L, "fto_", h)

    def close(self):
        if self._ctx:
            self._lib.fto_destroy(self._ctx)
            self._ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def render(self, camera, res_h, res_v, spp, jitter, max_depth=8, seed=20260104, tiles=None, threads=0, out=None):   # seed: functracer_amd.DEFAULT_SEED, so both sides draw the same streams by default
        jitter = np.zeros((1, 2)) if spp == 0 else _capi.as_f64(jitter, (spp, 2))   # spp == 0: `samples corner` (Image.fs:125-150)
        if out is None:
            out = np.zeros((res_v, res_h, 3))
        rects, n_rects = _capi.make_rects(tiles)
        st = fto_stats()
        self._check(self._lib.fto_render(self._ctx, C.byref(camera), res_h, res_v, spp, _capi.dptr(jitter), max_depth, int(seed), rects, n_rects,
                                         _capi.dptr(out), int(threads), C.byref(st)))
        return out, {k: getattr(st, k) for k, _ in st._fields_ if k != "_pad"}

    def closest(self, origins, dirs):
        o = _capi.as_f64(origins).reshape(-1, 3)
        d = _capi.as_f64(dirs).reshape(-1, 3)
        n = o.shape[0]
        hit = np.zeros(n, dtype=np.int32)
        t, p, nr, col = np.zeros(n), np.zeros((n, 3)), np.zeros((n, 3)), np.zeros((n, 3))
        self._check(self._lib.fto_closest(self._ctx, _capi.dptr(o), _capi.dptr(d), n, hit.ctypes.data_as(ip), _capi.dptr(t), _capi.dptr(p), _capi.dptr(nr), _capi.dptr(col)))
        return hit, t, p, nr, col

    def all_hits(self, origins, dirs, cap=16):
        o = _capi.as_f64(origins).reshape(-1, 3)
        d = _capi.as_f64(dirs).reshape(-1, 3)
        n = o.shape[0]
        counts = np.zeros(n, dtype=np.int32)
        t, p, nr = np.zeros((n, cap)), np.zeros((n, cap, 3)), np.zeros((n, cap, 3))
        self._check(self._lib.fto_all_hits(self._ctx, _capi.dptr(o), _capi.dptr(d), n, cap, counts.ctypes.data_as(ip), _capi.dptr(t), _capi.dptr(p), _capi.dptr(nr)))
        return counts, t, p, nr

    def blocked(self, origins, dirs, max_dist):
        o = _capi.as_f64(origins).reshape(-1, 3)
        d = _capi.as_f64(dirs).reshape(-1, 3)
        m = _capi.as_f64(max_dist).reshape(-1)
        out = np.zeros(o.shape[0], dtype=np.int32)
        self._check(self._lib.fto_blocked(self._ctx, _capi.dptr(o), _capi.dptr(d), _capi.dptr(m), o.shape[0], out.ctypes.data_as(ip)))
        return out

    def colour_for_ray(self, origins, dirs, max_depth=8):
        o = _capi.as_f64(origins).reshape(-1, 3)
        d = _capi.as_f64(dirs).reshape(-1, 3)
        rgb = np.zeros((o.shape[0], 3))
        self._check(self._lib.fto_colour_for_ray(self._ctx, _capi.dptr(o), _capi.dptr(d), o.shape[0], max_depth, _capi.dptr(rgb)))
        return rgb


def ray_through_pixel(camera, res_h, res_v, px, py, jx=0.0, jy=0.0):
    o, d = np.zeros(3), np.zeros(3)
    lib().fto_ray_through_pixel(C.byref(camera), res_h, res_v, px, py, jx, jy, _capi.dptr(o), _capi.dptr(d))
    return o, d


def image_plane(camera, res_h, res_v):
    out = np.zeros(13)
    lib().fto_image_plane(C.byref(camera), res_h, res_v, _capi.dptr(out))
    return {"pixel_width": out[0], "pixel_height": out[1], "top_left": (out[2], out[3]), "i": out[4:7].copy(), "j": out[7:10].copy(), "k": out[10:13].copy()}


def aabb_intersects(bmin, bmax, o, d):
    f = _capi.as_f64
    return bool(lib().fto_aabb_intersects(_capi.dptr(f(bmin)), _capi.dptr(f(bmax)), _capi.dptr(f(o)), _capi.dptr(f(d))))


def slice_triangle(p0, n, tri):
    f = _capi.as_f64
    above, below = np.zeros(18), np.zeros(18)
    na, nb = C.c_int32(), C.c_int32()
    rc = lib().fto_slice(_capi.dptr(f(p0)), _capi.dptr(f(n)), _capi.dptr(f(tri, (9,))), _capi.dptr(above), C.byref(na), _capi.dptr(below), C.byref(nb))
    if rc < 0:
        raise _capi.FtError(rc, "slice")
    return above[:9 * na.value].reshape(-1, 3, 3), below[:9 * nb.value].reshape(-1, 3, 3)


def bsp_stats(tris, depth):
    t = _capi.as_f64(tris).reshape(-1, 9)
    out = (C.c_int64 * 3)()
    rc = lib().fto_bsp_stats(_capi.dptr(t), t.shape[0], depth, out)
    if rc < 0:
        raise _capi.FtError(rc, "bsp")
    return {"max_depth": out[0], "leaves": out[1], "leaf_triangles": out[2]}


def quadratic(a, b, c):
    r = np.zeros(2)
    n = lib().fto_quadratic(a, b, c, _capi.dptr(r))
    return list(r[:n])


def quantise_rgba8(rgb):
    rgb = _capi.as_f64(rgb)
    n = rgb.size // 3
    out = np.zeros((n, 4), dtype=np.uint8)
    lib().fto_quantise_rgba8(_capi.dptr(rgb), n, out.ctypes.data_as(C.POINTER(C.c_uint8)))
    return out.reshape(rgb.shape[:-1] + (4,))
