"""ctypes bindings of the C ABI (include/functracer_hip.h, functracer_amd/host/host_api.h).

Plumbing only: structs, prototypes and a generic `SceneBuilder` that drives any library exporting
the builder half of the ABI under a symbol prefix (`ft_` for the HIP product; the test-only CPU
oracle exports the same shape under `fto_`, bound in oracle/ft_oracle_py.py).
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_DIR = os.path.join(_HERE, "lib")

FT_OK = 0
STATUS_NAMES = {0: "FT_OK", -1: "FT_ERR_INVALID", -2: "FT_ERR_NO_DEVICE", -3: "FT_ERR_HIP", -4: "FT_ERR_UNSUPPORTED",
                -5: "FT_ERR_STATE", -6: "FT_ERR_OVERFLOW", -7: "FT_ERR_BUILD"}

# Scene.Primitive (Scene.fs:10-17) and Csg ops (Scene.fs:36-39)
CIRCLE, SQUARE, CUBE, SPHERE, PLANE, CONE, SOLID_CYLINDER, CYLINDER = range(8)
UNION, INTERSECT, SUBTRACT, EXCLUDE = range(4)
TRANSLATE, SCALE, ROTATE = range(3)

c_double_p = C.POINTER(C.c_double)
c_int32_p = C.POINTER(C.c_int32)


class FtError(RuntimeError):
    def __init__(self, status, message=""):
        self.status = status
        super().__init__(f"{STATUS_NAMES.get(status, status)}: {message}")


class ft_transform(C.Structure):
    _fields_ = [("kind", C.c_int32), ("_pad", C.c_int32), ("v", C.c_double * 3), ("angle", C.c_double)]


class ft_material(C.Structure):
    _fields_ = [("colour", C.c_double * 3), ("roughness", C.c_double), ("reflectance", C.c_double),
                ("shineyness", C.c_double), ("apply_lighting", C.c_int32), ("_pad", C.c_int32)]


class ft_camera(C.Structure):
    _fields_ = [("o", C.c_double * 3), ("look_at", C.c_double * 3), ("up", C.c_double * 3), ("fov_y", C.c_double),
                ("aspect_ratio", C.c_double), ("has_focus", C.c_int32), ("_pad", C.c_int32),
                ("focal_length", C.c_double), ("aperture_angular_size", C.c_double)]


class ft_rect(C.Structure):
    _fields_ = [("x0", C.c_int32), ("y0", C.c_int32), ("w", C.c_int32), ("h", C.c_int32)]


class ft_stats(C.Structure):
    _fields_ = [("rays_primary", C.c_uint64), ("rays_shadow", C.c_uint64), ("rays_reflect", C.c_uint64),
                ("rays_traced", C.c_uint64), ("rays_reference_equivalent", C.c_double), ("hits_primary", C.c_uint64),
                ("csg_overflow", C.c_uint64), ("kernel_ms", C.c_double), ("wall_ms", C.c_double),
                ("trace_kernel_ms", C.c_double), ("algorithmic_bytes", C.c_uint64), ("n_launches", C.c_int32),
                ("n_chunks", C.c_int32), ("hits_total", C.c_uint64), ("algorithmic_bytes_closest", C.c_uint64),
                ("algorithmic_bytes_shade", C.c_uint64), ("rays_tail", C.c_uint64), ("rays_primary_culled", C.c_uint64),
                ("algorithmic_bytes_primary", C.c_uint64), ("rays_shadow_primary", C.c_uint64), ("rays_reflect_primary", C.c_uint64)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


class fth_options(C.Structure):
    _fields_ = [("camera", ft_camera), ("res_h", C.c_int32), ("res_v", C.c_int32), ("samples", C.c_int32), ("corner", C.c_int32)]


BUILDER_SIGNATURES = [
    ("sg_primitive", C.c_int32, [C.c_void_p, C.c_int32]),
    ("sg_triangle", C.c_int32, [C.c_void_p, c_double_p]),
    ("sg_bsp_mesh", C.c_int32, [C.c_void_p, C.c_int32, c_double_p, C.c_int64]),
    ("sg_transform", C.c_int32, [C.c_void_p, C.POINTER(ft_transform), C.c_int32, C.c_int32]),
    ("sg_material", C.c_int32, [C.c_void_p, C.POINTER(ft_material), C.c_int32]),
    ("sg_hue_shift", C.c_int32, [C.c_void_p, C.c_double, C.c_int32]),
    ("sg_ignore_light", C.c_int32, [C.c_void_p, C.c_int32]),
    ("sg_group", C.c_int32, [C.c_void_p, c_int32_p, C.c_int32]),
    ("sg_csg", C.c_int32, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32]),
    ("sg_texture_grid", C.c_int32, [C.c_void_p, c_double_p, c_double_p, c_double_p, C.c_int32, C.c_int32]),
    ("sg_texture_image", C.c_int32, [C.c_void_p, C.POINTER(C.c_uint8), C.c_int32, C.c_int32, c_double_p, C.c_int32, C.c_int32]),
    ("scene_clear", C.c_int32, [C.c_void_p]),
    ("scene_set_objects", C.c_int32, [C.c_void_p, C.c_int32]),
    ("scene_add_directional", C.c_int32, [C.c_void_p, c_double_p, c_double_p]),
    ("scene_add_soft_directional", C.c_int32, [C.c_void_p, c_double_p, C.c_int32, C.c_double, c_double_p]),
    ("scene_add_positional", C.c_int32, [C.c_void_p, c_double_p, c_double_p, c_double_p]),
    ("scene_commit", C.c_int32, [C.c_void_p]),
]


class fth_builder(C.Structure):
    _fields_ = [(name, C.CFUNCTYPE(res, *args)) for name, res, args in BUILDER_SIGNATURES]


def dptr(a):
    return a.ctypes.data_as(c_double_p)


def as_f64(a, shape=None):
    a = np.ascontiguousarray(a, dtype=np.float64)
    if shape is not None:
        a = a.reshape(shape)
    return a


def vec3(v):
    return (C.c_double * 3)(*[float(x) for x in v])


def load_library(path):
    if not os.path.exists(path):
        raise FtError(-2, f"{path} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` (there is no Python/CPU fallback)")
    return C.CDLL(path, mode=C.RTLD_GLOBAL)


def bind_builder(lib, prefix):
    """Set prototypes for the builder functions `prefix`sg_* / `prefix`scene_* and return an fth_builder table."""
    table = fth_builder()
    for name, res, args in BUILDER_SIGNATURES:
        fn = getattr(lib, prefix + name)
        fn.restype = res
        fn.argtypes = args
        setattr(table, name, C.cast(fn, C.CFUNCTYPE(res, *args)))
    return table


class SceneBuilder:
    """Mirror of the Scene.fs constructors over a C context (`lib`, `prefix`, `ctx` pointer)."""

    def __init__(self, lib, prefix, ctx):
        self._lib, self._prefix, self._ctx = lib, prefix, ctx
        self.table = bind_builder(lib, prefix)
        last = getattr(lib, prefix + "last_error")
        last.restype = C.c_char_p
        last.argtypes = [C.c_void_p]
        self._last_error = last

    def _f(self, name):
        return getattr(self._lib, self._prefix + name)

    def last_error(self):
        return (self._last_error(self._ctx) or b"").decode()

    def _check(self, rc):
        if rc < 0:
            raise FtError(rc, self.last_error())
        return rc

    # Scene.Primitive ------------------------------------------------------------------
    def primitive(self, kind):
        return self._check(self._f("sg_primitive")(self._ctx, kind))

    def triangle(self, a, b, c):
        v = as_f64([a, b, c], (9,))
        return self._check(self._f("sg_triangle")(self._ctx, dptr(v)))

    def bsp_mesh(self, depth, triangles):
        t = as_f64(triangles).reshape(-1, 9)
        return self._check(self._f("sg_bsp_mesh")(self._ctx, int(depth), dptr(t), t.shape[0]))

    # Scene.SceneFunction --------------------------------------------------------------
    def transform(self, ops, child):
        """ops: list of ('translate', v) | ('scale', v) | ('rotate', axis, angle_rad); >1 entries = Composed."""
        arr = (ft_transform * len(ops))()
        for i, op in enumerate(ops):
            kind = {"translate": TRANSLATE, "scale": SCALE, "rotate": ROTATE}[op[0]]
            arr[i].kind = kind
            v = op[1]
            if kind == SCALE and np.isscalar(v):
                v = (v, v, v)
            arr[i].v = (C.c_double * 3)(*[float(x) for x in v])
            arr[i].angle = float(op[2]) if kind == ROTATE else 0.0
        return self._check(self._f("sg_transform")(self._ctx, arr, len(ops), child))

    def translate(self, v, child):
        return self.transform([("translate", v)], child)

    def scale(self, v, child):
        return self.transform([("scale", v)], child)

    def rotate(self, axis, angle_rad, child):
        return self.transform([("rotate", axis, angle_rad)], child)

    def material(self, child, colour=(1, 1, 1), roughness=0.0, reflectance=0.0, shineyness=0.0, apply_lighting=True):
        m = ft_material()
        m.colour = (C.c_double * 3)(*[float(x) for x in colour])
        m.roughness, m.reflectance, m.shineyness = float(roughness), float(reflectance), float(shineyness)
        m.apply_lighting = 1 if apply_lighting else 0
        return self._check(self._f("sg_material")(self._ctx, C.byref(m), child))

    def hue_shift(self, angle, child):
        return self._check(self._f("sg_hue_shift")(self._ctx, float(angle), child))

    def ignore_light(self, child):
        return self._check(self._f("sg_ignore_light")(self._ctx, child))

    def group(self, children):
        arr = (C.c_int32 * len(children))(*children)
        return self._check(self._f("sg_group")(self._ctx, arr, len(children)))

    def csg(self, op, a, b):
        return self._check(self._f("sg_csg")(self._ctx, op, a, b))

    def union(self, a, b):
        return self.csg(UNION, a, b)

    def intersect(self, a, b):
        return self.csg(INTERSECT, a, b)

    def subtract(self, a, b):
        return self.csg(SUBTRACT, a, b)

    def exclude(self, a, b):
        return self.csg(EXCLUDE, a, b)

    def texture_grid(self, colour_a, colour_b, uv_ops, child):
        ops = as_f64(uv_ops).reshape(-1, 3) if len(uv_ops) else np.zeros((0, 3))
        return self._check(self._f("sg_texture_grid")(self._ctx, vec3(colour_a), vec3(colour_b), dptr(ops), ops.shape[0], child))

    def texture_image(self, pixels, uv_ops, child):
        """pixels: (height, width, 3) uint8, row 0 = top (image.SavePixelData() of an Rgb24 image)."""
        px = np.ascontiguousarray(pixels, dtype=np.uint8)
        assert px.ndim == 3 and px.shape[2] == 3
        ops = as_f64(uv_ops).reshape(-1, 3) if len(uv_ops) else np.zeros((0, 3))
        return self._check(self._f("sg_texture_image")(self._ctx, px.ctypes.data_as(C.POINTER(C.c_uint8)), px.shape[1], px.shape[0],
                                                       dptr(ops), ops.shape[0], child))

    # Scene.Scene / Light --------------------------------------------------------------
    def clear(self):
        self._check(self._f("scene_clear")(self._ctx))

    def set_objects(self, root):
        self._check(self._f("scene_set_objects")(self._ctx, root))

    def add_directional(self, direction, colour):
        self._check(self._f("scene_add_directional")(self._ctx, vec3(direction), vec3(colour)))

    def add_soft_directional(self, direction, samples, scatter_rad, colour):
        self._check(self._f("scene_add_soft_directional")(self._ctx, vec3(direction), int(samples), float(scatter_rad), vec3(colour)))

    def add_positional(self, position, falloff, colour):
        self._check(self._f("scene_add_positional")(self._ctx, vec3(position), vec3(falloff), vec3(colour)))

    def commit(self):
        self._check(self._f("scene_commit")(self._ctx))


def make_camera(o, look_at, up, fov_y_rad, aspect_ratio=1.0):
    cam = ft_camera()
    cam.o = (C.c_double * 3)(*[float(x) for x in o])
    cam.look_at = (C.c_double * 3)(*[float(x) for x in look_at])
    cam.up = (C.c_double * 3)(*[float(x) for x in up])
    cam.fov_y, cam.aspect_ratio = float(fov_y_rad), float(aspect_ratio)
    return cam


def make_rects(tiles):
    if tiles is None:
        return None, 0
    arr = (ft_rect * len(tiles))()
    for i, (x0, y0, w, h) in enumerate(tiles):
        arr[i].x0, arr[i].y0, arr[i].w, arr[i].h = int(x0), int(y0), int(w), int(h)
    return arr, len(tiles)
