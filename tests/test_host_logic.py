"""Host-side logic that needs no GPU: the .scene / PLY loaders, scene flattening and the product's
own BSP builder (through the host-only test hook of the C ABI), and the exported ABI surface."""
import ctypes as C
import math
import os
import re

import numpy as np
import pytest

import functracer_amd as ft
from oracle import ft_oracle_py as O

from . import helpers as H


def test_colour_literals(golden):                             # FuncTracer.Tests/Parser/Colour.fs:17-30
    for case in golden["reference_tests"]["colour"]["cases"]:
        assert list(ft.parse_colour(case["text"])) == case["expect"]
    with pytest.raises(ValueError):
        ft.parse_colour("red")


def test_scene_parser_options_and_counts():
    p = ft.parse_scene_file(H.scene_path("night-house-det"))
    assert p.resolution == (1920, 1080) and p.samples == 16 and not p.corner
    assert p.n_objects == 5 and p.n_lights == 3
    assert list(p.camera.o) == [15.0, 11.0, -20.0] and p.camera.fov_y == pytest.approx(np.pi / 3)
    d = ft.parse_scene("(material diffuse 1 reflectance 0 shineyness 0 sphere)\n")        # defaults: Scene.fs:61-65
    assert d.resolution == (400, 400) and d.samples == 8 and d.camera.fov_y == pytest.approx(50 * np.pi / 180)
    assert ft.parse_scene("samples corner\nsphere\n").corner


def test_scene_parser_errors_and_order():
    with pytest.raises(ValueError):
        ft.parse_scene("(material diffuse 1 spher)\n")
    with pytest.raises(ValueError):                           # lights before objects: sections are ordered (SceneParser.fs:357)
        ft.parse_scene("directional dir (0,-1,0) colour 1\nsphere\n")
    with pytest.raises(ValueError):
        ft.parse_scene('bspMesh 0 "does-not-exist.ply"\n')


def test_repeat_yields_n_plus_one_copies():                   # SceneParser.fs:242-251
    p = ft.parse_scene("(repeat 8 translate (-0.4,0,-1) (translate (-2,0,-5) sphere))\n")
    ctx = ft.Context(host_only=True)
    p.lower(ctx)
    assert ctx.scene_info()["leaves"] == 9
    orc = O.Oracle()
    p.lower(orc)
    # copy k (1-based) sits at (-2,0,-5) + k * (-0.4,0,-1); the untranslated original is not part of the group
    for k in range(1, 10):
        c = np.array([-2 - 0.4 * k, 0.0, -5 - 1.0 * k])
        hit, t, *_ = orc.closest([c + [0, 5, 0]], [[0, -1, 0]])
        assert hit[0] == 1 and t[0] == pytest.approx(4.0)
    hit, *_ = orc.closest([[-2, 5, -5]], [[0, -1, 0]])
    assert hit[0] == 0


def test_composed_function_applies_first_listed_first():      # SceneParser.fs:235-239
    p = ft.parse_scene("((translate (1,0,0)) . (scale (2,2,2) ) sphere )\n")
    orc = O.Oracle()
    p.lower(orc)
    hit, t, pp, *_ = orc.closest([[2, 0, -9]], [[0, 0, 1]])  # scale(translate(sphere)): centre (2,0,0), radius 2
    assert hit[0] == 1 and pp[0][2] == pytest.approx(-2.0)


def test_ply_loader_layout():                                 # PlyParser.fs:20-61
    text = "ply\nformat ascii 1.0\ncomment x\nelement vertex 3\nproperty float x\nelement face 1\nproperty list uchar int vertex_indices\nend_header\n" \
           "0 0 0 1 0.5\n1 0 0 1 0.5\n0 1 0 1 0.5\n3 0 1 2\n"
    tris = ft.parse_ply(text)
    assert tris.shape == (1, 9) and list(tris[0]) == [0, 0, 0, 1, 0, 0, 0, 1, 0]
    with pytest.raises(ValueError):                           # 3 numbers per vertex is not the accepted layout (pipe5, PlyParser.fs:42-49)
        ft.parse_ply(text.replace(" 1 0.5", ""))
    with pytest.raises(ValueError):
        ft.parse_ply(text.replace("3 0 1 2", "4 0 1 2 0"))
    full = open(H.scene_path("meshes/bunny_synth_res4").replace(".scene", ".ply")).read()
    assert ft.parse_ply(full).shape == (980, 9)


def test_product_slice_matches_reference_vectors(golden):     # Triangle.Tests.fs:12-54 against the product's BSP builder
    g = golden["reference_tests"]["triangle_slice"]
    a, b, c = g["a"], g["b"], g["c"]
    ab, ac = g["ab_intercept"], g["ac_intercept"]
    for tri in ([a, b, c], [c, a, b], [b, c, a]):
        above, below = ft.debug_slice(g["plane_p0"], g["plane_n"], tri)
        assert np.array_equal(above, np.array([[a, ab, ac]]))
        assert np.array_equal(below, np.array([[ab, b, c], [c, ac, ab]]))
    above, below = ft.debug_slice(g["plane_p0"], g["plane_n"], g["wholly_above"])
    assert np.array_equal(above[0], np.array(g["wholly_above"])) and below.shape[0] == 0
    above, below = ft.debug_slice(g["plane_p0"], g["plane_n"], g["wholly_below"])
    assert np.array_equal(below[0], np.array(g["wholly_below"])) and above.shape[0] == 0


def test_product_slice_matches_oracle_on_random_triangles():
    rng = np.random.default_rng(4)
    for _ in range(300):
        tri = rng.normal(size=(3, 3))
        p0, n = rng.normal(size=3) * 0.3, np.eye(3)[rng.integers(0, 3)]
        ga, gb = ft.debug_slice(p0, n, tri)
        wa, wb = O.slice_triangle(p0, n, tri)
        assert np.array_equal(ga, wa) and np.array_equal(gb, wb)


@pytest.mark.parametrize("depth", [0, 1, 4, 12])
def test_product_bsp_build_matches_oracle_statistics(depth):  # BspMesh.fs:51-65, 78-86: two independent builders agree
    tris = ft.parse_ply(open(H.scene_path("meshes/bunny_synth_res4").replace(".scene", ".ply")).read())
    ctx = ft.Context(host_only=True)
    ctx.clear(); ctx.set_objects(ctx.group([ctx.bsp_mesh(depth, tris)])); ctx.commit()
    info = ctx.scene_info()
    st = O.bsp_stats(tris, depth)
    assert info["bsp_leaves"] == st["leaves"] and info["triangles"] == st["leaf_triangles"]
    assert info["bsp_nodes"] == st["leaves"] - 1
    assert info["stack_capacity"] == (st["max_depth"] + 1 if st["max_depth"] else 0)


def test_flatten_config_scenes():
    ctx = ft.Context(host_only=True)
    expect = {"hollow-sphere": (52, 8), "night-house-det": (2 + 1 + 9 + 4 + 3 + 1, 6), "bunny": (1, 0)}
    for name, (leaves, csg_cap) in expect.items():
        ft.parse_scene_file(H.scene_path(name)).lower(ctx)
        info = ctx.scene_info()
        assert info["leaves"] == leaves and info["csg_capacity"] == csg_cap, (name, info)


def test_host_only_context_never_renders():
    ctx = ft.Context(host_only=True)
    H.single_prim(ctx, "sphere")
    cam = ft.make_camera((0, 0, -5), (0, 0, 0), (0, 1, 0), 1.0)
    with pytest.raises(ft.FtError) as e:
        ctx.render(cam, 8, 8, 1, ft.jitter_pattern(1))
    assert e.value.status == -2                               # FT_ERR_NO_DEVICE: no CPU fallback
    with pytest.raises(ft.FtError):
        ctx.closest([[0, 0, -5]], [[0, 0, 1]])


def test_unsupported_surface_is_rejected_loudly():
    ctx = ft.Context(host_only=True)
    ctx.clear()
    ops = [(0, 0.5, 0.5)] * 14                               # fourteen nested texture functions: beyond the device path's table of 13
    ctx.set_objects(ctx.group([ctx.texture_grid((1, 0, 0), (0, 1, 0), ops, ctx.primitive(ft.SPHERE))]))
    with pytest.raises(ft.FtError) as e:
        ctx.commit()
    assert e.value.status == -4 and "texture" in str(e.value)
    with pytest.raises(ValueError) as e:                      # like the reference, a texture that cannot be loaded fails the parse
        ft.parse_scene('(texture image "no-such-file.png" sphere)\n')
    assert "cannot open image file" in str(e.value)
    for src, what in (('bspMesh 0 "meshes"\n', "cannot open mesh file"), ('(texture image "textures" sphere)\n', "cannot open image file")):
        with pytest.raises(ValueError) as e:                  # a directory where a file is expected is a parse error too (found by tools/fuzz_parsers.py)
            ft.parse_scene(src, base_dir=os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scenes"))
        assert what in str(e.value)
    with pytest.raises(ValueError) as e:                      # Textures/Image.fs:11-13 fetches URLs; there is no network here
        ft.parse_scene('(texture image "http://example.invalid/moon.jpg" sphere)\n')
    assert "URL" in str(e.value)
    with pytest.raises(ft.FtError):
        ctx.primitive(99)
    with pytest.raises(ft.FtError):
        ctx.csg(ft.UNION, 12345, 0)


def test_jitter_pattern_rule():                               # Jitter.fs:15-24: unit disc by rejection, reproducible stream
    a, b = ft.jitter_pattern(64), ft.jitter_pattern(64)
    assert np.array_equal(a, b) and (np.hypot(a[:, 0], a[:, 1]) <= 1.0).all()
    assert not np.array_equal(a, ft.jitter_pattern(64, seed=1))
    assert np.array_equal(ft.jitter_pattern(16), a[:16])


def test_pipeline_options_are_validated():
    """ft_set_option: the frame-pipeline tunables of round 3 (include/functracer_hip.h) accept their documented range and refuse anything else."""
    ctx = ft.Context(host_only=True)
    for key, good, bad in (("mains", (1, 2, 3), (0, 4, -1)), ("two_mains", (0, 1), ()), ("primary_reserve", (0, 64, 4096), (-1, 4097)),
                           ("window_cap", (64, 80 << 20), (0, 63)), ("window_hint", (0, 1), ()), ("classify_ahead", (0, 1), ()),
                           ("resolve_aside", (0, 1), ()), ("zero_fill_skip", (0, 1), ())):
        for v in good:
            ctx.set_option(key, v)
        for v in bad:
            with pytest.raises(ft.FtError):
                ctx.set_option(key, v)
    with pytest.raises(ft.FtError):
        ctx.set_option("no_such_option", 1)
    ctx.close()


def test_abi_exports_every_declared_symbol():
    hdr = open(os.path.join(H.ROOT, "include", "functracer_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    names = sorted(set(re.findall(r"\b(ft_[a-z0-9_]+)\s*\(", hdr)))
    assert len(names) >= 25
    lib = C.CDLL(ft.HIP_LIB)
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing
    lib.ft_abi_version.restype = C.c_int32
    assert lib.ft_abi_version() == 2
    # no CPU backend: zero devices is an error, not an oracle-backed context (SURVEY 8b proposed one; the product has none)
    h = C.c_void_p()
    assert lib.ft_create(None, 0, C.byref(h)) == -2 and not h.value


def test_fsharp_binding_matches_the_header():
    """INTEGRATION.md's F# shim cannot be compiled here (no .NET toolchain): at least keep it consistent with the C header.  Every
    DllImport names an exported function with the header's number of arguments, and every function of the render path is bound."""
    hdr = open(os.path.join(H.ROOT, "include", "functracer_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    protos = {}
    for m in re.finditer(r"\b(ft_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", hdr, flags=re.S):
        args = m.group(2).strip()
        protos[m.group(1)] = 0 if args in ("", "void") else len(args.split(","))
    doc = open(os.path.join(H.ROOT, "INTEGRATION.md")).read()
    fs = "\n".join(re.findall(r"```fsharp\n(.*?)```", doc, flags=re.S))
    imports = {}
    aliases = {}                                              # a second binding of an entry point under another F# name (EntryPoint = "...")
    for m in re.finditer(r"\[<DllImport\(Lib(?:,\s*EntryPoint\s*=\s*\"(ft_[a-z0-9_]+)\")?\)>\]\s*extern\s+\w+\s+(ft_[a-z0-9_]+)\s*\(([^)]*)\)", fs, flags=re.S):
        args = m.group(3).strip()
        n_args = 0 if args == "" else len(args.split(","))
        if m.group(1):
            aliases[m.group(2)] = m.group(1)
            assert protos[m.group(1)] == n_args, f"{m.group(2)} -> {m.group(1)}: {n_args} arguments in the F# binding, {protos[m.group(1)]} in the header"
        else:
            imports[m.group(2)] = n_args
    assert len(imports) >= 30
    for name, n in imports.items():
        assert name in protos, f"{name} is not declared in functracer_hip.h"
        assert protos[name] == n, f"{name}: {n} arguments in the F# binding, {protos[name]} in the header"
    test_hooks = {n for n in protos if n.startswith("ft_debug_")} | {"ft_create_host_only"}
    assert set(protos) - test_hooks <= set(imports), sorted(set(protos) - test_hooks - set(imports))
    for used in re.findall(r"\b(ft_[a-z0-9_]+)\s*\(", re.sub(r"\[<DllImport.*", "", fs)):     # every call in the shim's code is bound
        assert used in imports or used in aliases, used
    # the pieces the round-1 shim lacked: the Transform lowering, Focus, soft-light units
    assert "let rec toFt (t: Transform)" in fs and "Composed ts -> List.collect toFt ts" in fs
    assert "cam.focus with" in fs and "f.apetureAngularSize" in fs
    # struct layouts: same number of fields as the C structs
    def c_fields(name):
        body = re.search(r"typedef struct " + name + r"\s*\{(.*?)\}\s*" + name + ";", hdr, flags=re.S).group(1)
        n = 0
        for decl in body.split(";"):
            decl = decl.strip()
            if not decl:
                continue
            arr = re.search(r"\[(\d+)\]", decl)
            n += (int(arr.group(1)) if arr else 1) * len(decl.split(","))
        return n
    def fs_fields(name):
        body = re.search(r"type " + name + r"\s*=\s*\{(.*?)\}", fs, flags=re.S).group(1)
        return len(re.findall(r"\w+\s*:", body))
    for c_name, f_name in (("ft_transform", "FtTransform"), ("ft_material", "FtMaterial"), ("ft_camera", "FtCamera"), ("ft_rect", "FtRect"), ("ft_stats", "FtStats")):
        assert c_fields(c_name) == fs_fields(f_name), (c_name, c_fields(c_name), fs_fields(f_name))


def test_product_library_does_not_link_the_oracle():
    import subprocess
    out = subprocess.run(["ldd", ft.HIP_LIB], capture_output=True, text=True).stdout + subprocess.run(["nm", "-D", ft.HIP_LIB], capture_output=True, text=True).stdout
    assert "ft_oracle" not in out and "fto_" not in out


def test_png_writer_roundtrip(tmp_path):
    from PIL import Image
    rgba = (np.arange(7 * 5 * 4) % 251).astype(np.uint8).reshape(5, 7, 4)
    rgba[..., 3] = 255
    path = tmp_path / "x.png"
    ft.write_png(path, rgba)
    assert np.array_equal(np.asarray(Image.open(path).convert("RGBA")), rgba)


def test_image_loader_formats(tmp_path):                      # harness counterpart of Image.Load<Rgb24> (Textures/Image.fs:21-24)
    from PIL import Image
    rng = np.random.default_rng(5)
    rgb = rng.integers(0, 256, size=(13, 17, 3), dtype=np.uint8)
    Image.fromarray(rgb).save(tmp_path / "rgb.png")                                   # PIL picks adaptive filters per row
    assert np.array_equal(ft.load_image(tmp_path / "rgb.png"), rgb)
    rgba = np.concatenate([rgb, rng.integers(0, 256, size=(13, 17, 1), dtype=np.uint8)], -1)
    Image.fromarray(rgba).save(tmp_path / "rgba.png")                                 # alpha is dropped
    assert np.array_equal(ft.load_image(tmp_path / "rgba.png"), rgb)
    grey = rng.integers(0, 256, size=(9, 11), dtype=np.uint8)
    Image.fromarray(grey).save(tmp_path / "grey.png")
    assert np.array_equal(ft.load_image(tmp_path / "grey.png"), np.repeat(grey[..., None], 3, -1))
    pal = Image.fromarray(rgb).quantize(colors=7)                                     # 4-bit palette image
    pal.save(tmp_path / "pal.png", bits=4)
    assert np.array_equal(ft.load_image(tmp_path / "pal.png"), np.asarray(pal.convert("RGB")))
    bw = Image.fromarray(grey > 127)                                                  # 1-bit grey
    bw.save(tmp_path / "bw.png")
    assert np.array_equal(ft.load_image(tmp_path / "bw.png"), np.repeat((np.asarray(bw) * 255).astype(np.uint8)[..., None], 3, -1))
    Image.fromarray(rgb).save(tmp_path / "rgb.ppm")
    assert np.array_equal(ft.load_image(tmp_path / "rgb.ppm"), rgb)
    (tmp_path / "ascii.ppm").write_text("P3\n# comment\n2 1\n255\n1 2 3  250 251 252\n")
    assert ft.load_image(tmp_path / "ascii.ppm").tolist() == [[[1, 2, 3], [250, 251, 252]]]
    Image.fromarray(rgb).save(tmp_path / "x.jpg")                                    # baseline JPEG: read since round 3 (test_jpeg_loader_equals_libjpeg)
    assert np.array_equal(ft.load_image(tmp_path / "x.jpg"), np.asarray(Image.open(tmp_path / "x.jpg").convert("RGB")))
    Image.fromarray(rgb).save(tmp_path / "p.jpg", progressive=True)
    with pytest.raises(ft.FtError) as e:
        ft.load_image(tmp_path / "p.jpg")
    assert "JPEG" in str(e.value) and "progressive" in str(e.value)
    bad = bytearray((tmp_path / "rgb.png").read_bytes())
    bad[40] ^= 0xFF                                                                   # corrupt the IDAT body: CRC check refuses it
    (tmp_path / "bad.png").write_bytes(bad)
    with pytest.raises(ft.FtError):
        ft.load_image(tmp_path / "bad.png")


def test_committed_texture_png_is_what_the_tool_writes():
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_texture_png", os.path.join(H.ROOT, "tools", "make_texture_png.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    img = ft.load_image(os.path.join(H.ROOT, "scenes", "textures", "moon_synth_256x128.png"))   # rows use all five PNG filters
    assert np.array_equal(img, m.moon(256, 128, 1969))


def test_image_texture_lookup_follows_the_reference_index_rule():       # Textures/Image.fs:27-35
    img = np.arange(4 * 3 * 3, dtype=np.uint8).reshape(3, 4, 3) * 7      # height 3, width 4
    orc = O.Oracle()
    orc.clear()
    # Plane.fs:28-33 gives uv = (x, z) of the model-space hit point on the plane y = 0.
    orc.set_objects(orc.group([orc.texture_image(img, [], orc.primitive(ft.PLANE))]))
    orc.add_directional((0, -1, 0), (1, 1, 1))
    orc.commit()
    pts = np.array([[0.1, 0.1], [0.3, 0.1], [0.99, 0.9], [1.6, 0.4], [-0.2, 2.5], [0.5, 0.5]])
    o = np.column_stack([pts[:, 0], np.full(len(pts), 2.0), pts[:, 1]])
    d = np.tile([0.0, -1.0, 0.0], (len(pts), 1))
    hit, t, p, n, c = orc.closest(o, d)
    assert hit.all()
    for (u, v), got in zip(pts, c):
        ru, rv = abs(u - math.floor(u)), abs(v - math.floor(v))          # Texture.repeat
        x, y = math.floor(ru * 4), math.floor(rv * 3)
        assert np.allclose(got, img[y, x] / 255.0, rtol=0, atol=0), (u, v)


def test_jpeg_loader_on_the_committed_texture():
    """host/ImageLoader.cpp reads baseline JPEG (what `Image.Load<Rgb24>` of Scenes/sample.scene:6 needs).  The committed sky texture decodes
    to the bytes libjpeg produces (hash taken with Pillow agreeing byte for byte, tools/make_texture_jpg.py); the scene as written parses."""
    import hashlib
    img = ft.load_image(os.path.join(H.ROOT, "scenes", "textures", "env4_synth.jpg"))
    assert img.shape == (192, 384, 3) and hashlib.sha256(img.tobytes()).hexdigest() == "eccc8619bb3cdbdfaa3b68f27e3611a3fcf7f02d30885be439dbfaf11ea6468d"
    p = ft.parse_scene_file(os.path.join(H.ROOT, "scenes", "sample.scene"))
    assert p.n_objects == 6 and p.n_lights == 2 and p.samples == 1


def test_jpeg_loader_equals_libjpeg(tmp_path):
    """Grey, 4:4:4, 4:2:2 and 4:2:0 files of odd and tiny sizes, two qualities, with and without restart markers, written by Pillow (libjpeg)
    and read back by it: the loader's integer inverse DCT, triangle upsampling and fixed-point colour conversion give the same bytes."""
    Image = pytest.importorskip("PIL.Image")
    rng = np.random.default_rng(3)
    def picture(h, w):
        y, x = np.mgrid[0:h, 0:w]
        img = np.stack([127 + 100 * np.sin(x / 9.0) * np.cos(y / 7.0), 127 + 90 * np.cos(x / 5.0 + y / 11.0), 127 + 80 * np.sin((x + y) / 13.0)], -1) + rng.normal(size=(h, w, 3)) * 12
        return np.clip(img, 0, 255).astype(np.uint8)
    n = 0
    for h, w in ((64, 64), (37, 53), (16, 16), (1, 1), (9, 130), (100, 17)):
        for sub in (0, 1, 2):
            for kw in ({"quality": 95}, {"quality": 40, "restart_marker_blocks": 3}):
                path = str(tmp_path / f"t_{h}_{w}_{sub}_{kw['quality']}.jpg")
                Image.fromarray(picture(h, w)).save(path, subsampling=sub, **kw)
                assert np.array_equal(ft.load_image(path), np.asarray(Image.open(path).convert("RGB"))), path
                n += 1
        path = str(tmp_path / f"g_{h}_{w}.jpg")
        Image.fromarray(picture(h, w)[..., 0]).save(path, quality=80)
        assert np.array_equal(ft.load_image(path), np.asarray(Image.open(path).convert("RGB"))), path
    assert n == 36
    path = str(tmp_path / "progressive.jpg")
    Image.fromarray(picture(32, 32)).save(path, progressive=True)
    with pytest.raises(Exception) as e:                         # refused loudly, not mis-decoded
        ft.load_image(path)
    assert "progressive" in str(e.value)
