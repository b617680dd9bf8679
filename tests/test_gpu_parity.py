"""Parity of the HIP path (through the C ABI) against the CPU oracle.  Every test needs a GPU."""
import os

import numpy as np
import pytest

import functracer_amd as ft
from oracle import ft_oracle_py as O

from . import helpers as H

pytestmark = pytest.mark.gpu

XFORMS = {
    "identity": None,
    "translate": [("translate", (0.3, -0.2, 0.5))],
    "scale": [("scale", (1.5, 0.7, 2.0))],
    "rotate": [("rotate", (1, 2, 3), H.deg(37.0))],
    "composed": [("scale", (2.0, 0.5, 1.25)), ("rotate", (0, 1, 0), H.deg(-64.0)), ("translate", (0.5, 0.25, -0.75))],
}


def test_known_answers_through_the_device_path(hip, golden):
    for case in golden["hand_derived"]["closest"]:
        H.single_prim(hip, case["prim"], lights=False)
        hit, t, p, n, _ = hip.closest([case["o"]], [case["d"]])
        assert bool(hit[0]) == case["hit"], case["name"]
        if case["hit"]:
            assert t[0] == pytest.approx(case["t"], abs=1e-12), case["name"]
            assert np.allclose(p[0], case["p"], atol=1e-12), case["name"]
            assert np.allclose(n[0], case["n"], atol=1e-12), case["name"]


def _shading_cases():
    import json
    with open(os.path.join(H.ROOT, "tests", "golden", "known_answers.json")) as f:
        return json.load(f)["hand_derived_shading"]["cases"]


@pytest.mark.parametrize("case", _shading_cases(), ids=lambda c: c["name"])
def test_hand_derived_shading_on_the_device(hip, case):
    """Closed-form colours derived from Shading.fs / Light.fs / Csg.fs (tests/tools/derive_shading_answers.py; no oracle involved):
    unclamped Lambert, the three regimes of the specular power, attenuation and the shadow test, the per-light unlit colour, the
    per-light mirror bounce and the recursion limit, shaded by getColourForRay on the device (ft_debug_colour)."""
    H.check_shading_case(hip, case)


def test_hand_derived_csg_tables_on_the_device(hip, golden):  # Csg.fs:19-55, 59-72
    for op in ("union", "intersect", "subtract", "exclude"):
        cases = [c for c in golden["hand_derived_csg"]["cases"] if c["op"] == op]
        H.csg_pair(hip, op)
        hit, t, p, n, _ = hip.closest([c["o"] for c in cases], [c["d"] for c in cases])
        for k, c in enumerate(cases):
            assert bool(hit[k]) == c["hit"], (op, c["why"])
            if c["hit"]:
                assert t[k] == pytest.approx(c["t"], abs=1e-12) and np.allclose(n[k], c["n"], atol=1e-12), (op, c["why"], t[k], n[k])


@pytest.mark.parametrize("case", H.round3_cases("closest"), ids=lambda c: c["name"])
def test_hand_derived_triangle_and_transformed_normals_on_the_device(hip, case):    # Triangle.fs:43-66, Transform.fs:77-87
    H.check_closest_case(hip, case)


@pytest.mark.parametrize("case", H.round3_cases("shading"), ids=lambda c: c["name"])
def test_hand_derived_oren_nayar_textures_soft_shadows_on_the_device(hip, case):    # Shading.fs:24-31, 50-63; Texture.fs:8-29; Sphere.fs:6-10
    H.check_shading_case(hip, case)


@pytest.mark.parametrize("case", H.round3_cases("frames"), ids=lambda c: c["name"])
def test_hand_derived_blend_and_corner_average_on_the_device(hip, case):            # Image.fs:83-89, 112-116, 125-145
    H.check_frame_case(hip, case)


def test_debug_colour_equals_the_rendered_pixel(hip):
    """ft_debug_colour is the frame's own shading path fed with explicit rays: the colour of the ray through a pixel centre equals
    that pixel of a 1-sample frame with a zero jitter offset (the device contracts the ray's a*b+c into FMAs, the oracle's
    ray_through_pixel used here does not: 1e-12)."""
    p = _load("hollow-sphere")
    p.lower(hip)
    w, h = 64, 48
    frame, _ = hip.render(p.camera, w, h, 1, np.zeros((1, 2)))
    o, d = zip(*[O.ray_through_pixel(p.camera, w, h, x, y) for y in range(0, h, 5) for x in range(0, w, 7)])
    got = hip.colour_for_ray(np.array(o), np.array(d))
    want = np.array([frame[y, x] for y in range(0, h, 5) for x in range(0, w, 7)])
    assert np.allclose(got, want, rtol=1e-12, atol=0)


def test_csg_hollow_shell_known_answer(hip, golden):
    g = golden["hand_derived"]["csg_hollow_shell"]
    hip.clear()
    shell = hip.subtract(hip.scale(11, hip.primitive(ft.SPHERE)), hip.scale(10, hip.primitive(ft.SPHERE)))
    hip.set_objects(hip.group([shell]))
    hip.commit()
    hit, t, p, n, _ = hip.closest([g["o"]], [g["d"]])
    assert hit[0] == 1 and t[0] == pytest.approx(g["t"], rel=1e-14) and np.allclose(p[0], g["p"]) and np.allclose(n[0], g["n"])


@pytest.mark.parametrize("prim", sorted(H.PRIMS))
@pytest.mark.parametrize("xf", sorted(XFORMS))
def test_primitive_closest_and_blocked(hip, prim, xf):
    orc = O.Oracle()
    for b in (orc, hip):
        H.single_prim(b, prim, XFORMS[xf], lights=False)
    o, d = H.random_rays(20000, seed=hash((prim, xf)) & 0xFFFF)
    # add axis-aligned and exactly-parallel rays: they exercise the Plane.intersect parallel rule and a = 0 quadratics
    axes = np.array([[1, 0, 0], [0, 1, 0], [0, 0, 1], [-1, 0, 0], [0, -1, 0], [0, 0, -1]], dtype=np.float64)
    rng = np.random.default_rng(7)
    ao = rng.uniform(-1.5, 1.5, size=(600, 3))
    ad = axes[rng.integers(0, 6, size=600)] * rng.uniform(0.5, 2.0, size=(600, 1))
    o, d = np.vstack([o, ao]), np.vstack([d, ad])
    H.assert_hits_match(hip.closest(o, d), orc.closest(o, d), what=f"{prim}/{xf}")
    md = np.abs(np.random.default_rng(3).normal(size=o.shape[0])) * 4.0
    assert np.array_equal(hip.blocked(o, d, md), orc.blocked(o, d, md)), f"{prim}/{xf}: lightIsBocked differs"


def _csg_scene(b, op, a_kind, b_kind, nested=False):
    b.clear()
    A = b.primitive(a_kind)
    Bn = b.transform([("scale", 0.65), ("translate", (0.2, 0.1, -0.15))], b.primitive(b_kind))
    node = b.csg(op, A, Bn)
    if nested:
        C = b.transform([("scale", (0.4, 2.0, 0.4))], b.primitive(ft.CYLINDER))
        node = b.csg(ft.SUBTRACT, node, b.translate((0, -1.0, 0), C))
        node = b.union(node, b.translate((1.2, 0, 0), b.scale(0.5, b.primitive(ft.SPHERE))))
    node = b.material(b.rotate((1, 1, 0), H.deg(25.0), node), colour=(0.2, 0.4, 0.8), reflectance=0.3, shineyness=10)
    b.set_objects(b.group([node, b.translate((0, -2.5, 0), b.primitive(ft.PLANE))]))
    b.add_positional((3, 4, -5), (1, 0.01, 0.02), (1, 1, 1))
    b.commit()


@pytest.mark.parametrize("op", [ft.UNION, ft.INTERSECT, ft.SUBTRACT, ft.EXCLUDE])
@pytest.mark.parametrize("kinds", [(ft.CUBE, ft.SPHERE), (ft.SPHERE, ft.CUBE), (ft.SOLID_CYLINDER, ft.SPHERE), (ft.CONE, ft.CUBE)])
@pytest.mark.parametrize("nested", [False, True])
def test_csg_closest_and_blocked(hip, op, kinds, nested):
    orc = O.Oracle()
    for b in (orc, hip):
        _csg_scene(b, op, kinds[0], kinds[1], nested)
    o, d = H.random_rays(20000, seed=op * 10 + kinds[0], origin_scale=2.5)
    H.assert_hits_match(hip.closest(o, d), orc.closest(o, d), what=f"csg {op} {kinds} nested={nested}")
    md = np.abs(np.random.default_rng(5).normal(size=o.shape[0])) * 5.0
    assert np.array_equal(hip.blocked(o, d, md), orc.blocked(o, d, md))


def _bunny_tris():
    with open(H.scene_path("meshes/bunny_synth_res4").replace(".scene", ".ply")) as f:
        return ft.parse_ply(f.read())


@pytest.mark.parametrize("depth", [0, 1, 3, 12])
def test_bsp_mesh_closest_and_blocked(hip, depth):
    tris = _bunny_tris()
    orc = O.Oracle()
    for b in (orc, hip):
        b.clear()
        mesh = b.transform([("rotate", (0, 1, 0), H.deg(180.0)), ("scale", 8)], b.bsp_mesh(depth, tris))
        b.set_objects(b.group([mesh]))
        b.commit()
    if depth:
        info = hip.scene_info()
        assert info["bsp_nodes"] > 0 and info["stack_capacity"] >= 2
    o, d = H.random_rays(30000, seed=depth + 11, origin_scale=2.0, toward=(0.13, 0.88, 0.0), spread=0.5)
    H.assert_hits_match(hip.closest(o, d), orc.closest(o, d), what=f"bspMesh depth {depth}")
    md = np.abs(np.random.default_rng(9).normal(size=o.shape[0])) * 3.0
    assert np.array_equal(hip.blocked(o, d, md), orc.blocked(o, d, md))


def test_mesh_under_csg(hip):
    tris = _bunny_tris()
    orc = O.Oracle()
    for b in (orc, hip):
        b.clear()
        mesh = b.scale(8, b.bsp_mesh(4, tris))
        cut = b.translate((-0.13, 0.9, 0.0), b.scale(0.45, b.primitive(ft.SPHERE)))
        b.set_objects(b.group([b.subtract(mesh, cut)]))
        b.commit()
    o, d = H.random_rays(20000, seed=77, origin_scale=2.0, toward=(-0.13, 0.88, 0.0), spread=0.4)
    H.assert_hits_match(hip.closest(o, d), orc.closest(o, d), what="mesh under subtract")


def test_triangle_primitives_as_group(hip):
    """The `mesh` keyword (SceneParser.fs:116-126): a Group of bare Triangle primitives."""
    tris = _bunny_tris()[:200]
    orc = O.Oracle()
    for b in (orc, hip):
        b.clear()
        nodes = [b.triangle(t[0:3], t[3:6], t[6:9]) for t in tris]
        b.set_objects(b.group([b.scale(8, b.group(nodes))]))
        b.commit()
    o, d = H.random_rays(10000, seed=5, origin_scale=2.0, toward=(-0.13, 0.88, 0.0), spread=0.5)
    H.assert_hits_match(hip.closest(o, d), orc.closest(o, d), what="triangle group")


SMALL = [("moon", 200, 200, 1), ("sample-det", 96, 96, 2), ("hollow-sphere", 160, 90, 1), ("bunny", 160, 90, 4), ("bunny-bsp12", 160, 90, 2), ("night-house-det", 160, 90, 3)]


def _load(name):
    p = ft.parse_scene_file(H.scene_path(name))
    return p


@pytest.mark.parametrize("name,w,h,spp", SMALL)
def test_config_scene_frames_match_oracle(hip, name, w, h, spp):
    p = _load(name)
    orc = O.Oracle()
    p.lower(orc)
    p.lower(hip)
    jit = ft.jitter_pattern(spp)
    want, ost = orc.render(p.camera, w, h, spp, jit)
    got, st = hip.render(p.camera, w, h, spp, jit)
    worst = H.assert_frames_match(got, want, what=name)
    assert worst < 1e-7, f"{name}: expected ~1e-12 agreement, got {worst}"
    assert st["rays_primary"] == w * h * spp == ost["rays_primary"]
    assert st["csg_overflow"] == 0
    # the F#-equivalent ray count reproduces what the literal recursion of the oracle traces
    assert st["rays_reference_equivalent"] == pytest.approx(ost["rays_traced"], rel=1e-12)


def test_golden_frames(hip):
    """Committed oracle frames (tests/golden/frames.npz, made by tests/tools/make_goldens.py)."""
    import os
    path = os.path.join(H.ROOT, "tests", "golden", "frames.npz")
    z = np.load(path)
    for name, w, h, spp in [("hollow-sphere", 96, 54, 1), ("bunny", 96, 54, 2), ("night-house-det", 96, 54, 2), ("sample-det", 64, 64, 1)]:
        p = _load(name)
        p.lower(hip)
        got, _ = hip.render(p.camera, w, h, spp, z[name + "_jitter"])
        H.assert_frames_match(got, z[name], what="golden " + name)


def test_edge_cases(hip):
    cam = ft.make_camera((0, 0, -5), (0, 0, 0), (0, 1, 0), H.deg(50.0))
    jit = ft.jitter_pattern(1)
    # empty scene: every pixel black, no hits
    hip.clear(); hip.set_objects(hip.group([])); hip.add_directional((0, -1, 0), (1, 1, 1)); hip.commit()
    img, st = hip.render(cam, 33, 17, 1, jit)
    assert not img.any() and st["hits_primary"] == 0 and st["rays_shadow"] == 0
    # no lights: every pixel black although rays hit (Shading.fs:139 sums over zero fragments)
    hip.clear(); hip.set_objects(hip.group([hip.primitive(ft.SPHERE)])); hip.commit()
    img, st = hip.render(cam, 33, 17, 1, jit)
    assert not img.any() and st["hits_primary"] > 0
    # ragged / clipped / empty tiles
    hip.add_directional((0, -1, 1), (1, 1, 1)); hip.commit()
    full, _ = hip.render(cam, 33, 17, 1, jit)
    part = np.full_like(full, -1.0)
    hip.render(cam, 33, 17, 1, jit, tiles=[(30, 15, 10, 10), (0, 0, 1, 1), (5, 5, 0, 3), (-4, 3, 6, 2)], out=part)
    mask = np.zeros((17, 33), dtype=bool)
    mask[15:17, 30:33] = True; mask[0, 0] = True; mask[3:5, 0:2] = True
    assert np.array_equal(part[mask], full[mask]) and (part[~mask] == -1.0).all()


def test_full_size_frame_properties(hip):
    """BASELINE size (1920x1080): tile union == whole frame bit for bit, run-to-run determinism,
    chunking invariance, ray accounting."""
    p = _load("hollow-sphere")
    p.lower(hip)
    w, h = 1920, 1080
    jit = ft.jitter_pattern(1)
    full, st = hip.render(p.camera, w, h, 1, jit)
    again, _ = hip.render(p.camera, w, h, 1, jit)
    assert np.array_equal(full, again), "render is not run-to-run deterministic"
    assert st["rays_primary"] == w * h and st["hits_primary"] == w * h          # the camera sits inside the shell
    assert st["rays_shadow"] >= w * h and st["rays_traced"] == st["rays_primary"] + st["rays_shadow"] + st["rays_reflect"]
    tiled = np.zeros_like(full)
    bands = [(0, y, w, 8) for y in range(0, h, 8)]
    for r in range(4):                                                          # 4 "ranks", interleaved 8-row bands
        hip.render(p.camera, w, h, 1, jit, tiles=bands[r::4], out=tiled)
    assert np.array_equal(tiled, full), "union of tiles differs from the single-launch frame"
    hip.set_option("chunk_samples", 300000)
    chunked, st2 = hip.render(p.camera, w, h, 1, jit)
    hip.set_option("chunk_samples", 16 << 20)
    assert st2["n_chunks"] > 1 and np.array_equal(chunked, full), "chunking changes the frame"
    assert np.isfinite(full).all()          # negative channels are legitimate: Lambert is unclamped (Shading.fs:69)


def test_quantise_matches_oracle(hip):
    rng = np.random.default_rng(1)
    rgb = rng.uniform(-0.5, 1.5, size=(1000, 3))
    assert np.array_equal(ft.quantise_rgba8(rgb), O.quantise_rgba8(rgb))


def test_mesh_tie_break_follows_list_order(hip):
    """Equal-t hits: closest keeps the earliest triangle of the list (stable sort, Scene.fs:114-116).
    Coincident triangle pairs with opposite winding and exactly representable coordinates give exactly
    equal t, so the winner shows in the sign of the normal.  16 triangles => the device-side BVH is used."""
    rng = np.random.default_rng(12)
    tris = []
    for k in range(8):
        x0, y0 = float(4 * k), float(rng.integers(-3, 3))
        a, b, c = [x0, y0, 0.0], [x0 + 2, y0, 0.0], [x0, y0 + 2, 0.0]
        pair = [a + b + c, a + c + b]                    # normals +Z and -Z
        if k % 2:
            pair.reverse()
        tris += pair
    tris = np.array(tris)
    orc = O.Oracle()
    for b_ in (orc, hip):
        b_.clear(); b_.set_objects(b_.group([b_.bsp_mesh(0, tris)])); b_.commit()
    o = np.array([[4.0 * k + 0.5, tris[2 * k][1] + 0.5, -4.0] for k in range(8)])
    d = np.tile([0.0, 0.0, 2.0], (8, 1))
    got, want = hip.closest(o, d), orc.closest(o, d)
    assert want[0].all() and np.array_equal(want[3][:, 2], [1.0 if k % 2 == 0 else -1.0 for k in range(8)])
    assert np.array_equal(got[1], want[1]) and np.array_equal(got[3], want[3])


def _textured_scene(b):
    """Grid textures on every uv-carrying primitive (Sphere.fs:6-10, Plane.fs:28-33 through squares and discs),
    nested texture functions, hueShift after a texture, and Oren-Nayar materials (Shading.fs:50-63)."""
    b.clear()
    tex = lambda node, ops: b.texture_grid((0.55, 1.0, 0.41), (0.78, 0.51, 1.0), ops, node)
    objs = [
        tex(b.translate((0, -1, 0), b.primitive(ft.PLANE)), [(0, 0.7, 0.35), (1, H.deg(30.0), 0.0)]),
        tex(b.material(b.translate((-2.2, 0, 0), b.primitive(ft.SPHERE)), colour=(0, 0, 0), reflectance=0.1, shineyness=30), [(0, 0.2, 0.2)]),
        b.hue_shift(1.0, tex(b.translate((0, 0, 0.5), b.rotate((1, 1, 0), H.deg(35.0), b.primitive(ft.CUBE))), [(0, 0.25, 0.25)])),
        tex(b.translate((2.2, -0.7, 0), b.primitive(ft.SOLID_CYLINDER)), [(1, H.deg(-20.0), 0.0), (0, 0.3, 0.3)]),
        b.material(b.translate((-1.0, 1.6, 1.0), b.scale(0.6, b.primitive(ft.SPHERE))), colour=(1, 1, 1), roughness=0.4),
        b.material(b.translate((1.0, 1.6, 1.0), b.scale(0.6, b.primitive(ft.SPHERE))), colour=(0.9, 0.8, 0.7), roughness=0.8, shineyness=5),
    ]
    b.set_objects(b.group(objs))
    b.add_directional((0.5, -1, 1), (1, 1, 1))
    b.add_positional((0, 4, -4), (1, 0.05, 0.01), (0.6, 0.6, 0.9))
    b.commit()


def test_textures_and_oren_nayar(hip):
    orc = O.Oracle()
    for b in (orc, hip):
        _textured_scene(b)
    o, d = H.random_rays(20000, seed=21, origin_scale=3.0, toward=(0, 0, 0), spread=2.0)
    H.assert_hits_match(hip.closest(o, d), orc.closest(o, d), what="textured scene")     # includes the textured colour of each hit
    cam = ft.make_camera((0, 2.5, -7), (0, 0, 0), (0, 1, 0), H.deg(55.0))
    jit = ft.jitter_pattern(2)
    want, _ = orc.render(cam, 160, 120, 2, jit)
    got, st = hip.render(cam, 160, 120, 2, jit)
    worst = H.assert_frames_match(got, want, what="textures + Oren-Nayar")
    assert worst < 1e-6


def test_thirteen_nested_texture_functions(hip):
    """Scene.TextureFunction nests without bound in the reference (Scene.fs:47-53, 68-75); the flat texture record holds thirteen uv
    functions (five until round 3).  A plane and a sphere under the full thirteen - scales and rotations alternating - against the oracle."""
    ops = [(0, 1.1 + 0.05 * k, 0.9 - 0.03 * k) if k % 2 == 0 else (1, H.deg(17.0 + 9.0 * k), 0.0) for k in range(13)]
    orc = O.Oracle()
    for b in (orc, hip):
        b.clear()
        tex = lambda node: b.texture_grid((0.9, 0.2, 0.1), (0.1, 0.3, 0.8), ops, node)
        b.set_objects(b.group([tex(b.translate((0, -1, 0), b.primitive(ft.PLANE))), tex(b.translate((0, 0.2, 0), b.primitive(ft.SPHERE)))]))
        b.add_directional((0.3, -1, 0.6), (1, 1, 1))
        b.commit()
    cam = ft.make_camera((0, 1.5, -5), (0, 0, 0), (0, 1, 0), H.deg(50.0))
    jit = ft.jitter_pattern(2)
    want, _ = orc.render(cam, 160, 120, 2, jit)
    got, _ = hip.render(cam, 160, 120, 2, jit)
    assert H.assert_frames_match(got, want, what="13 uv functions") < 1e-6
    o, d = H.random_rays(5000, seed=5, origin_scale=3.0, toward=(0, 0, 0), spread=2.0)
    H.assert_hits_match(hip.closest(o, d), orc.closest(o, d), what="13 uv functions")


STOCHASTIC = [("night-house", 160, 90, 3), ("sample-soft", 96, 96, 4), ("repeat", 160, 90, 2), ("house", 160, 90, 2),
              ("sample", 128, 128, 2)]                            # Scenes/sample.scene as written: JPEG sky texture, focus, two soft lights


@pytest.mark.parametrize("name,w,h,spp", STOCHASTIC)
def test_seeded_soft_shadows_and_depth_of_field(hip, name, w, h, spp):
    """softdirectional lights (Shading.fs:24-31, Jitter.fs:26-39) and camera focus (Image.fs:91-94) on the seeded
    counter-based stream: the reference is unseeded, so parity is against the oracle drawing the same stream."""
    p = _load(name)
    orc = O.Oracle()
    p.lower(orc)
    p.lower(hip)
    jit = ft.jitter_pattern(spp)
    want, ost = orc.render(p.camera, w, h, spp, jit, seed=1234)
    got, st = hip.render(p.camera, w, h, spp, jit, seed=1234)
    worst = H.assert_frames_match(got, want, what=name)
    assert worst < 1e-6
    assert st["rays_reference_equivalent"] == pytest.approx(ost["rays_traced"], rel=1e-12)
    other, _ = hip.render(p.camera, w, h, spp, jit, seed=99)
    assert not np.array_equal(other, got), "the seed does not reach the stochastic paths"
    # streams are keyed by (pixel, sample, depth, light): tiling and chunking cannot change a pixel
    tiled = np.zeros_like(got)
    hip.set_option("chunk_samples", 4096)
    hip.render(p.camera, w, h, spp, jit, seed=1234, tiles=[(0, 0, w, h // 2)], out=tiled)
    hip.render(p.camera, w, h, spp, jit, seed=1234, tiles=[(0, h // 2, w, h - h // 2)], out=tiled)
    hip.set_option("chunk_samples", 16 << 20)
    assert np.array_equal(tiled, got)


@pytest.mark.parametrize("name", ["hollow-sphere", "night-house", "sample-soft"])
def test_corner_sampling(hip, name):
    """`samples corner` (CornerSampling, Image.fs:125-150): spp = 0 through the ABI."""
    p = _load(name)
    orc = O.Oracle()
    p.lower(orc)
    p.lower(hip)
    w, h = 72, 40
    want, ost = orc.render(p.camera, w, h, 0, None, seed=5)
    got, st = hip.render(p.camera, w, h, 0, None, seed=5)
    H.assert_frames_match(got, want, what=name + " corner")
    assert st["rays_primary"] == (w + 1) * (h + 1) == ost["rays_primary"]
    part = np.full_like(got, -7.0)
    hip.set_option("chunk_samples", 1000)                    # forces the corner grid to be split by rows
    hip.render(p.camera, w, h, 0, None, seed=5, tiles=[(8, 4, 40, 30), (60, 0, 12, 7)], out=part)
    hip.set_option("chunk_samples", 16 << 20)
    mask = np.zeros((h, w), dtype=bool)
    mask[4:34, 8:48] = True; mask[0:7, 60:72] = True
    assert np.array_equal(part[mask], got[mask]) and (part[~mask] == -7.0).all()


def test_multi_device_context_equals_single_device(hip):
    """ft_create with several device ordinals: the scene is replicated and frames are band-partitioned inside the
    library, every device copying its bands (whole rows) straight into the caller's frame.  On a box with one GPU the same
    ordinal is listed three times (three sub-contexts); with more GPUs visible, distinct ordinals are used."""
    import torch
    n_dev = torch.cuda.device_count()
    ordinals = list(range(min(n_dev, 4))) if n_dev > 1 else [0, 0, 0]
    p = _load("night-house")
    p.lower(hip)
    multi = ft.Context(device=ordinals)
    assert multi.devices() == ordinals and (n_dev == 1 or len(set(multi.devices())) == len(ordinals))   # distinct GPUs whenever there are several
    p.lower(multi)
    jit = ft.jitter_pattern(2)
    want, st1 = hip.render(p.camera, 320, 180, 2, jit, seed=3)
    got, st3 = multi.render(p.camera, 320, 180, 2, jit, seed=3)
    assert np.array_equal(got, want)
    assert st3["rays_traced"] == st1["rays_traced"] and st3["rays_primary"] == 320 * 180 * 2
    _, _ = multi.render(p.camera, 320, 180, 2, jit, seed=3, fetch=False)       # frame left in HBM on every device
    later = multi.fetch_frame(np.zeros_like(want))
    assert np.array_equal(later, want)
    part = np.full_like(want, -1.0)
    multi.render(p.camera, 320, 180, 2, jit, seed=3, tiles=[(16, 8, 64, 40)], out=part)
    assert np.array_equal(part[8:48, 16:80], want[8:48, 16:80]) and (part[:8] == -1.0).all()
    # pipelined frames on every device (ft_render_enqueue on a multi-device context), FP64 and RGBA8
    for _ in range(3):
        multi.render_enqueue(p.camera, 320, 180, 2, jit, seed=3)
    stq = multi.wait()
    assert np.array_equal(multi.fetch_frame(np.zeros_like(want)), want)
    for key in ("rays_primary", "rays_shadow", "rays_reflect", "rays_traced", "hits_primary", "rays_reference_equivalent"):
        assert stq[key] == st1[key], key
    multi.render_enqueue(p.camera, 320, 180, 2, jit, seed=3, rgba8=True)
    multi.wait()
    assert np.array_equal(multi.fetch_frame_rgba8(np.zeros((180, 320, 4), dtype=np.uint8)), ft.quantise_rgba8(want))
    u8, _ = multi.render_rgba8(p.camera, 320, 180, 2, jit, seed=3)
    assert np.array_equal(u8, ft.quantise_rgba8(want))
    with ft.PinnedArray(want.shape) as pinned:                      # ft_host_alloc: the bands of every device as one strided copy each, into page-locked memory
        pinned[:] = -3.0
        multi.render(p.camera, 320, 180, 2, jit, seed=3, out=pinned)
        assert np.array_equal(pinned, want)
    tall = multi.render(p.camera, 320, 188, 2, jit, seed=3)[0]      # a frame whose last band is four rows high: the strided copy stops in front of it
    assert np.array_equal(tall, hip.render(p.camera, 320, 188, 2, jit, seed=3)[0])
    multi.close()


def test_program_cli_writes_the_png(tmp_path):
    """functracer_amd/host/Program.cpp (the C++ stand-in for Program.fs): scene file in, PNG out, on the GPU."""
    import os
    import subprocess
    from PIL import Image
    exe = os.path.join(H.ROOT, "functracer_amd", "lib", "functracer")
    out = tmp_path / "sample.png"
    res = subprocess.run([exe, H.scene_path("sample-det"), str(out)], capture_output=True, text=True, timeout=120)
    assert res.returncode == 0, res.stderr
    assert res.stdout == "" and "Shaded scene" in res.stderr          # nothing but the image may ever reach stdout
    p = _load("sample-det")
    orc = O.Oracle()
    p.lower(orc)
    w, h = p.resolution
    want, _ = orc.render(p.camera, w, h, p.samples, ft.jitter_pattern(p.samples))
    got = np.asarray(Image.open(out).convert("RGBA")).astype(int)
    ref = O.quantise_rgba8(want).astype(int)
    assert got.shape == ref.shape
    # truncation to bytes is a step function of the pixel value: allow one level on the few channels that sit on a step
    assert (np.abs(got - ref) <= 1).all() and (got != ref).mean() < 1e-3
    res = subprocess.run([exe, str(tmp_path / "missing.scene")], capture_output=True, text=True, timeout=60)
    assert res.returncode == 1 and "cannot open" in res.stderr         # readScene: message + exit code 1 (Program.fs:10-16)


def test_config5_resolution_properties(hip):
    """BASELINE config 5 geometry (3840x2160, tiled over 8 ranks) at 1 spp: band union == whole frame, and the
    frame is linear in the light colours (every shader term is: Shading.fs:65-98) - exactly so for a factor 2."""
    p = _load("bunny")
    p.lower(hip)
    w, h = 3840, 2160
    jit = ft.jitter_pattern(1)
    full, st = hip.render(p.camera, w, h, 1, jit)
    assert st["rays_primary"] == w * h and np.isfinite(full).all() and full.max() > 0
    from functracer_amd import tiling
    tiled = np.zeros_like(full)
    for r in range(8):
        hip.render(p.camera, w, h, 1, jit, tiles=tiling.bands_for_rank(w, h, r, 8), out=tiled)
    assert np.array_equal(tiled, full)
    # same scene with the light twice as bright
    scene2 = open(H.scene_path("bunny")).read().replace("colour (1,1,1)", "colour (2,2,2)")
    p2 = ft.parse_scene(scene2, base_dir=os.path.join(H.ROOT, "scenes"))
    p2.lower(hip)
    bright, _ = hip.render(p2.camera, w, h, 1, jit)
    assert np.array_equal(bright, 2.0 * full)


def _band_sample_parity(hip, name, w, h, spp, stride, seed=ft.DEFAULT_SEED):
    """Render the whole frame on the device, every `stride`th 8-row band of it with the oracle, and compare those rows."""
    from functracer_amd import tiling
    p = _load(name)
    p.lower(hip)
    orc = O.Oracle()
    p.lower(orc)
    jit = ft.jitter_pattern(spp)
    got, st = hip.render(p.camera, w, h, spp, jit, seed=seed)
    sample = tiling.bands_for_rank(w, h, 0, stride)
    want = np.zeros_like(got)
    _, ost = orc.render(p.camera, w, h, spp, jit, seed=seed, tiles=sample, out=want)
    rows_got, rows_want = tiling.pack_bands(got, sample), tiling.pack_bands(want, sample)
    worst = H.assert_frames_match(rows_got, rows_want, what=f"{name} {w}x{h}x{spp} on every {stride}th band")
    assert worst < 1e-6
    return got, st, ost


@pytest.mark.parametrize("name,spp,stride", [("hollow-sphere", 1, 1), ("bunny", 4, 2), ("bunny", 16, 6), ("night-house-det", 16, 10), ("night-house", 16, 10)])
def test_full_size_configs_against_the_oracle(hip, name, spp, stride):
    """BASELINE configs 2-4 (and the headline frame) at their full 1920x1080 size: the device frame against the oracle on an
    interleaved band sample of the same frame (the whole frame for hollow-sphere), 1e-4 contract, < 1e-6 observed."""
    _, st, ost = _band_sample_parity(hip, name, 1920, 1080, spp, stride)
    assert st["rays_primary"] == 1920 * 1080 * spp and st["csg_overflow"] == 0
    if stride == 1:
        assert st["rays_reference_equivalent"] == ost["rays_traced"]


def test_config5_full_size_64_samples(hip):
    """BASELINE config 5 as named: 3840x2160 x 64 spp (a multi-chunk frame on one GPU).  The union of the eight ranks' band shares
    equals the single-context frame bit for bit, the ray accounting adds up over the shares, and the oracle agrees on a thin band
    sample (every 54th 8-row band = 40 rows x 3840 pixels x 64 samples)."""
    from functracer_amd import tiling
    w, h, spp = 3840, 2160, 64
    full, st, _ = _band_sample_parity(hip, "bunny", w, h, spp, 54)
    assert st["n_chunks"] > 1 and st["rays_primary"] == w * h * spp
    p = _load("bunny")
    jit = ft.jitter_pattern(spp)
    tiled = np.zeros_like(full)
    shares = []
    for r in range(8):
        _, sr = hip.render(p.camera, w, h, spp, jit, tiles=tiling.bands_for_rank(w, h, r, 8), out=tiled)
        shares.append(sr)
    assert np.array_equal(tiled, full), "union of the eight band shares differs from the whole frame"
    for key in ("rays_primary", "rays_shadow", "rays_reflect", "hits_primary", "rays_reference_equivalent"):
        assert sum(s[key] for s in shares) == st[key], key
    work = [s["rays_traced"] for s in shares]
    assert max(work) < 1.15 * (sum(work) / 8), f"band shares uneven: {work}"


def test_rgba8_frames_are_the_quantised_fp64_frames(hip):
    """ft_render_rgba8 quantises on the device (Image.fs:36): bytes identical to ft_quantise_rgba8 of the FP64 frame, whole frame,
    tiles and corner sampling, including a NaN pixel and the pixels of finished blocks; the two frame formats do not mix."""
    for name, spp in (("bunny", 3), ("hollow-sphere", 2), ("moon", 1)):
        p = _load(name)
        p.lower(hip)
        jit = ft.jitter_pattern(spp)
        f64, st = hip.render(p.camera, 256, 192, spp, jit)
        u8, st8 = hip.render_rgba8(p.camera, 256, 192, spp, jit)
        assert np.array_equal(u8, ft.quantise_rgba8(f64)) and st8["rays_traced"] == st["rays_traced"]
        with pytest.raises(ft.FtError):
            hip.fetch_frame(np.zeros_like(f64))                     # the frame in HBM is the RGBA8 one now
        tiles = [(8, 16, 64, 40), (128, 0, 56, 24)]
        part = np.full((192, 256, 4), 7, dtype=np.uint8)
        hip.render_rgba8(p.camera, 256, 192, spp, jit, tiles=tiles, out=part)
        mask = np.zeros((192, 256), dtype=bool)
        for (x0, y0, tw, th) in tiles:
            mask[y0:y0 + th, x0:x0 + tw] = True
        assert np.array_equal(part[mask], u8[mask]) and (part[~mask] == 7).all()
        corner64, _ = hip.render(p.camera, 96, 64, 0, None)
        corner8, _ = hip.render_rgba8(p.camera, 96, 64, 0, None)
        assert np.array_equal(corner8, ft.quantise_rgba8(corner64))
    # NaN passes the clamp and becomes byte 0 (Math.fs:12-16, Image.fs:36): the fractional-power specular case
    case = [c for c in _shading_cases() if c["name"] == "specular_negative_base_fractional_exponent_is_nan"][0]
    H.build_described_scene(hip, case["objects"], case["lights"])
    cam = ft.make_camera((0, 0, -3), (0, 0, 0), (0, 1, 0), H.deg(40.0), 1.0)
    f64, _ = hip.render(cam, 64, 64, 1, np.zeros((1, 2)))
    u8, _ = hip.render_rgba8(cam, 64, 64, 1, np.zeros((1, 2)))
    assert np.isnan(f64).any() and np.array_equal(u8, ft.quantise_rgba8(f64))


def test_jitter_offsets_outside_the_unit_disc(hip):
    """The jitter pattern is the caller's (the reference draws it in the unit disc, Jitter.fs:15-21): offsets beyond one pixel
    widen the bundle k_classify bounds instead of cutting silhouettes off."""
    p = _load("bunny")
    p.lower(hip)
    orc = O.Oracle()
    p.lower(orc)
    jit = np.array([[1.5, -1.5], [-1.5, 1.5], [2.75, 0.25], [0.0, -3.5]])
    got, st = hip.render(p.camera, 320, 240, 4, jit)
    want, _ = orc.render(p.camera, 320, 240, 4, jit)
    assert st["rays_primary_culled"] > 0
    assert H.assert_frames_match(got, want, what="jitter offsets up to 3.5 pixels") < 1e-6
    hip.set_option("classify_pixels", 0)
    try:
        plain, _ = hip.render(p.camera, 320, 240, 4, jit)
    finally:
        hip.set_option("classify_pixels", 1)
    assert np.array_equal(plain, got)
    wild, stw = hip.render(p.camera, 320, 240, 1, np.array([[1e9, 0.0]]))   # absurd offsets: classification stands down
    assert stw["rays_primary_culled"] == 0 and np.isfinite(wild).all()


def test_device_built_bvh_equals_host_built_bvh(hip):
    """The exact BVH of a `bspMesh 0` is built on the device (linear BVH, ft_bvh.hip; "bvh_builder" = 1) or by the host's surface-area
    sweep (0); the default (2) takes the host's tree below 4096 triangles and the device's from there on.  Both stand in for the reference's linear scan (BspMesh.fs:95-97): closest hits (ties included: many
    coincident triangles, whose hits must go to the lowest list index), shadow queries and frames agree bit for bit with each
    other, and with the oracle's brute force within the contract."""
    rng = np.random.default_rng(11)
    centres = rng.normal(size=(3000, 1, 3)) * 0.8
    tris = centres + rng.normal(size=(3000, 3, 3)) * 0.08
    tris[100:400] = tris[100]                                       # 300 coincident triangles: one Morton cell, equal t on every hit
    o, d = H.random_rays(30000, seed=5, origin_scale=2.5, spread=0.8)
    md = np.abs(np.random.default_rng(6).normal(size=o.shape[0])) * 4.0
    cam = ft.make_camera((0, 0.5, -4), (0, 0, 0), (0, 1, 0), H.deg(50.0), 1.0)
    jit = ft.jitter_pattern(2)
    results = {}
    try:
        for builder in (0, 1, 2, 3):
            hip.set_option("bvh_builder", builder)
            hip.clear()
            hip.set_objects(hip.group([hip.material(hip.bsp_mesh(0, tris.reshape(-1, 9)), colour=(0.9, 0.5, 0.2), shineyness=4.0)]))
            hip.add_directional((1, -2, 1), (1, 1, 1))
            hip.add_positional((2, 3, -2), (1, 0.1, 0.01), (0.5, 0.5, 1.0))
            hip.commit()
            ct = hip.commit_times()
            assert (ct["device_bvh_height"] > 0) == (builder in (1, 3)) and (ct["device_bvh_ms"] > 0) == (builder in (1, 3))   # 3000 triangles: the default asks the host
            results[builder] = (hip.closest(o, d), hip.blocked(o, d, md), hip.render(cam, 320, 240, 2, jit)[0])
        big = np.concatenate([tris, tris[:1500] + 0.01])            # 4500 triangles: past the default's threshold
        hip.clear()
        hip.set_objects(hip.group([hip.bsp_mesh(0, big.reshape(-1, 9))]))
        hip.add_directional((1, -2, 1), (1, 1, 1))
        hip.commit()
        assert hip.commit_times()["device_bvh_height"] > 0
    finally:
        hip.set_option("bvh_builder", 2)
    (c0, b0, f0), (c1, b1, f1) = results[0], results[1]
    for x, y in zip(c0, c1):
        assert np.array_equal(x, y)
    assert np.array_equal(b0, b1) and np.array_equal(f0, f1)
    for x, y in zip(c0, results[2][0]):
        assert np.array_equal(x, y)
    assert np.array_equal(b0, results[2][1]) and np.array_equal(f0, results[2][2])
    for x, y in zip(c0, results[3][0]):                             # the device's binned surface-area tree
        assert np.array_equal(x, y)
    assert np.array_equal(b0, results[3][1]) and np.array_equal(f0, results[3][2])
    orc = O.Oracle()
    orc.clear()
    orc.set_objects(orc.group([orc.material(orc.bsp_mesh(0, tris.reshape(-1, 9)), colour=(0.9, 0.5, 0.2), shineyness=4.0)]))
    orc.add_directional((1, -2, 1), (1, 1, 1))
    orc.add_positional((2, 3, -2), (1, 0.1, 0.01), (0.5, 0.5, 1.0))
    orc.commit()
    H.assert_hits_match(c1, orc.closest(o, d), what="device-built BVH")
    assert np.array_equal(b1, orc.blocked(o, d, md))
    want, _ = orc.render(cam, 320, 240, 2, jit)
    assert H.assert_frames_match(f1, want, what="device-built BVH frame") < 1e-6


def test_device_bvh_refuses_a_non_finite_mesh_and_the_host_takes_over(hip):
    """A mesh with a NaN vertex gets no BVH at all from either builder (the reference's linear scan is what remains): the device
    builder refuses it, the commit falls back to the host flattener, and frames still match the oracle."""
    rng = np.random.default_rng(3)
    tris = (rng.normal(size=(40, 1, 3)) * 0.5 + rng.normal(size=(40, 3, 3)) * 0.2)
    tris[7, 1, 2] = np.nan
    cam = ft.make_camera((0, 0, -4), (0, 0, 0), (0, 1, 0), H.deg(40.0), 1.0)
    frames = []
    hip.set_option("bvh_builder", 1)                                # the device builder is asked, whatever the size
    try:
        for b in (hip, O.Oracle()):
            b.clear()
            b.set_objects(b.group([b.bsp_mesh(0, tris.reshape(-1, 9))]))
            b.add_directional((0, -1, 1), (1, 1, 1))
            b.commit()
            frames.append(b.render(cam, 96, 96, 1, np.zeros((1, 2)))[0])
    finally:
        hip.set_option("bvh_builder", 2)
    assert hip.commit_times()["device_bvh_height"] == 0            # the scene in HBM is the host-built one
    assert H.assert_frames_match(frames[0], frames[1], what="NaN vertex") < 1e-6


def test_unclipped_bvh_fast_mode_stays_within_the_contract(hip):
    """Non-default mode: bspMesh depth is ignored and the original (unclipped) triangles are traced through the BVH.
    The reference-shaped clipped BSP stays the parity mode; this one must stay inside the 1e-4 pixel contract."""
    p = _load("bunny-bsp12")
    p.lower(hip)
    jit = ft.jitter_pattern(2)
    want, _ = hip.render(p.camera, 320, 180, 2, jit)
    hip.set_option("mesh_unclipped_bvh", 1)
    try:
        p.lower(hip)
        assert hip.scene_info()["bsp_nodes"] == 0
        got, _ = hip.render(p.camera, 320, 180, 2, jit)
    finally:
        hip.set_option("mesh_unclipped_bvh", 0)
    err = H.pixel_errors(got, want)
    moved = int((err > H.PIXEL_RTOL).any(axis=-1).sum())
    # Measured: 77 of 57,600 pixels (0.13 %) - fragment edges where the reference's clipped slivers and the original
    # triangle disagree about a grazing hit.  That is why the mode is opt-in and not the parity path.
    assert moved <= 0.005 * err.shape[0] * err.shape[1], moved
    assert np.median(err) < 1e-12


def test_unclipped_bvh_equals_the_reference_at_depth_0(hip):
    """What the fast mode computes IS a reference configuration: `bspMesh 0 file` - every original triangle, no clipping, brute force
    (BspMesh.fs:95-97).  So the mode has an oracle: the device with "mesh_unclipped_bvh" = 1 on the depth-12 scene against the ORACLE
    rendering the same PLY at depth 0, under the parity contract (1e-4 relative per channel, no pixel outside)."""
    text = open(H.scene_path("bunny-bsp12")).read()
    assert "bspMesh 12" in text
    flat = ft.parse_scene(text.replace("bspMesh 12", "bspMesh 0"), os.path.join(H.ROOT, "scenes"))
    deep = _load("bunny-bsp12")
    orc = O.Oracle()
    flat.lower(orc)
    jit = ft.jitter_pattern(2)
    want, wst = orc.render(deep.camera, 320, 180, 2, jit)
    hip.set_option("mesh_unclipped_bvh", 1)
    try:
        deep.lower(hip)
        assert hip.scene_info()["bsp_nodes"] == 0
        got, st = hip.render(deep.camera, 320, 180, 2, jit)
    finally:
        hip.set_option("mesh_unclipped_bvh", 0)
    worst = H.assert_frames_match(got, want, what="mesh_unclipped_bvh on bunny-bsp12 vs the oracle at bspMesh 0")
    assert worst < 1e-6 and st["rays_reference_equivalent"] == wst["rays_traced"]


def test_image_textures_match_oracle(hip):
    """Texture.Image under uv functions, hueShift and ignoreLight (the sky-sphere use of Scenes/sample.scene:6)."""
    rng = np.random.default_rng(11)
    sky = rng.integers(0, 256, size=(32, 64, 3), dtype=np.uint8)
    tile = rng.integers(0, 256, size=(5, 7, 3), dtype=np.uint8)          # odd sizes: u*width is rarely exact
    orc = O.Oracle()
    for b in (orc, hip):
        b.clear()
        dome = b.ignore_light(b.texture_image(sky, [], b.scale(40, b.primitive(ft.SPHERE))))
        floor = b.texture_image(tile, [(0, 0.7, 1.3), (1, 0.4, 0.0)], b.translate((0, -1, 0), b.primitive(ft.PLANE)))
        ball = b.hue_shift(0.0, b.texture_image(tile, [(1, -1.1, 0.0)], b.material(b.primitive(ft.SPHERE), colour=(1, 1, 1), reflectance=0.3, shineyness=20)))
        b.set_objects(b.group([dome, floor, ball]))
        b.add_directional((-1, -2, 1.5), (1, 1, 1))
        b.add_positional((2, 3, -2), (1, 0.05, 0.01), (0.8, 0.7, 0.6))
        b.commit()
    o, d = H.random_rays(20000, seed=33, origin_scale=2.5, toward=(0, 0, 0), spread=2.0)
    H.assert_hits_match(hip.closest(o, d), orc.closest(o, d), what="image textures")
    cam = ft.make_camera((0, 1.5, -5), (0, 0, 0), (0, 1, 0), H.deg(60.0))
    jit = ft.jitter_pattern(2)
    want, _ = orc.render(cam, 160, 120, 2, jit)
    got, _ = hip.render(cam, 160, 120, 2, jit)
    assert H.assert_frames_match(got, want, what="image textures") < 1e-6


@pytest.mark.parametrize("name", ["hollow-sphere", "night-house", "sample-soft"])
def test_levels_launched_change_no_pixel_and_no_count(hip, name):
    """One k_bounce per level of the reflection tree; the host launches as many levels as the previous frame of the same signature
    needed (+ 1) and the last one follows what it still spawns in registers.  Every way of cutting the levels - all of them, the hint,
    a hint that is too short because the previous frame looked at nothing reflective - gives the same frame and the same counts."""
    p = _load(name)
    p.lower(hip)
    jit = ft.jitter_pattern(2)
    away = ft.make_camera(tuple(p.camera.o), tuple(2 * o - l for o, l in zip(p.camera.o, p.camera.look_at)), tuple(p.camera.up), p.camera.fov_y, p.camera.aspect_ratio)
    frames, stats = [], []
    try:
        hip.set_option("level_hint", 0)
        frames.append(hip.render(p.camera, 160, 90, 2, jit)); stats.append(frames[-1][1])          # every level launched
        hip.set_option("level_hint", 1)
        hip.render(p.camera, 160, 90, 2, jit)                                                      # sets the hint
        frames.append(hip.render(p.camera, 160, 90, 2, jit)); stats.append(frames[-1][1])          # hinted
        hip.render(away, 160, 90, 2, jit)                                                          # same signature, shallow frame: the hint shrinks
        frames.append(hip.render(p.camera, 160, 90, 2, jit)); stats.append(frames[-1][1])          # the hint is too short: the last level follows
        for depth in (0, 1, 3):                                                                    # the recursion limit cuts the levels too
            a, sa = hip.render(p.camera, 160, 90, 2, jit, max_depth=depth)
            hip.set_option("level_hint", 0)
            b, sb = hip.render(p.camera, 160, 90, 2, jit, max_depth=depth)
            hip.set_option("level_hint", 1)
            assert np.array_equal(a, b) and sa["rays_reflect"] == sb["rays_reflect"]
        for few in (0, 1000, 10 ** 9):                                                            # levels with few rays get no launch: every threshold, the same frame
            hip.set_option("follow_below", few)
            hip.render(p.camera, 160, 90, 2, jit)                                                  # sets the hint under this threshold
            frames.append(hip.render(p.camera, 160, 90, 2, jit)); stats.append(frames[-1][1])
        assert stats[-1]["n_launches"] < stats[0]["n_launches"]                                    # everything after level 1 followed
    finally:
        hip.set_option("level_hint", 1)
        hip.set_option("follow_below", -1)
    assert stats[0]["rays_reflect"] > 0
    for (img, _), st in zip(frames[1:], stats[1:]):
        assert np.array_equal(img, frames[0][0])
        for key in ("rays_shadow", "rays_reflect", "rays_traced", "hits_total", "rays_reference_equivalent"):
            assert st[key] == stats[0][key], key
    assert stats[2]["n_launches"] < stats[0]["n_launches"]


@pytest.mark.parametrize("name", ["bunny", "night-house", "hollow-sphere", "sample-soft"])
def test_samples_per_wavefront_change_no_pixel_and_no_count(hip, name):
    """k_primary deals the samples of an 8x8 pixel block to wavefronts as one offset of 64 pixels, or as 2 .. 16 offsets of fewer
    pixels (option wave_samples; 16 when 0), and numbers the samples accordingly (slot_at).  A sample's ray and its random streams
    are the same under every numbering, and k_resolve sums a pixel's samples in sample order under every numbering, so the frame and every count are identical, with sample counts that have a power of two
    in them (12 = 4 x 3, 16) and without (3, 1), on whole frames and on tiles."""
    p = _load(name)
    p.lower(hip)
    w, h = 192, 128
    tiles = [(0, 0, 64, 64), (64, 32, 128, 96)]
    try:
        for spp in (1, 3, 12, 16):
            jit = ft.jitter_pattern(spp)
            ref = ref_t = None
            for g in (1, 0, 2, 4, 8, 16):
                hip.set_option("wave_samples", g)
                img, st = hip.render(p.camera, w, h, spp, jit)
                img_t, st_t = hip.render(p.camera, w, h, spp, jit, tiles=tiles)
                if ref is None:
                    ref, ref_t = (img, st), (img_t, st_t)
                    continue
                assert np.array_equal(img, ref[0]) and np.array_equal(img_t, ref_t[0]), (spp, g)
                for key in ("rays_shadow", "rays_reflect", "rays_traced", "hits_total", "rays_reference_equivalent"):
                    assert st[key] == ref[1][key] and st_t[key] == ref_t[1][key], (spp, g, key)
    finally:
        hip.set_option("wave_samples", 0)
    with pytest.raises(Exception):
        hip.set_option("wave_samples", 3)
    with pytest.raises(Exception):
        hip.set_option("wave_samples", 32)


@pytest.mark.parametrize("name", ["bunny", "moon", "hollow-sphere", "bunny-bsp12", "sample-det"])
def test_pixel_block_classification_changes_no_pixel(hip, name):
    """k_classify finishes 64-pixel blocks that cannot see any object before a single ray is generated: the frame, the hit
    counts and the ray counts must not notice, whole frame or tiles."""
    p = _load(name)
    p.lower(hip)
    jit = ft.jitter_pattern(3)
    w, h = 256, 192
    tiles = [(0, 0, 128, 192), (128, 64, 128, 128), (128, 0, 64, 64)]
    out = {}
    try:
        for on in (0, 1):
            hip.set_option("classify_pixels", on)
            out[on] = hip.render(p.camera, w, h, 3, jit) + hip.render(p.camera, w, h, 3, jit, tiles=tiles)
    finally:
        hip.set_option("classify_pixels", 1)
    (f0, s0, t0, st0), (f1, s1, t1, st1) = out[0], out[1]
    assert np.array_equal(f0, f1) and np.array_equal(t0, t1)
    assert s0["rays_primary_culled"] == 0
    if name != "hollow-sphere":                                   # its camera sits inside the shell: every block sees it
        assert s1["rays_primary_culled"] > 0 and st1["rays_primary_culled"] > 0
    for key in ("rays_primary", "rays_shadow", "rays_reflect", "hits_primary", "hits_total", "rays_reference_equivalent"):
        assert s0[key] == s1[key] and st0[key] == st1[key], key
    # rays_traced counts what was generated: the primaries of finished blocks are exactly the difference
    assert s0["rays_traced"] == s1["rays_traced"] + s1["rays_primary_culled"] and st0["rays_traced"] == st1["rays_traced"] + st1["rays_primary_culled"]


def _grid_of_pairs(b, nx, ny, reflect=0.4):
    """nx * ny CSG pairs (cube - sphere / cube & sphere alternating, as Scenes/hollow-sphere.scene:7-33 does) inside a shell."""
    b.clear()
    items = [b.material(b.subtract(b.scale(30, b.primitive(ft.SPHERE)), b.scale(29, b.primitive(ft.SPHERE))), colour=(0.4, 0.4, 0.4))]
    for j in range(ny):
        for i in range(nx):
            op = b.subtract if (i + j) % 2 == 0 else b.intersect
            node = op(b.primitive(ft.CUBE), b.scale(0.65, b.primitive(ft.SPHERE)))
            items.append(b.material(b.translate((1.3 * (i - nx / 2), 1.3 * (j - ny / 2), 0.0), node), colour=(1, (i % 3) / 2, (j % 3) / 2), reflectance=reflect, shineyness=10))
    b.set_objects(b.group(items))
    b.add_positional((0, 0, -8), (1, 0.01, 0.02), (1, 1, 1))
    b.commit()


def test_axis_aligned_rays_take_the_generic_csg_route(hip):
    """Rays parallel to cube faces get hits at the ray origin from Plane.fs:13-16: an operand then has more than two hits and
    OP_CSG_PAIR must fall back to the generic sequence, coherent and incoherent alike."""
    orc = O.Oracle()
    for b in (orc, hip):
        _grid_of_pairs(b, 3, 3)
    rng = np.random.default_rng(77)
    n = 6000
    o = rng.uniform(-2.5, 2.5, size=(n, 3))
    o[: n // 2] = np.round(o[: n // 2] * 4) / 4                  # many origins exactly on face planes (x, y, z = +-0.5 + k * 1.3 are not; 0.25 steps inside cubes are)
    axes = np.eye(3)[rng.integers(0, 3, size=n)] * rng.choice([-1.0, 1.0], size=(n, 1)) * rng.uniform(0.5, 2.0, size=(n, 1))
    d = axes.copy()
    d[n // 2:] += rng.normal(scale=1e-9, size=(n - n // 2, 3))    # and nearly parallel ones, on both sides of the 1e-7 rule
    H.assert_hits_match(hip.closest(o, d), orc.closest(o, d), what="axis-aligned rays through CSG pairs")
    md = np.abs(rng.normal(size=n)) * 5.0
    assert np.array_equal(hip.blocked(o, d, md), orc.blocked(o, d, md))


def test_more_items_than_the_item_mask_covers(hip):
    """150 top-level items: the wave-level item masks cover 128, the rest runs behind its own OP_CULL; frame, ray counts and
    explicit incoherent rays must match the oracle."""
    orc = O.Oracle()
    for b in (orc, hip):
        _grid_of_pairs(b, 15, 10, reflect=0.3)
    assert hip.scene_info()["leaves"] == 2 + 2 * 150
    o, d = H.random_rays(20000, seed=5, origin_scale=6.0, toward=(0, 0, 0), spread=6.0)
    H.assert_hits_match(hip.closest(o, d), orc.closest(o, d), what="150 CSG pairs")
    cam = ft.make_camera((6, 5, -14), (0, 0, 0), (0, 1, 0), H.deg(60.0))
    jit = ft.jitter_pattern(2)
    want, ost = orc.render(cam, 192, 128, 2, jit)
    got, st = hip.render(cam, 192, 128, 2, jit)
    assert H.assert_frames_match(got, want, what="150 CSG pairs") < 1e-6
    assert st["rays_reference_equivalent"] == ost["rays_traced"]


def test_coincident_items_resolve_ties_in_scene_order(hip):
    """Two identical spheres with different materials: every hit is a tie, and the reference's stable sort keeps the first
    object of the scene (Scene.fs:112-116) - whatever route the wavefront takes to the items."""
    orc = O.Oracle()
    for b in (orc, hip):
        b.clear()
        objs = [b.material(b.primitive(ft.SPHERE), colour=c) for c in ((1, 0, 0), (0, 1, 0), (0, 0, 1))]
        objs += [b.material(b.translate((2.5 * k, 0, 0), b.primitive(ft.CUBE)), colour=(1, 1, 1)) for k in (-1, 1)]
        b.set_objects(b.group(objs))
        b.add_directional((0, -1, 1), (1, 1, 1))
        b.commit()
    o, d = H.random_rays(5000, seed=9, origin_scale=3.0)
    got, want = hip.closest(o, d), orc.closest(o, d)
    H.assert_hits_match(got, want, what="coincident spheres")
    hit = want[0].astype(bool)
    on_sphere = hit & (np.abs(np.linalg.norm(want[2], axis=1) - 1.0) < 1e-9)
    assert on_sphere.sum() > 100 and (got[4][on_sphere] == (1.0, 0.0, 0.0)).all()
    cam = ft.make_camera((0, 1, -6), (0, 0, 0), (0, 1, 0), H.deg(50.0))
    jit = ft.jitter_pattern(1)
    want_f, _ = orc.render(cam, 128, 96, 1, jit)
    got_f, _ = hip.render(cam, 128, 96, 1, jit)
    assert H.assert_frames_match(got_f, want_f, what="coincident spheres") < 1e-9


def test_pipelined_frames_equal_blocking_frames(hip):
    """ft_render_enqueue / ft_render_wait: frames queued back to back, each equal to its blocking twin; the wait reports the
    last frame's statistics and the stage times summed over the queued frames."""
    p = _load("hollow-sphere")
    p.lower(hip)
    jit = ft.jitter_pattern(2)
    cams = [ft.make_camera((5.7 - 0.4 * k, 5.7, -5.7), (0, 0, 4), (0, 1, 0), H.deg(60.0)) for k in range(4)]
    blocking = [hip.render(c, 160, 96, 2, jit) for c in cams]
    per_frame_launches = hip.kernel_times()["closest"]["launches"]
    for c in cams[:3]:
        hip.render_enqueue(c, 160, 96, 2, jit)
    st = hip.wait()
    assert np.array_equal(hip.fetch_frame(np.zeros((96, 160, 3))), blocking[2][0])
    for key in ("rays_traced", "rays_shadow", "rays_reflect", "hits_primary", "rays_reference_equivalent"):
        assert st[key] == blocking[2][1][key], key
    assert hip.kernel_times()["closest"]["launches"] == 3 * per_frame_launches
    # ft_render_enqueue_into: every queued frame's copy to the host queued behind it, into page-locked memory, FP64 and RGBA8, two in flight
    with ft.PinnedArray((2, 96, 160, 3)) as f64, ft.PinnedArray((2, 96, 160, 4), dtype=np.uint8) as u8:
        f64[:] = -1.0
        u8[:] = 7
        for k in range(4):
            hip.render_enqueue(cams[k], 160, 96, 2, jit, out=f64[k & 1])
            if k >= 1:
                pass                                               # (slot k & 1 is free again once frame k - 2 was retired by the enqueue of frame k)
            if k == 1:
                hip.wait()
                assert np.array_equal(f64[0], blocking[0][0]) and np.array_equal(f64[1], blocking[1][0])
        hip.wait()
        assert np.array_equal(f64[0], blocking[2][0]) and np.array_equal(f64[1], blocking[3][0])
        for k in range(2):
            hip.render_enqueue(cams[k], 160, 96, 2, jit, rgba8=True, out=u8[k])
        hip.wait()
        assert np.array_equal(u8[0], ft.quantise_rgba8(blocking[0][0])) and np.array_equal(u8[1], ft.quantise_rgba8(blocking[1][0]))
    hip.render_enqueue(cams[3], 160, 96, 2, jit)                  # a blocking call retires what is in flight first
    img, st2 = hip.render(cams[0], 160, 96, 2, jit)
    assert np.array_equal(img, blocking[0][0]) and st2["rays_traced"] == blocking[0][1]["rays_traced"]
    assert hip.wait()["rays_traced"] == 0                         # nothing queued


def test_three_frames_in_flight_on_two_main_streams(hip):
    """Queued frames of one chunk without reflection levels are SIMPLE frames (frame pipeline): three in flight, their k_primary
    launches alternating between two main streams, k_classify a frame ahead, k_resolve and the copy out behind on a third.  Seven frames from
    three cameras, each delivered into its own page-locked buffer, FP64 and RGBA8 mixed, a tile list and an upload (new jitter pattern) in
    between: every buffer must hold its frame's blocking twin, whatever the option says."""
    p = _load("bunny")
    p.lower(hip)
    w, h, spp = 256, 192, 4
    jit, jit2 = ft.jitter_pattern(spp), ft.jitter_pattern(spp, seed=99)
    cams = [ft.make_camera((0.0, 0.9, -7.0), (x, 0.7, 0.0), (0, 1, 0), H.deg(40.0)) for x in (0.6, -0.9, 0.0)]
    left = [(0, 0, 128, 192)]
    plan = [(0, jit, None, False), (1, jit, None, False), (2, jit, None, True), (0, jit, None, False), (1, jit2, None, False), (2, jit2, left, False), (0, jit, None, True)]
    hip.set_option("two_mains", 0)
    want = []
    for cam, j, tiles, rgba8 in plan:
        if rgba8:
            want.append(hip.render_rgba8(cams[cam], w, h, spp, j)[0])
        else:
            full = np.full((h, w, 3), -5.0)
            hip.render(cams[cam], w, h, spp, j, tiles=tiles, out=full)
            want.append(full)
    for two in (1, 0):
        hip.set_option("two_mains", two)
        with ft.PinnedArray((len(plan), h, w, 3)) as f64, ft.PinnedArray((len(plan), h, w, 4), dtype=np.uint8) as u8:
            f64[:] = -5.0
            for rep in range(2):                                    # the second round starts with three frames of the first still in flight
                for k, (cam, j, tiles, rgba8) in enumerate(plan):
                    hip.render_enqueue(cams[cam], w, h, spp, j, tiles=tiles, rgba8=rgba8, out=u8[k] if rgba8 else f64[k])
            st = hip.wait()
            for k, (cam, j, tiles, rgba8) in enumerate(plan):
                assert np.array_equal(u8[k] if rgba8 else f64[k], want[k]), (two, k)
            assert st["rays_primary"] == w * h * spp
    hip.set_option("two_mains", 1)


def test_zero_fill_skip_and_classification_ahead_change_no_pixel(hip):
    """Two things a stream of frames does that a single frame does not: a queued frame's k_classify runs on a second stream
    beside the frame before it, and Colour.Zero is not written again into blocks the last frame of the same signature (scene, camera,
    size, pixel list) left zero.  The bunny covers a small part of the frame and the two cameras see it in different places, so a stale
    zero (or a stale colour) from the frame before would show at once.  Every sequence must end in its last frame's blocking twin."""
    p = _load("bunny")
    p.lower(hip)
    w, h, spp = 256, 192, 2
    jit = ft.jitter_pattern(spp)
    A = ft.make_camera((0.0, 0.9, -7.0), (0.6, 0.8, 0.0), (0, 1, 0), H.deg(40.0))
    B = ft.make_camera((0.0, 0.9, -7.0), (-0.9, 0.6, 0.0), (0, 1, 0), H.deg(40.0))
    for opt in ("zero_fill_skip", "classify_ahead"):
        hip.set_option(opt, 0)
    ref = {k: hip.render(c, w, h, spp, jit)[0] for k, c in (("A", A), ("B", B))}
    ref8 = {k: hip.render_rgba8(c, w, h, spp, jit)[0] for k, c in (("A", A), ("B", B))}
    left = [(0, 0, 128, 192)]
    ref_left = hip.render(A, w, h, spp, jit, tiles=left)[0]
    for opt in ("zero_fill_skip", "classify_ahead"):
        hip.set_option(opt, 1)
    seen = lambda img, cam: (np.abs(img).sum(axis=-1) > 0)
    assert seen(ref["A"], A).any() and (seen(ref["A"], A) != seen(ref["B"], B)).any() and (~seen(ref["A"], A)).mean() > 0.5   # sparse, and differently so
    cams = {"A": A, "B": B}
    for seq in ("AA", "AB", "ABA", "BBAAB", "ABABAB", "AAAA"):
        for name in seq:
            hip.render_enqueue(cams[name], w, h, spp, jit)
        hip.wait()
        assert np.array_equal(hip.fetch_frame(np.zeros((h, w, 3))), ref[seq[-1]]), seq
    for seq in ("aA", "AaA", "AbA", "bBbB", "AbaB", "aabb", "bAbA"):          # lower case: an RGBA8 frame in between (its own buffer, its own signature)
        for name in seq:
            hip.render_enqueue(cams[name.upper()], w, h, spp, jit, rgba8=name.islower())
        hip.wait()
        last = seq[-1]
        if last.islower():
            assert np.array_equal(hip.fetch_frame_rgba8(np.zeros((h, w, 4), dtype=np.uint8)), ref8[last.upper()]), seq
        else:
            assert np.array_equal(hip.fetch_frame(np.zeros((h, w, 3))), ref[last]), seq
    # blocking calls, tiles in between (another pixel list: another signature) and the same camera again
    for _ in range(2):
        assert np.array_equal(hip.render(A, w, h, spp, jit)[0], ref["A"])
    assert np.array_equal(hip.render(A, w, h, spp, jit, tiles=left)[0][:, :128], ref_left[:, :128])
    assert np.array_equal(hip.render(A, w, h, spp, jit)[0], ref["A"])
    assert np.array_equal(hip.render(B, w, h, spp, jit)[0], ref["B"])
    # another scene under the same camera: the signature follows the commit
    hip.clear()
    hip.set_objects(hip.group([hip.translate((1.5, 1.0, 0.0), hip.primitive(ft.SPHERE))]))
    hip.add_directional((0, -1, 1), (1, 1, 1))
    hip.commit()
    hip.set_option("zero_fill_skip", 0)
    want = hip.render(B, w, h, spp, jit)[0]
    hip.set_option("zero_fill_skip", 1)
    p.lower(hip)
    assert np.array_equal(hip.render(B, w, h, spp, jit)[0], ref["B"])
    hip.clear()
    hip.set_objects(hip.group([hip.translate((1.5, 1.0, 0.0), hip.primitive(ft.SPHERE))]))
    hip.add_directional((0, -1, 1), (1, 1, 1))
    hip.commit()
    assert np.array_equal(hip.render(B, w, h, spp, jit)[0], want)


def test_overflowing_hit_lists_grow_and_the_frame_still_matches(hip):
    """A mesh under CSG at a capacity of 2 hits: most lines through the bunny cross it more often.  The blocking call doubles the
    capacity until the lists hold every hit and delivers the oracle's frame; with "csg_auto_grow" off the same frame is refused
    loudly (FT_ERR_OVERFLOW) - a hit is never dropped silently."""
    from functracer_amd._capi import FtError
    tris = np.asarray(_bunny_tris()).reshape(-1, 9)
    orc = O.Oracle()
    cam = ft.make_camera((0, 1, -6), (0, 0.6, 0), (0, 1, 0), H.deg(40.0))
    jit = ft.jitter_pattern(1)

    def build(b):
        b.clear()
        m = b.scale(7.0, b.bsp_mesh(3, tris))
        node = b.subtract(m, b.translate((0.0, 0.9, -0.3), b.scale(0.5, b.primitive(ft.SPHERE))))
        b.set_objects(b.group([b.material(node, colour=(0.8, 0.5, 0.3), reflectance=0.2, shineyness=10)]))
        b.add_directional((-1, -1, 1), (1, 1, 1))
        b.commit()

    try:
        hip.set_option("csg_mesh_capacity", 2)
        hip.set_option("csg_auto_grow", 0)
        build(hip)
        small = hip.scene_info()["csg_capacity"]
        with pytest.raises(FtError, match="OVERFLOW"):
            hip.render(cam, 96, 64, 1, jit)
        hip.set_option("csg_auto_grow", 1)
        build(orc)
        want, ost = orc.render(cam, 96, 64, 1, jit)
        got, st = hip.render(cam, 96, 64, 1, jit)
        assert H.assert_frames_match(got, want, what="grown hit lists") < 1e-9
        assert st["rays_reference_equivalent"] == ost["rays_traced"] and st["csg_overflow"] == 0
        assert hip.scene_info()["csg_capacity"] > small           # and the larger lists stay for the next frame
        again, _ = hip.render(cam, 96, 64, 1, jit)
        assert np.array_equal(again, got)
        hip.set_option("csg_mesh_capacity", 2)                    # a QUEUED frame is not rendered again: its overflow is reported, by
        build(hip)                                                # whichever call retires it, and not mistaken for that call's own
        hip.render_enqueue(cam, 96, 64, 1, jit)
        with pytest.raises(FtError, match="OVERFLOW"):
            hip.render(cam, 96, 64, 1, jit)
        assert hip.scene_info()["csg_capacity"] == small
        once_more, _ = hip.render(cam, 96, 64, 1, jit)            # nothing queued now: this one grows and delivers
        assert np.array_equal(once_more, got)
    finally:
        hip.set_option("csg_auto_grow", 1)
        hip.set_option("csg_mesh_capacity", 32)


def test_hit_lists_larger_than_the_lds_fold_lanes(hip):
    """Two meshes under nested CSG at a capacity of 48 hits each: 100+ list entries per lane, 400 KiB for 256 lanes - more LDS than
    a workgroup has.  The scene then runs with fewer live lanes per wave, each owning several lanes' columns (HitList)."""
    tris = np.asarray(_bunny_tris()).reshape(-1, 9)
    orc = O.Oracle()
    hip.set_option("csg_mesh_capacity", 48)
    try:
        for b in (orc, hip):
            b.clear()
            m1 = b.scale(7.0, b.bsp_mesh(0, tris))
            m2 = b.translate((0.25, 0.1, 0.0), b.scale(7.0, b.bsp_mesh(2, tris)))
            node = b.subtract(b.union(m1, b.translate((0.1, 0.9, 0.0), b.scale(0.4, b.primitive(ft.SPHERE)))), m2)
            b.set_objects(b.group([b.material(node, colour=(0.8, 0.5, 0.3), reflectance=0.3, shineyness=10), b.translate((0, -0.2, 0), b.primitive(ft.PLANE))]))
            b.add_directional((-1, -2, 1.5), (1, 1, 1))
            b.commit()
        assert hip.scene_info()["csg_capacity"] * 16 * 256 > 160 * 1024          # really beyond one workgroup's LDS
        o, d = H.random_rays(6000, seed=3, origin_scale=2.5, toward=(0, 0.8, 0), spread=1.0)
        H.assert_hits_match(hip.closest(o, d), orc.closest(o, d), what="folded lanes")
        md = np.abs(np.random.default_rng(8).normal(size=o.shape[0])) * 5.0
        assert np.array_equal(hip.blocked(o, d, md), orc.blocked(o, d, md))
        cam = ft.make_camera((0.5, 1.2, -3.0), (0, 0.8, 0), (0, 1, 0), H.deg(50.0))
        jit = ft.jitter_pattern(2)
        want, ost = orc.render(cam, 96, 64, 2, jit)
        got, st = hip.render(cam, 96, 64, 2, jit)
        assert H.assert_frames_match(got, want, what="folded lanes") < 1e-6 and st["rays_reference_equivalent"] == ost["rays_traced"]
    finally:
        hip.set_option("csg_mesh_capacity", 32)
