"""The CPU oracle against every known-answer vector the reference's own tests hold for the render
path (FuncTracer.Tests/Geometry/*.fs), plus the hand-derived answers of tests/golden/known_answers.json.
Runs without a GPU."""
import numpy as np
import pytest
from hypothesis import given, settings
from hypothesis import strategies as st

from oracle import ft_oracle_py as O

from . import helpers as H


def test_aabb_reference_vectors(golden):                      # FuncTracer.Tests/Geometry/BoundingBox.fs:11-27
    g = golden["reference_tests"]["aabb"]
    for case in g["cases"]:
        assert O.aabb_intersects(g["box_min"], g["box_max"], case["o"], case["d"]) == case["expect"]


def _slice_example(g, a, b, c):                               # Triangle.Tests.fs:19-27
    above, below = O.slice_triangle(g["plane_p0"], g["plane_n"], [a, b, c])
    A, B, Cc = g["a"], g["b"], g["c"]
    ab, ac = g["ab_intercept"], g["ac_intercept"]
    assert above.shape[0] == 1 and below.shape[0] == 2
    assert np.array_equal(above[0], np.array([A, ab, ac]))    # exact structural equality, like Assert.Equal
    assert np.array_equal(below[0], np.array([ab, B, Cc]))
    assert np.array_equal(below[1], np.array([Cc, ac, ab]))


def test_triangle_slice_reference_vectors(golden):            # Triangle.Tests.fs:29-38
    g = golden["reference_tests"]["triangle_slice"]
    a, b, c = g["a"], g["b"], g["c"]
    _slice_example(g, a, b, c)
    _slice_example(g, c, a, b)
    _slice_example(g, b, c, a)


def test_triangle_slice_wholly_above_below(golden):           # Triangle.Tests.fs:40-54
    g = golden["reference_tests"]["triangle_slice"]
    above, below = O.slice_triangle(g["plane_p0"], g["plane_n"], g["wholly_above"])
    assert np.array_equal(above[0], np.array(g["wholly_above"])) and below.shape[0] == 0
    above, below = O.slice_triangle(g["plane_p0"], g["plane_n"], g["wholly_below"])
    assert np.array_equal(below[0], np.array(g["wholly_below"])) and above.shape[0] == 0


finite = st.floats(min_value=-1e3, max_value=1e3, allow_nan=False, allow_infinity=False)


@settings(max_examples=300, deadline=None)
@given(st.tuples(finite, finite, finite), st.tuples(finite, finite, finite))
def test_sphere_zero_or_two_hits_on_surface(o, d):            # FuncTracer.Tests/Geometry/Sphere.fs:18-30
    orc = _sphere_oracle()
    counts, t, p, n = orc.all_hits([o], [d], cap=4)
    assert counts[0] in (0, 2)
    if np.linalg.norm(d) > 1e-3:                              # the on-surface property is skipped upstream only because of zero directions
        for k in range(counts[0]):
            if np.isfinite(t[0, k]) and abs(t[0, k]) * np.linalg.norm(d) < 1e6:
                assert abs(np.linalg.norm(p[0, k]) - 1.0) < 1e-5 * max(1.0, np.linalg.norm(o)) ** 2


_cache = {}


def _sphere_oracle():
    if "sphere" not in _cache:
        o = O.Oracle()
        H.single_prim(o, "sphere", lights=False)
        _cache["sphere"] = o
    return _cache["sphere"]


def test_quadratic_far_root_first(golden):                    # Math.fs:4-10
    for case in golden["hand_derived"]["quadratic"]["cases"]:
        assert O.quadratic(*case["abc"]) == case["roots"]


def test_image_plane_constants(golden):                       # Image.fs:67-81 incl. the resH/resV swap
    import functracer_amd as ft
    g = golden["hand_derived"]["image_plane_1920x1080_fov60"]
    cam = ft.make_camera((0, 2, -2), (0, 0, 3), (0, 1, 0), H.deg(60.0), 1.0)
    ip = O.image_plane(cam, 1920, 1080)
    assert ip["pixel_height"] == pytest.approx(g["pixel_height"], rel=1e-15)
    assert ip["pixel_width"] == pytest.approx(g["pixel_width"], rel=1e-15)
    assert ip["top_left"][0] == pytest.approx(-g["height"] / 2 + g["pixel_width"] / 2, rel=1e-15)
    # left-handed frame: i = up x k, j = k x i (Image.fs:48-53)
    assert np.allclose(np.cross(ip["k"], ip["i"]), ip["j"], atol=1e-15)
    o, d = O.ray_through_pixel(cam, 1920, 1080, 0, 0, 0.0, 0.0)
    assert np.allclose(d, ip["k"] + ip["top_left"][0] * ip["i"] + ip["top_left"][1] * ip["j"], atol=1e-15)


def test_hand_derived_closest(golden):
    for case in golden["hand_derived"]["closest"]:
        o = O.Oracle()
        H.single_prim(o, case["prim"], lights=False)
        hit, t, p, n, _ = o.closest([case["o"]], [case["d"]])
        assert bool(hit[0]) == case["hit"], case["name"]
        if case["hit"]:
            assert t[0] == pytest.approx(case["t"], abs=1e-12), case["name"]
            assert np.allclose(p[0], case["p"], atol=1e-12), case["name"]
            assert np.allclose(n[0], case["n"], atol=1e-12), case["name"]


def test_csg_hollow_shell_walk(golden):                       # Csg.fs:27-33, 59-72
    g = golden["hand_derived"]["csg_hollow_shell"]
    o = O.Oracle()
    o.clear()
    shell = o.subtract(o.scale(11, o.primitive(H.PRIMS["sphere"])), o.scale(10, o.primitive(H.PRIMS["sphere"])))
    o.set_objects(o.group([shell]))
    o.commit()
    counts, t, p, n = o.all_hits([g["o"]], [g["d"]], cap=8)
    assert counts[0] == 4 and np.allclose(t[0, :4], g["all_t"], rtol=1e-14, atol=0)   # 1/10 is inexact: t = -9.999999999999998
    assert np.allclose(n[0, :4], [[0, 0, -1], [0, 0, 1], [0, 0, -1], [0, 0, 1]], atol=1e-15)   # Take, Flip, Flip, Take
    hit, tt, pp, nn, _ = o.closest([g["o"]], [g["d"]])
    assert hit[0] == 1 and tt[0] == pytest.approx(g["t"], rel=1e-14) and np.allclose(pp[0], g["p"]) and np.allclose(nn[0], g["n"])


def test_quantise_truncates(golden):                          # Image.fs:36
    for case in golden["hand_derived"]["quantise"]["cases"]:
        assert list(O.quantise_rgba8(np.array([case["rgb"]]))[0]) == case["bytes"]


def test_reflection_is_added_once_per_light():                # Shading.fs:119-127, 89-98
    """A mirror sphere above a lit unlit-material... two lights double the reflected term."""
    def build(n_lights):
        o = O.Oracle()
        o.clear()
        mirror = o.material(o.primitive(H.PRIMS["sphere"]), colour=(0, 0, 0), reflectance=0.5)
        glow = o.ignore_light(o.material(o.translate((0, 0, -6), o.primitive(H.PRIMS["sphere"])), colour=(0.25, 0.5, 1.0)))
        o.set_objects(o.group([mirror, glow]))
        for _ in range(n_lights):
            o.add_directional((0, 0, 1), (0, 0, 0))          # black lights: only the unlit colour and the reflection contribute
        o.commit()
        return o.colour_for_ray([[0, 0, -3]], [[0, 0, 1]])[0]
    one, two = build(1), build(2)
    # 1 light: 0.5 * glow; 2 lights: reflection traced per light, and the unlit glow colour is itself added per light
    assert np.allclose(one, 0.5 * np.array([0.25, 0.5, 1.0]))
    assert np.allclose(two, 2 * 0.5 * (2 * np.array([0.25, 0.5, 1.0])))


def _shading_cases():
    import json
    import os
    with open(os.path.join(H.ROOT, "tests", "golden", "known_answers.json")) as f:
        return json.load(f)["hand_derived_shading"]["cases"]


@pytest.mark.parametrize("case", _shading_cases(), ids=lambda c: c["name"])
def test_hand_derived_shading(case):                          # Shading.fs:33-139, derivations in tests/tools/derive_shading_answers.py
    H.check_shading_case(O.Oracle(), case)


def test_hand_derived_csg_tables(golden):                     # Csg.fs:19-55, 59-72
    by_op = {}
    for case in golden["hand_derived_csg"]["cases"]:
        by_op.setdefault(case["op"], []).append(case)
    assert sorted(by_op) == ["exclude", "intersect", "subtract", "union"]
    for op, cases in by_op.items():
        o = O.Oracle()
        H.csg_pair(o, op)
        hit, t, p, n, _ = o.closest([c["o"] for c in cases], [c["d"] for c in cases])
        for k, c in enumerate(cases):
            assert bool(hit[k]) == c["hit"], (op, c["why"])
            if c["hit"]:
                assert t[k] == pytest.approx(c["t"], abs=1e-12) and np.allclose(n[k], c["n"], atol=1e-12), (op, c["why"], t[k], n[k])


# ---- round 3: closed forms for what rounds 1-2 pinned by oracle == device alone (tests/tools/derive_round3_answers.py) -------------
@pytest.mark.parametrize("case", H.round3_cases("closest"), ids=lambda c: c["name"])
def test_hand_derived_triangle_and_transformed_normals(case):    # Triangle.fs:43-66, Transform.fs:77-87
    H.check_closest_case(O.Oracle(), case)


@pytest.mark.parametrize("case", H.round3_cases("shading"), ids=lambda c: c["name"])
def test_hand_derived_oren_nayar_textures_soft_shadows(case):    # Shading.fs:24-31, 50-63; Texture.fs:8-29; Sphere.fs:6-10
    H.check_shading_case(O.Oracle(), case)


@pytest.mark.parametrize("case", H.round3_cases("frames"), ids=lambda c: c["name"])
def test_hand_derived_blend_and_corner_average(case):            # Image.fs:83-89, 112-116, 125-145
    H.check_frame_case(O.Oracle(), case)
