#!/usr/bin/env python3
"""Hand-derived known answers for the rows of SURVEY 8(a)/(f) that round 2 still pinned by oracle == device alone, written into
tests/golden/known_answers.json under "hand_derived_round3":

  a10  Moller-Trumbore: t, p = o + normalise d * (t * |d|), the UNFLIPPED winding normal, the t > 1e-7 cut      Triangle.fs:43-66
  a3   a hit normal through the inverse transpose under a non-uniform scale followed by a rotation              Transform.fs:77-87
  f1   Oren-Nayar at roughness 0.5                                                                               Shading.fs:50-63
  f1   grid texture cells incl. negative coordinates and an exact 0.5, texture scale / rotate, sphere uv        Texture.fs:8-29, Plane.fs:28-30, Sphere.fs:6-10
  a20  the soft-shadow fraction (samples - occluded) / samples on the documented seeded stream                  Shading.fs:24-31, Jitter.fs:15-39
  a2   blendPixels = mean of the samples in sample order, and CornerSampling's four-corner average              Image.fs:83-89, 112-116, 125-145

Nothing here runs the oracle, the device path or the reference: every value is a closed form worked out from the cited lines of
/root/reference/FuncTracer/*.fs and evaluated with plain Python floats (the seeded stream is re-implemented here from its written
definition, DESIGN.md 2: a third, independent implementation).  tests/test_oracle_golden.py checks the oracle against them on the
CPU, tests/test_gpu_parity.py the device (-m gpu).
"""
import json
import math
import os

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
M = (0.5, 0.25, 1.0)
LC = (0.8, 0.6, 0.4)


def mul(a, b):
    return [x * y for x, y in zip(a, b)]


def scale(k, a):
    return [k * x for x in a]


closest, shading, frames = [], [], []

# ---- a10: Triangle.triangle (Triangle.fs:43-66) --------------------------------------------------------------------------------
# v0 = (0,0,0), v1 = (1,0,0), v2 = (0,1,0); ray o = (0.25,0.25,-2), d = (0,0,4) (|d| = 4, not normalised):
#   edge1 = (1,0,0), edge2 = (0,1,0); h = d x edge2 = (0*0 - 4*1, 4*0 - 0*0, 0*1 - 0*0) = (-4,0,0); a = edge1.h = -4; f = -1/4
#   s = o - v0 = (0.25,0.25,-2); u = f (s.h) = -1/4 * -1 = 0.25; q = s x edge1 = (0.25*0 - (-2)*0, (-2)*1 - 0.25*0, 0.25*0 - 0.25*1) = (0,-2,-0.25)
#   v = f (d.q) = -1/4 * (4 * -0.25) = 0.25; u + v = 0.5 <= 1; t = f (edge2.q) = -1/4 * -2 = 0.5 > 1e-7
#   p = o + normalise d * (t * |d|) = (0.25,0.25,-2) + (0,0,1) * 2 = (0.25,0.25,0); n = normalise (edge1 x edge2) = (0,0,1):
#   the ray travels along +z and the normal points along +z too - it is NOT turned towards the ray (Triangle.fs:64).
TRI = [[0, 0, 0], [1, 0, 0], [0, 1, 0]]
closest.append({"name": "triangle_hit_unflipped_normal", "cites": "Triangle.fs:43-66", "objects": [{"triangle": TRI}],
                "o": [0.25, 0.25, -2], "d": [0, 0, 4], "hit": True, "t": 0.5, "p": [0.25, 0.25, 0.0], "n": [0, 0, 1]})
closest.append({"name": "triangle_hit_from_the_other_side_keeps_the_winding_normal", "cites": "Triangle.fs:63-64",
                "objects": [{"triangle": TRI}], "o": [0.25, 0.25, 3], "d": [0, 0, -2], "hit": True, "t": 1.5, "p": [0.25, 0.25, 0.0], "n": [0, 0, 1]})
closest.append({"name": "triangle_t_below_epsilon_is_no_hit", "cites": "Triangle.fs:62 (t > 0.0000001)",
                "objects": [{"triangle": TRI}], "o": [0.25, 0.25, -0.5e-7], "d": [0, 0, 1], "hit": False})
closest.append({"name": "triangle_t_just_above_epsilon_is_a_hit", "cites": "Triangle.fs:62", "objects": [{"triangle": TRI}],
                "o": [0.25, 0.25, -2e-7], "d": [0, 0, 1], "hit": True, "t": 2e-7, "p": [0.25, 0.25, 0.0], "n": [0, 0, 1], "p_atol": 1e-15})
closest.append({"name": "triangle_behind_the_origin_is_no_hit", "cites": "Triangle.fs:62 (no negative t, unlike every other primitive)",
                "objects": [{"triangle": TRI}], "o": [0.25, 0.25, 1], "d": [0, 0, 1], "hit": False})
closest.append({"name": "triangle_outside_u_plus_v", "cites": "Triangle.fs:58-59 (u + v > 1)", "objects": [{"triangle": TRI}],
                "o": [0.75, 0.75, -1], "d": [0, 0, 1], "hit": False})

# ---- a3: normals through the inverse transpose (Transform.fs:77-87) -------------------------------------------------------------
# unit sphere under Composed [scale (2,1,1); rotate Z 90 deg] (first listed applied first, Transform.fs:70-71): M = R S,
# R (x,y,z) = (-y,x,z).  Model point pm = (a,a,0), a = 1/sqrt 2, model normal nm = pm.  World point R S pm = R (2a,a,0) = (-a,2a,0).
# normalToWorld = transpose (matrix (inverse t)) = (S^-1 R^-1)^T = R S^-1 (R orthogonal, S diagonal):
#   n = normalise (R (a/2, a, 0)) = normalise (-a, a/2, 0) = (-1, 1/2, 0) / sqrt 1.25          (NOT the direction of R S nm = (-a, 2a, 0))
# ray along the inward normal through that point: d = (1,-1/2,0), o = p - 2 d: enters the ellipsoid there at t = 2.
a = 1.0 / math.sqrt(2.0)
pw = [-a, 2 * a, 0.0]
dn = [1.0, -0.5, 0.0]
closest.append({"name": "normal_under_scale_then_rotate_uses_the_inverse_transpose", "cites": "Transform.fs:47-51, 70-71, 77-87",
                "objects": [{"prim": "sphere", "xf": [["scale", [2, 1, 1]], ["rotate", [0, 0, 1], 90.0]]}],
                "o": [pw[0] - 2 * dn[0], pw[1] - 2 * dn[1], 0.0], "d": dn, "hit": True, "t": 2.0, "p": pw,
                "n": [-1.0 / math.sqrt(1.25), 0.5 / math.sqrt(1.25), 0.0]})
# the same primitive hit on its long axis: p = R S (1,0,0) = (0,2,0), n = normalise (R (1/2,0,0)) = (0,1,0)
closest.append({"name": "normal_under_scale_then_rotate_on_the_long_axis", "cites": "Transform.fs:77-87",
                "objects": [{"prim": "sphere", "xf": [["scale", [2, 1, 1]], ["rotate", [0, 0, 1], 90.0]]}],
                "o": [0, 5, 0], "d": [0, -1.5, 0], "hit": True, "t": 2.0, "p": [0, 2, 0], "n": [0, 1, 0]})

# ---- f1: Oren-Nayar (Shading.fs:50-63) ----------------------------------------------------------------------------------------------
# plane y = 0 (n = (0,1,0), never flipped), view ray o = (0,1,-1), d = (0,-1,1): hits the origin; -d makes 45 deg with n:
#   rayAngle = 45 deg, tangentRay = normalise (perpendicularComponent n (-d)) = (0,0,-1).
# light: -L = cos b n + sin b (sin 60, 0, -cos 60) with b = 30 deg: lightAngle = 30 deg, tangentLight.tangentRay = cos 60 = 1/2.
#   sigma^2 = roughness ** 2 = 0.25; A = 1 - 0.5*0.25/(0.25+0.33); B = 0.45*0.25/(0.25+0.09); alpha = 45 deg, beta = 30 deg
#   intensity = cos 30 * (A + B * 1/2 * sin 45 * tan 30); colour = intensity * material colour - the light's colour is not used (sic, :63)
# specular: shineyness 0 -> black; reflectance 0.
def oren_nayar(rough, ray_angle, light_angle, cos_phi):
    s2 = rough ** 2
    A = 1.0 - 0.5 * s2 / (s2 + 0.33)
    B = 0.45 * s2 / (s2 + 0.09)
    alpha, beta = max(ray_angle, light_angle), min(ray_angle, light_angle)
    return math.cos(light_angle) * (A + B * max(0.0, cos_phi) * math.sin(alpha) * math.tan(beta))


for name, phi, note in (("oren_nayar_light_and_view_60_degrees_apart_in_azimuth", 60.0, "tangentLight.tangentRay = 1/2"),
                        ("oren_nayar_back_scatter_term_is_clamped_at_zero", 120.0, "tangentLight.tangentRay = -1/2 -> max 0.0: only the A term")):
    b = math.radians(30.0)
    tl = [math.sin(math.radians(phi)), 0.0, -math.cos(math.radians(phi))]
    to_light = [math.sin(b) * tl[0], math.cos(b), math.sin(b) * tl[2]]
    k = oren_nayar(0.5, math.radians(45.0), b, math.cos(math.radians(phi)))
    shading.append({"name": name, "cites": "Shading.fs:50-63, 72-76; CommonTypes.fs:74-79", "derivation": "plane, view 45 deg and light 30 deg off the normal, " + note,
                    "objects": [{"prim": "plane", "material": {"colour": list(M), "roughness": 0.5}}],
                    "lights": [{"kind": "directional", "dir": [-x for x in to_light], "colour": list(LC)}],
                    "rays": [{"o": [0, 1, -1], "d": [0, -1, 1], "rgb": scale(k, M)}]})
# view along the normal (sphere, head on): rayAngle = 0 -> beta = 0, tan 0 = 0 and tangentRay = normalise 0 = 0: intensity = cos lightAngle * A
k = oren_nayar(0.5, 0.0, math.radians(45.0), 0.0)
shading.append({"name": "oren_nayar_view_along_the_normal_leaves_the_A_term", "cites": "Shading.fs:50-63; CommonTypes.fs:63-67 (normalise of a zero vector is the zero vector)",
                "derivation": "unit sphere hit head on (n = (0,0,-1)), light dir (1,0,1): lightAngle 45 deg; colour = cos 45 * A * m",
                "objects": [{"prim": "sphere", "material": {"colour": list(M), "roughness": 0.5}}],
                "lights": [{"kind": "directional", "dir": [1, 0, 1], "colour": list(LC)}],
                "rays": [{"o": [0, 0, -3], "d": [0, 0, 2], "rgb": scale(k, M)}]})

# ---- f1: grid texture, uv, texture functions (Texture.fs:8-29; Plane.fs:28-30; Sphere.fs:6-10; Ray.fs:57-59) -----------------------
# plane y = 0 textured grid c1 c2, uv = (p.x, p.z); light straight down: (-L).n = 1 -> colour = cell * lc.
# repeat x = |x - floor x| (the flipNegative branch can never fire): -0.25 -> 0.75.  grid: (u<.5 & v<.5) c1 | (u<.5) c2 | (u>.5 & v>.5) c1 | else c2,
# so a coordinate of exactly 0.5 falls through to c2 whatever the other one is.
C1, C2 = (1.0, 0.5, 0.25), (0.125, 0.25, 0.5)
down = {"kind": "directional", "dir": [0, -1, 0], "colour": list(LC)}
cells = [((0.25, 0.25), C1, "both < 0.5"), ((-0.25, 0.25), C2, "u = -0.25 -> 0.75: u > 0.5, v < 0.5"), ((-0.25, -0.25), C1, "(0.75, 0.75)"),
         ((0.25, 0.75), C2, "u < 0.5, v > 0.5"), ((0.5, 0.75), C2, "u = 0.5 exactly: neither < nor >"), ((0.75, 0.75), C1, "both > 0.5"), ((-1.75, 3.25), C1, "(0.25, 0.25) two and three cells away")]
shading.append({"name": "grid_texture_cells_on_a_plane", "cites": "Texture.fs:8-12, 24-29; Plane.fs:28-30; Ray.fs:57-59",
                "derivation": "; ".join(f"uv {uv}: {why}" for uv, _, why in cells),
                "objects": [{"prim": "plane", "material": {"colour": [9, 9, 9]}, "texture": {"grid": [list(C1), list(C2)], "ops": []}}], "lights": [down],
                "rays": [{"o": [uv[0], 1, uv[1]], "d": [0, -2, 0], "rgb": mul(c, LC)} for uv, c, _ in cells]})
# Texture.scale (2,4): (u,v) -> (u/2, v/4) before the grid (Texture.fs:14-16): (1.5, 1.0) -> (0.75, 0.25) -> c2; (0.5, 3.0) -> (0.25, 0.75) -> c2; (0.5,1.0) -> (.25,.25) -> c1
sc = [((1.5, 1.0), C2), ((0.5, 3.0), C2), ((0.5, 1.0), C1), ((1.5, 3.0), C1)]
shading.append({"name": "texture_scale_divides_the_coordinates", "cites": "Texture.fs:14-16; Scene.fs:68-75",
                "derivation": "scale (2,4): (u/2, v/4)", "objects": [{"prim": "plane", "texture": {"grid": [list(C1), list(C2)], "ops": [["scale", 2, 4]]}}], "lights": [down],
                "rays": [{"o": [uv[0], 1, uv[1]], "d": [0, -2, 0], "rgb": mul(c, LC)} for uv, c in sc]})
# Texture.rotate 90 deg: matrix (rotate Y 90) * (u,0,v) = (c u + s v, 0, -s u + c v) = (v, -u) (Texture.fs:18-22, Transform.fs:60-69):
#   (0.25, 0.75) -> (0.75, -0.25) -> repeat (0.75, 0.75) -> c1 (unrotated it is c2); (0.25, 0.25) -> (0.25, -0.25) -> (0.25, 0.75) -> c2 (unrotated c1)
ro = [((0.25, 0.75), C1), ((0.25, 0.25), C2)]
shading.append({"name": "texture_rotate_turns_the_coordinates_about_y", "cites": "Texture.fs:18-22; Transform.fs:60-69",
                "derivation": "rotate 90 deg: (u,v) -> (v,-u)", "objects": [{"prim": "plane", "texture": {"grid": [list(C1), list(C2)], "ops": [["rotate", 90.0]]}}], "lights": [down],
                "rays": [{"o": [uv[0], 1, uv[1]], "d": [0, -2, 0], "rgb": mul(c, LC)} for uv, c in ro]})
# Sphere.setUV (Sphere.fs:6-10): u = 0.5 + atan2 (n.z, n.x) / 2pi, v = 0.5 - asin n.y / pi.  Ray o = (0, 0.6, -3), d = (0,0,1) hits n = (0, 0.6, -0.8):
#   u = 0.5 - 0.25 = 0.25, v = 0.5 - asin 0.6 / pi = 0.2951... -> c1; the light along +z: (-L).n = ... L = (0,0,1): (-L).n = 0.8 -> 0.8 * (c1 * lc).
#   Ray o = (0,-0.6,-3): n = (0,-0.6,-0.8): v = 0.7048 -> (u < .5) -> c2.
shading.append({"name": "sphere_uv", "cites": "Sphere.fs:6-10, 17-19", "derivation": "n = (0, +-0.6, -0.8): u = 0.25, v = 0.5 -+ asin 0.6 / pi",
                "objects": [{"prim": "sphere", "texture": {"grid": [list(C1), list(C2)], "ops": []}}],
                "lights": [{"kind": "directional", "dir": [0, 0, 1], "colour": list(LC)}],
                "rays": [{"o": [0, 0.6, -3], "d": [0, 0, 1], "rgb": scale(0.8, mul(C1, LC))}, {"o": [0, -0.6, -3], "d": [0, 0, 1], "rgb": scale(0.8, mul(C2, LC))}]})

# ---- a20: softShadowLightIntensity (Shading.fs:24-31) on the seeded stream ------------------------------------------------------------
MASK = (1 << 64) - 1


def sm64(z):
    z = (z + 0x9E3779B97F4A7C15) & MASK
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & MASK
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & MASK
    return z ^ (z >> 31)


def stream(seed, sample, depth, light, purpose):
    key = sm64(sm64(sm64(seed ^ sample) ^ ((depth << 32) | (light << 8) | purpose)))
    n = 0
    while True:
        yield (sm64((key + n) & MASK) >> 11) * (1.0 / 9007199254740992.0)
        n += 1


def circle(gen):                                                   # Jitter.circle (Jitter.fs:15-21): rejection from [-1,1]^2
    while True:
        x, y = 2.0 * next(gen) - 1.0, 2.0 * next(gen) - 1.0
        if x * x + y * y > 1.0:
            continue
        return x, y


# floor y = 0 lit by a softdirectional light straight down with 8 samples, scatter 10 deg; a 10 x 10 square at y = 1 covering x in [0,10],
# z in [-5,5]: its edge x = 0 passes right above the shaded point (0,0,0).  jitterVector (Jitter.fs:26-39) around -direction = (0,1,0):
# generator = X (normalised.x <= 0.9), i = normalise (X x (0,1,0)) = (0,0,1), j = i x (0,1,0) = (-1,0,0); sample k has direction
# normalise ((0,1,0) + m x_k i + m y_k j) = normalise (-m y_k, 1, m x_k), m = tan 5 deg: from (0,1e-4,0) it crosses y = 1 at x = -m y_k (1 - 1e-4),
# |z| < 0.09: inside the square iff y_k <= 0.  occluded = #{k : y_k <= 0}; intensity = (8 - occluded) / 8 (Shading.fs:31); colour = intensity * (m * lc).
# ft_debug_colour / fto_colour_for_ray key the stream with seed 0, sample = ray index, depth 0, light 0, purpose 1.
soft = {"kind": "soft", "dir": [0, -1, 0], "samples": 8, "scatter": 10.0, "colour": list(LC)}
rays, notes = [], []
for ray_index in range(6):
    g = stream(0, ray_index, 0, 0, 1)
    ys = [circle(g)[1] for _ in range(8)]
    occluded = sum(1 for y in ys if y <= 0.0)
    assert all(abs(y) > 1e-6 for y in ys)                            # no draw sits on the edge
    rays.append({"o": [0, 0.5, 0], "d": [0, -1, 0], "rgb": scale((8 - occluded) / 8.0, mul(M, LC))})
    notes.append(f"ray {ray_index}: {occluded} of 8 occluded")
assert len({tuple(r["rgb"]) for r in rays}) > 1                      # the six rays do not all see the same fraction
shading.append({"name": "soft_shadow_fraction_on_the_seeded_stream", "cites": "Shading.fs:24-31; Jitter.fs:9-39; DESIGN.md 2 (stream definition)",
                "derivation": "occluded = #{k : y_k <= 0} of the ray's own 8 draws; " + "; ".join(notes),
                "objects": [{"prim": "plane", "material": {"colour": list(M)}},
                            {"prim": "square", "xf": [["scale", [10, 1, 10]], ["translate", [0, 1, -5]]], "material": {"colour": [1, 1, 1]}}],
                "lights": [soft], "rays": rays})

# ---- a2: blendPixels and CornerSampling (Image.fs:48-53, 67-89, 112-116, 125-145) ---------------------------------------------------------
# camera at the origin looking along +z, up +y, fov 90 deg, aspect 1, resolution 2 x 2: k = (0,0,1), i = up x k = (1,0,0), j = k x i = (0,1,0);
# height = 2 tan 45 = 2, width = 2, pixelHeight = height / (resH - 1) = 2, pixelWidth = width / (resV - 1) = 2, topLeft = (-1 + 1, 1 - 1) = (0,0).
# rayThroughPixel (px,py) (jx,jy): d = k + (2 px + 2 jx) i + (-2 py + 2 jy) j.
# Unlit spheres (applyLighting = false: the fragment's colour is the material colour, once per light; ONE black light) of radius 0.5:
#   RED at (0,0,5) is hit by d = (0,0,1); GREEN at (5,0,5) by d = (1,0,1); nothing lies along d = (0,1,1).
RED, GREEN = (1.0, 0.25, 0.125), (0.25, 1.0, 0.5)
glow = lambda c, at: {"prim": "sphere", "xf": [["scale", [0.5, 0.5, 0.5]], ["translate", list(at)]], "material": {"colour": list(c)}, "ignore_light": True}
cam = {"o": [0, 0, 0], "look_at": [0, 0, 1], "up": [0, 1, 0], "fov_deg": 90.0, "aspect": 1.0}
third = [(RED[c] + GREEN[c] + 0.0) / 3.0 for c in range(3)]        # sum from Zero in sample order, then DivideByInt (CommonTypes.fs:43-48)
frames.append({"name": "blend_is_the_mean_of_the_samples", "cites": "Image.fs:83-89, 100-116; CommonTypes.fs:43-48",
               "derivation": "pattern [(0,0); (0.5,0); (0,0.5)]: pixel (0,0) sees d = (0,0,1) RED, (1,0,1) GREEN, (0,1,1) nothing -> (RED + GREEN + 0) / 3; "
                             "pixel (1,0): d = (2,0,1), (3,0,1), (2,1,1); pixel (0,1): d = (0,-2,1), (1,-2,1), (0,-1,1); pixel (1,1): all miss",
               "camera": cam, "res": [2, 2], "spp": 3, "jitter": [[0, 0], [0.5, 0], [0, 0.5]],
               "objects": [glow(RED, (0, 0, 5)), glow(GREEN, (5, 0, 5))], "lights": [{"kind": "directional", "dir": [0, 0, 1], "colour": [0, 0, 0]}],
               "frame": [[third, [0, 0, 0]], [[0, 0, 0], [0, 0, 0]]]})
# CornerSampling: one ray per pixel CORNER (cx,cy), cx,cy in 0..2, jitter (-0.5,+0.5): d = k + (2 cx - 1) i + (-2 cy + 1) j; pixel (x,y) = average of
# corners (x,y), (x+1,y), (x,y+1), (x+1,y+1) (Seq.average: sum / 4).  RED at (5,5,5) is hit by corner (1,0): d = (1,1,1) only:
#   pixels (0,0) and (1,0) share that corner -> RED / 4 each; the lower row sees nothing.
quarter = [c / 4.0 for c in RED]
frames.append({"name": "corner_sampling_averages_the_four_corners", "cites": "Image.fs:125-145",
               "derivation": "corner (1,0) -> d = (1,1,1) hits RED; every other corner ray misses",
               "camera": cam, "res": [2, 2], "spp": 0, "jitter": [],
               "objects": [glow(RED, (5, 5, 5))], "lights": [{"kind": "directional", "dir": [0, 0, 1], "colour": [0, 0, 0]}],
               "frame": [[quarter, quarter], [[0, 0, 0], [0, 0, 0]]]})

path = os.path.join(ROOT, "tests", "golden", "known_answers.json")
with open(path) as f:
    doc = json.load(f)
doc["hand_derived_round3"] = {"_comment": "closed forms derived from the cited lines, evaluated by tests/tools/derive_round3_answers.py (no oracle, no device, no reference run)",
                              "closest": closest, "shading": shading, "frames": frames}
with open(path, "w") as f:
    json.dump(doc, f, indent=1)
    f.write("\n")
print(len(closest), "closest cases,", len(shading), "shading cases,", len(frames), "frame cases")
