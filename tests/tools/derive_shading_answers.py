#!/usr/bin/env python3
"""Hand-derived known answers for the shading half of the path (a13, a16, a17), written into
tests/golden/known_answers.json under "hand_derived_shading" and "hand_derived_csg".

Nothing here runs the oracle, the device path or the reference: every expected value is a closed form worked out on paper from
the cited lines of /root/reference/FuncTracer/*.fs and evaluated below with plain Python floats.  The tests
(tests/test_oracle_golden.py on the CPU, tests/test_gpu_parity.py on the device) build each scene through the scene-builder API,
shade the listed rays with getColourForRay (Shading.fs:131-139) and compare.

Geometry common to most cases: the unit sphere at the origin (Sphere.fs:11-21), ray o = (0,0,-3), d = (0,0,2) (not normalised,
Image.fs:88-89).  slightOffset (Shading.fs:129) moves the origin to (0,0,-2.9998); the far root comes first (Math.fs:10) but
closest takes the smallest t >= 0 (Scene.fs:112-116): p = (0,0,-1), n = normalise p = (0,0,-1) (outward, never flipped:
Sphere.fs:17).  Shadow rays start at p + 1e-4 n = (0,0,-1.0001) (Shading.fs:111).  The view direction is normalise d = (0,0,1).

Shader = shadeIfRequired (multiPartShader [specular; reflection; diffuse]) (Program.fs:59), summed over one fragment per light
(Shading.fs:119-127, 139).  With n = (0,0,-1) and view = (0,0,1):
    diffuse  = ((-L) . n) * (m * lightColour)                    Shading.fs:65-70    (unclamped)
    specular = lightColour * (view . (-r)) ** s,  r = normalise (L - 2 (L.n) n) = (Lx, Ly, -Lz)  =>  base = Lz      Shading.fs:78-87
               black when s <= 0 or the power is <= 0; a NaN power (negative base, fractional s) passes the test and poisons the pixel
    lightColour = intensity * colour, intensity = 0 / 1 (directional, Shading.fs:36) or attenuate falloff distance when unblocked
               (Shading.fs:38-42, Light.fs:16-17), distance measured from the OFFSET point, L from the un-offset one (Shading.fs:48, 116)
"""
import json
import math
import os

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
M = (0.5, 0.25, 1.0)          # material colour of the shaded sphere
LC = (0.8, 0.6, 0.4)          # light colour
RAY = {"o": [0, 0, -3], "d": [0, 0, 2]}


def mul(a, b):
    return [x * y for x, y in zip(a, b)]


def scale(k, a):
    return [k * x for x in a]


def add(a, b):
    return [x + y for x, y in zip(a, b)]


def sphere(colour=M, **mat):
    return {"prim": "sphere", "material": dict(colour=list(colour), **mat)}


cases = []

# ---- Lambert (Shading.fs:65-70) -------------------------------------------------------------------------------------------
cases.append({"name": "lambert_head_on", "cites": "Shading.fs:65-70, 33-36; Light.fs:19-20",
              "derivation": "L = (0,0,1); (-L).n = 1; shadow ray from (0,0,-1.0001) along -L = (0,0,-1) leaves the sphere (both roots negative) -> intensity 1; colour = m * lc",
              "objects": [sphere()], "lights": [{"kind": "directional", "dir": [0, 0, 1], "colour": list(LC)}],
              "rays": [dict(RAY, rgb=mul(M, LC))]})
a = 1.0 / math.sqrt(2.0)
cases.append({"name": "lambert_oblique", "cites": "Shading.fs:65-70; CommonTypes.fs:63-67 (directional normalises its direction)",
              "derivation": "dir (1,0,1) -> L = (a,0,a), a = 1/sqrt 2; (-L).n = a; colour = a * (m * lc)",
              "objects": [sphere()], "lights": [{"kind": "directional", "dir": [1, 0, 1], "colour": list(LC)}],
              "rays": [dict(RAY, rgb=scale(a, mul(M, LC)))]})
c = 1.0 / math.sqrt(1.0 + 1e-6)
graze = {"kind": "directional", "dir": [-1, 0, -0.001], "colour": list(LC)}
graze_note = ("dir (-1,0,-0.001) -> L = (-c,0,-0.001c), c = 1/sqrt(1+1e-6): the light comes from BEHIND the surface at a grazing angle; the shadow ray from "
              "s = (0,0,-1.0001) along -L passes the sphere (b^2 - 4ac = 4.0008e-6 c^2 - 8.0004e-4 < 0), so the light counts in full and (-L).n = -0.001c is NOT clamped")
cases.append({"name": "lambert_negative_is_not_clamped", "cites": "Shading.fs:69 (no max 0), 33-36", "derivation": graze_note + "; colour = -0.001c * (m * lc)",
              "objects": [sphere()], "lights": [graze], "rays": [dict(RAY, rgb=scale(-0.001 * c, mul(M, LC)))]})

# ---- specular (Shading.fs:78-87) ------------------------------------------------------------------------------------------
for name, s, power in (("specular_integral_exponent", 3.0, a ** 3), ("specular_fractional_exponent", 2.5, a ** 2.5)):
    cases.append({"name": name, "cites": "Shading.fs:78-87 (** = Math.Pow); Program.fs:59 (sum order specular, reflection, diffuse)",
                  "derivation": f"dir (1,0,1): base = view.(-r) = Lz = a = 1/sqrt 2; specular = lc * a**{s}; diffuse = a * (m * lc); the material colour does not enter the highlight",
                  "objects": [sphere(shineyness=s)], "lights": [{"kind": "directional", "dir": [1, 0, 1], "colour": list(LC)}],
                  "rays": [dict(RAY, rgb=add(scale(power, LC), scale(a, mul(M, LC))))]})
base = -0.001 * c
cases.append({"name": "specular_negative_base_even_exponent_gives_a_back_side_highlight", "cites": "Shading.fs:85-87", "derivation": graze_note + "; base = Lz = -0.001c; (-0.001c)**2 = 1e-6 c^2 > 0 -> highlight lc * 1e-6 c^2, plus the negative diffuse term",
              "objects": [sphere(shineyness=2.0)], "lights": [graze], "rays": [dict(RAY, rgb=add(scale(base * base, LC), scale(base, mul(M, LC))))]})
cases.append({"name": "specular_negative_base_odd_exponent_is_black", "cites": "Shading.fs:86", "derivation": graze_note + "; (-0.001c)**3 < 0 -> Colour.black; only the diffuse term remains",
              "objects": [sphere(shineyness=3.0)], "lights": [graze], "rays": [dict(RAY, rgb=scale(base, mul(M, LC)))]})
cases.append({"name": "specular_negative_base_fractional_exponent_is_nan", "cites": "Shading.fs:85-87 (Math.Pow of a negative base with a non-integral exponent is NaN; NaN <= 0.0 is false)",
              "derivation": graze_note + "; (-0.001c)**2.5 = NaN, the test `intensity <= 0.0` fails, lightColour * NaN -> every channel NaN",
              "objects": [sphere(shineyness=2.5)], "lights": [graze], "rays": [dict(RAY, rgb=["nan", "nan", "nan"])]})

# ---- point light: attenuation and shadow (Shading.fs:33-42, 44-48; Light.fs:16-17; Scene.fs:119-121) -----------------------
pos, fall = (3.0, 0.0, -4.0), (1.0, 0.5, 0.25)
s_pt = (0.0, 0.0, -1.0001)
dist = math.sqrt((pos[0] - s_pt[0]) ** 2 + (pos[2] - s_pt[2]) ** 2)
att = 1.0 / (fall[0] + dist * (fall[1] + dist * fall[2]))
lit = scale(a, mul(M, scale(att, LC)))
point = {"kind": "point", "pos": list(pos), "falloff": list(fall), "colour": list(LC)}
pt_note = ("light at (3,0,-4), falloff (1,0.5,0.25): distance = |pos - (p + 1e-4 n)| = sqrt(9 + 2.9999^2), intensity = 1/(1 + d(0.5 + 0.25 d)); "
           "L = normalise (p - pos) = (-a,0,a) from the UN-offset p; (-L).n = a; colour = a * (m * (intensity * lc))")
mid = [(pos[0] + s_pt[0]) / 2, 0.0, (pos[2] + s_pt[2]) / 2]
beyond = [pos[0] + (pos[0] - s_pt[0]) / 2, 0.0, pos[2] + (pos[2] - s_pt[2]) / 2]


def blocker(at, **kw):
    return dict({"prim": "sphere", "xf": [["scale", [0.2, 0.2, 0.2]], ["translate", at]], "material": {"colour": [1, 1, 1]}}, **kw)


cases.append({"name": "point_light_attenuation", "cites": "Shading.fs:38-42, 44-48, 109-117; Light.fs:16-17", "derivation": pt_note,
              "objects": [sphere()], "lights": [point], "rays": [dict(RAY, rgb=lit)]})
cases.append({"name": "point_light_blocked_is_black", "cites": "Shading.fs:41; Scene.fs:119-121",
              "derivation": pt_note + "; a sphere of radius 0.2 half way to the light: lightIsBocked -> intensity 0 -> lightColour (0,0,0) -> the fragment is still shaded, a * (m * 0) = 0",
              "objects": [sphere(), blocker(mid)], "lights": [point], "rays": [dict(RAY, rgb=[0.0, 0.0, 0.0])]})
cases.append({"name": "point_light_blocker_beyond_the_light_is_ignored", "cites": "Scene.fs:121 (t < maxDistance)",
              "derivation": pt_note + "; the same sphere half a distance BEHIND the light: its hits have t > distance",
              "objects": [sphere(), blocker(beyond)], "lights": [point], "rays": [dict(RAY, rgb=lit)]})
cases.append({"name": "unlit_blocker_casts_no_shadow", "cites": "Scene.fs:121 (material.applyLighting); Ray.fs:47",
              "derivation": pt_note + "; the blocker half way to the light is ignoreLight: it never blocks",
              "objects": [sphere(), blocker(mid, ignore_light=True)], "lights": [point], "rays": [dict(RAY, rgb=lit)]})

# ---- shadeIfRequired (Shading.fs:100-104) ----------------------------------------------------------------------------------
G = (0.2, 0.3, 0.4)
three = [{"kind": "directional", "dir": [0, -1, 0], "colour": [1, 1, 1]}, {"kind": "directional", "dir": [1, 0, 0], "colour": [0, 1, 0]}, dict(point)]
cases.append({"name": "unlit_colour_is_added_once_per_light", "cites": "Shading.fs:100-104, 119-127, 139",
              "derivation": "applyLighting = false: every fragment (one per light, whatever its intensity) returns material.colour; three lights -> ((0 + g) + g) + g",
              "objects": [dict(sphere(colour=G), ignore_light=True)], "lights": three, "rays": [dict(RAY, rgb=[(0.0 + x + x) + x for x in G])]})
cases.append({"name": "no_lights_no_fragments", "cites": "Shading.fs:109-117, 139 (Seq.sumBy over an empty sequence = Colour.Zero)",
              "derivation": "no lights -> no fragments -> black, lit or not",
              "objects": [dict(sphere(colour=G), ignore_light=True)], "lights": [], "rays": [dict(RAY, rgb=[0.0, 0.0, 0.0])]})

# ---- reflection (Shading.fs:89-98, 119-127, 131-139) -----------------------------------------------------------------------
GLOW = (0.25, 0.5, 1.0)
mirror = {"prim": "sphere", "material": {"colour": [0, 0, 0], "reflectance": 0.5}}
glow = {"prim": "sphere", "xf": [["translate", [0, 0, -6]]], "material": {"colour": list(GLOW)}, "ignore_light": True}
black = {"kind": "directional", "dir": [0, 0, 1], "colour": [0, 0, 0]}
for n in (1, 2, 3):
    cases.append({"name": f"mirror_bounce_{n}_black_light{'s' if n > 1 else ''}", "cites": "Shading.fs:89-98, 119-127; Vector.reflect CommonTypes.fs:72",
                  "derivation": f"mirror sphere (colour 0, reflectance 0.5), unlit glow sphere behind the camera at z = -6, {n} black light(s): reflect n d = d - 2(d.n)n = (0,0,-2); "
                                f"every one of the {n} fragments traces the reflection itself, and the glow is unlit: glow * {n} per trace -> {n} * 0.5 * ({n} * glow)",
                  "objects": [mirror, glow], "lights": [dict(black) for _ in range(n)], "rays": [dict(RAY, rgb=scale(n * 0.5 * n, GLOW))]})

corridor_light = {"kind": "point", "pos": [0, 1, 0], "falloff": [1, 0, 0], "colour": list(LC)}
floor = {"prim": "plane", "material": {"colour": list(M), "reflectance": 0.5}}
ceil_facing = {"prim": "plane", "xf": [["rotate", [1, 0, 0], 180.0], ["translate", [0, 2, 0]]], "material": {"colour": list(M), "reflectance": 0.5}}
ceil_plain = {"prim": "plane", "xf": [["translate", [0, 2, 0]]], "material": {"colour": list(M), "reflectance": 0.5}}
cval = mul(M, LC)
down = {"o": [0, 1.5, 0], "d": [0, -1, 0]}
for depth in (8, 2, 0):
    total = sum(0.5 ** k for k in range(depth + 1))
    cases.append({"name": f"two_facing_mirrors_recursion_limit_{depth}", "max_depth": depth, "cites": "Shading.fs:131-139 (recursionLimit 8 at Shading.fs:142), 89-98; Plane.fs:9-20; Transform.fs:60-71",
                  "derivation": "floor y = 0 and a ceiling at y = 2 turned to face it, both reflectance 0.5, point light (0,1,0) with falloff (1,0,0) half way: every hit sees the light at "
                                "distance 0.9999 unblocked with (-L).n = 1, so each of the depth + 1 rays of the path adds m * lc weighted 0.5^k; the ray at recursion limit 0 is still "
                                f"shaded, only its reflection is black: sum_(k=0..{depth}) 0.5^k = {total}",
                  "objects": [floor, ceil_facing], "lights": [corridor_light], "rays": [dict(down, rgb=scale(total, cval))]})
total = sum(0.25 ** j for j in range(5))
cases.append({"name": "plane_normal_is_not_turned_towards_the_ray", "max_depth": 8, "cites": "Plane.fs:19, 28-33 (normal as given); Shading.fs:111 (shadow origin p + 1e-4 n)",
              "derivation": "as above but the ceiling is a plain translate (0,2,0) plane whose normal (0,1,0) points AWAY from the corridor: its shadow origin (0,2.0001,0) lies above it, the ray to the "
                            "light hits the ceiling itself at t = 1e-4 -> blocked -> ceiling hits add (-1) * (m * 0) = 0 but still reflect (reflect n v is even in n); only the floor hits at depths "
                            f"0,2,4,6,8 count: sum_(j=0..4) 0.25^j = {total}",
              "objects": [floor, ceil_plain], "lights": [corridor_light], "rays": [dict(down, rgb=scale(total, cval))]})

# ---- CSG rule tables (Csg.fs:19-55, 59-72, 74-94) over two unit spheres: A at the origin, B at (0,0,1) ------------------------
#   along +z from (0,0,-3):  t = 2 A-in (z=-1), 3 B-in (z=0), 4 A-out (z=1), 5 B-out (z=2)
#       types: OutsideIntoA, AIntoAB, ABleaveA, BIntoOutside
#   along -z from (0,0,4):   t = 2 B (z=2), 3 A (z=1), 4 B (z=0), 5 A (z=-1)
#       types: OutsideIntoB, BIntoAB, ABleaveB, AIntoOutside
#   from inside both, (0,0,0.5) along +z: t = -1.5 A, -0.5 B, 0.5 A (z=1), 1.5 B (z=2): the negative hits set the state (Csg.fs:81-93)
#       types: OutsideIntoA, AIntoAB, ABleaveA, BIntoOutside
# closest = smallest t >= 0 among the kept hits (Scene.fs:112-116); Flip negates the normal (Csg.fs:90).
Z = [0, 0, 1]
NZ = [0, 0, -1]
csg = []
front, back, inside = {"o": [0, 0, -3], "d": [0, 0, 1]}, {"o": [0, 0, 4], "d": [0, 0, -1]}, {"o": [0, 0, 0.5], "d": [0, 0, 1]}
table = {
    "union":     [(front, 2.0, NZ, "Take OutsideIntoA"), (back, 2.0, Z, "Take OutsideIntoB"), (inside, 1.5, Z, "inside hits discarded (AIntoAB, ABleaveA); Take BIntoOutside")],
    "intersect": [(front, 3.0, NZ, "OutsideIntoA discarded; Take AIntoAB: B's surface at z = 0"), (back, 3.0, Z, "Take BIntoAB: A's surface at z = 1"), (inside, 0.5, Z, "Take ABleaveA at z = 1")],
    "subtract":  [(front, 2.0, NZ, "Take OutsideIntoA"), (back, 4.0, Z, "OutsideIntoB, BIntoAB discarded; Flip ABleaveB: B's normal (0,0,-1) at z = 0 turned"),
                  (inside, None, None, "Take OutsideIntoA (t < 0), Flip AIntoAB (t < 0), ABleaveA and BIntoOutside discarded: nothing at t >= 0")],
    "exclude":   [(front, 2.0, NZ, "Take OutsideIntoA"), (back, 2.0, Z, "Take OutsideIntoB"), (inside, 0.5, NZ, "Flip ABleaveA: A's normal (0,0,1) at z = 1 turned")],
}
for op, rows in table.items():
    for ray, t, n, why in rows:
        csg.append({"op": op, "o": ray["o"], "d": ray["d"], "hit": t is not None, "t": t, "n": n, "why": why})

path = os.path.join(ROOT, "tests", "golden", "known_answers.json")
with open(path) as f:
    doc = json.load(f)
doc["hand_derived_shading"] = {"_comment": "closed forms derived from the cited lines, evaluated by tests/tools/derive_shading_answers.py (no oracle, no device, no reference run); "
                                           "checked by getColourForRay on the oracle (CPU tests) and on the device (-m gpu)", "cases": cases}
doc["hand_derived_csg"] = {"_comment": "closest hit of A (unit sphere at the origin) op B (unit sphere at (0,0,1)) for one value per rule table and ray, Csg.fs:19-55, 59-72", "cases": csg}
with open(path, "w") as f:
    json.dump(doc, f, indent=1)
    f.write("\n")
print(len(cases), "shading cases,", len(csg), "csg cases")
