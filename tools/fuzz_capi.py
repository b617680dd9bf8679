#!/usr/bin/env python3
"""Random calls into the scene-graph half of the C ABI on a host-only context: valid and invalid node ids, degenerate / NaN /
huge operands, empty meshes, deep nesting - every call must return a status, never crash.  Run under the sanitized libraries of
tools/sanitize_host.sh (see tools/fuzz_parsers.py for the environment):  python tools/fuzz_capi.py [seconds]"""
import os, random, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import functracer_amd as ft  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 30.0
rng = random.Random(99)
nrng = np.random.default_rng(99)
odd = [0.0, -0.0, 1.0, -1.0, 1e-300, 1e300, float("inf"), float("-inf"), float("nan"), 3.5, -2.25]


def num():
    return rng.choice(odd) if rng.random() < 0.25 else rng.uniform(-4, 4)


def vec():
    return (num(), num(), num())


def tris(n):
    t = nrng.normal(size=(n, 9))
    if n and rng.random() < 0.3: t[rng.randrange(n)] = 0.0                       # a degenerate triangle
    if n and rng.random() < 0.2: t[rng.randrange(n), rng.randrange(9)] = rng.choice(odd)
    return t


ctx = ft.Context(host_only=True)
stats = {"graphs": 0, "committed": 0, "refused": 0}
t_end = time.time() + budget
while time.time() < t_end:
    ctx.clear()
    nodes = []

    def pick():
        if not nodes: nodes.append(ctx.primitive(ft.SPHERE))
        if rng.random() < 0.03: return rng.choice([-1, 10 ** 6, -2 ** 31, 2 ** 31 - 1])   # not a node
        return rng.choice(nodes)

    try:
        for _ in range(rng.randint(1, 25)):
            k = rng.randrange(12)
            if k == 0: n = ctx.primitive(rng.choice([ft.SPHERE, ft.PLANE, ft.CUBE, ft.CONE, ft.CYLINDER, 0, 99, -3]))
            elif k == 1: n = ctx.triangle(vec(), vec(), vec())
            elif k == 2: n = ctx.bsp_mesh(rng.choice([0, 1, 3, 6, -1, 40]), tris(rng.choice([0, 1, 2, 7, 40])))
            elif k == 3: n = ctx.translate(vec(), pick())
            elif k == 4: n = ctx.scale(rng.choice([num(), vec()]), pick())
            elif k == 5: n = ctx.rotate(vec(), num(), pick())
            elif k == 6: n = ctx.material(pick(), colour=vec(), roughness=num(), reflectance=num(), shineyness=num(), apply_lighting=rng.random() < 0.8)
            elif k == 7: n = ctx.group([pick() for _ in range(rng.randint(0, 5))])
            elif k == 8: n = ctx.csg(rng.choice([0, 1, 2, 3, 4, -1, 17]), pick(), pick())
            elif k == 9: n = ctx.texture_grid(vec(), vec(), [(rng.randrange(-1, 4), num(), num()) for _ in range(rng.randint(0, 7))], pick())
            elif k == 10: n = ctx.hue_shift(num(), pick())
            else: n = ctx.texture_image(nrng.integers(0, 256, size=(rng.randint(1, 4), rng.randint(1, 4), 3), dtype=np.uint8), [], pick())
            nodes.append(n)
        ctx.set_objects(pick())
        for _ in range(rng.randint(0, 3)):
            r = rng.randrange(3)
            if r == 0: ctx.add_directional(vec(), vec())
            elif r == 1: ctx.add_soft_directional(vec(), rng.choice([0, 1, 4, -2, 1000]), num(), vec())
            else: ctx.add_positional(vec(), vec(), vec())
        stats["graphs"] += 1
        ctx.commit()
        stats["committed"] += 1
        ctx.scene_info()
    except (ft.FtError, ValueError, TypeError, OverflowError):
        stats["refused"] += 1
print("survived:", stats)
