"""Random scene graphs through both builders: the device path against the CPU oracle on explicit rays and small frames.
Seeded, so a failure names its scene.  Complements the structured cases of test_gpu_parity.py: nesting, grouping and
material stacking are drawn at random here, which is where the item table, the fused CSG pairs (top level and nested),
the wave-level culls and the tail kernel meet combinations nobody wrote down."""
import numpy as np
import pytest

import functracer_amd as ft
from oracle import ft_oracle_py as O

from . import helpers as H

pytestmark = pytest.mark.gpu

PRIMS = [ft.SPHERE, ft.CUBE, ft.CONE, ft.CYLINDER, ft.SOLID_CYLINDER, ft.CIRCLE, ft.SQUARE]
OPS = [ft.UNION, ft.INTERSECT, ft.SUBTRACT, ft.EXCLUDE]


class SceneRecipe:
    """A random scene as a list of builder calls, replayable on any builder (oracle, device)."""

    def __init__(self, seed, ground=True):
        self.ground = ground                                      # False: no ground plane, so that pixel blocks can see nothing
        self.rng = np.random.default_rng(seed)
        self.calls = []          # (method, args, kwargs) with node references as ("n", index)
        root = self._group(depth=0, top=True)
        self.calls.append(("set_objects", (root,), {}))
        r = self.rng
        for _ in range(int(r.integers(1, 4))):
            kind = r.integers(0, 3)
            colour = tuple(r.uniform(0.3, 1.0, size=3))
            if kind == 0:
                self.calls.append(("add_directional", (tuple(r.normal(size=3) + (0, -1.5, 0)), colour), {}))
            elif kind == 1:
                self.calls.append(("add_positional", (tuple(r.uniform(-6, 6, size=3) + (0, 6, 0)), (1.0, float(r.uniform(0, 0.05)), float(r.uniform(0, 0.02))), colour), {}))
            else:
                self.calls.append(("add_soft_directional", (tuple(r.normal(size=3) + (0, -1.5, 0)), int(r.integers(1, 4)), float(r.uniform(0.01, 0.2)), colour), {}))

    def _emit(self, method, *args, **kwargs):
        self.calls.append((method, args, kwargs))
        return ("n", len(self.calls) - 1)

    def _xf(self, node, spread):
        r = self.rng
        ops = []
        if r.random() < 0.7:
            ops.append(("scale", tuple(r.uniform(0.4, 1.6, size=3)) if r.random() < 0.5 else float(r.uniform(0.4, 1.6))))
        if r.random() < 0.6:
            ops.append(("rotate", tuple(r.normal(size=3)), float(r.uniform(-3, 3))))
        ops.append(("translate", tuple(r.uniform(-spread, spread, size=3))))
        return self._emit("transform", ops, node)

    def _material(self, node):
        r = self.rng
        if r.random() < 0.15:
            node = self._emit("texture_grid", tuple(r.uniform(0, 1, size=3)), tuple(r.uniform(0, 1, size=3)), [(0, 0.7, 1.3)] if r.random() < 0.5 else [], node)
        node = self._emit("material", node, colour=tuple(r.uniform(0.1, 1.0, size=3)), roughness=float(r.choice([0.0, 0.0, 0.3])),
                          reflectance=float(r.choice([0.0, 0.0, 0.3, 0.6])), shineyness=float(r.choice([0.0, 5.0, 10.0, 2.5])))
        if r.random() < 0.1:
            node = self._emit("ignore_light", node)
        if r.random() < 0.1:
            node = self._emit("hue_shift", 0.0, node)
        return node

    def _mesh(self):
        """A bumpy closed-ish fan of triangles around the origin: enough of them (>= 8) for the device-side BVH at depth 0."""
        r = self.rng
        n = int(r.integers(10, 40))
        ring = [(np.cos(a) * (0.7 + 0.3 * r.random()), np.sin(a) * (0.7 + 0.3 * r.random()), 0.3 * r.normal()) for a in np.linspace(0, 2 * np.pi, n, endpoint=False)]
        top, bottom = (0.05 * r.normal(), 0.05 * r.normal(), 0.8), (0.05 * r.normal(), 0.05 * r.normal(), -0.8)
        tris = []
        for k in range(n):
            a, b = ring[k], ring[(k + 1) % n]
            tris.append([a, b, top]); tris.append([b, a, bottom])
        return self._emit("bsp_mesh", int(r.choice([0, 0, 2, 4])), np.array(tris, dtype=np.float64).reshape(-1, 9).tolist())

    def _solid(self, depth):
        r = self.rng
        if r.random() < (0.15 if depth == 0 else 0.04):
            return self._xf(self._mesh(), 0.4)
        if depth >= 3 or r.random() < 0.55:
            return self._xf(self._emit("primitive", int(r.choice(PRIMS))), 0.4)
        a, b = self._solid(depth + 1), self._solid(depth + 1)
        node = self._emit("csg", int(r.choice(OPS)), a, b)
        return self._xf(node, 0.3) if r.random() < 0.5 else node

    def _group(self, depth, top=False):
        r = self.rng
        kids = []
        for _ in range(int(r.integers(3, 9) if top else r.integers(1, 4))):
            if depth < 2 and r.random() < 0.2:
                kids.append(self._xf(self._group(depth + 1), 2.0))
            else:
                kids.append(self._material(self._xf(self._solid(0), 2.5)))
        if top and self.ground and r.random() < 0.6:
            kids.append(self._material(self._emit("translate", (0.0, -3.0, 0.0), self._emit("primitive", ft.PLANE))))
        return self._emit("group", kids)

    def build(self, b):
        b.clear()
        made = {}

        def ref(x):
            if isinstance(x, tuple) and len(x) == 2 and x[0] == "n":
                return made[x[1]]
            if isinstance(x, list) and x and isinstance(x[0], tuple) and x[0][0] == "n":
                return [made[i] for _, i in x]
            return x

        for k, (method, args, kwargs) in enumerate(self.calls):
            made[k] = getattr(b, method)(*[ref(a) for a in args], **kwargs)
        b.commit()


@pytest.mark.parametrize("seed", range(32))
def test_random_scene_matches_oracle(hip, seed):
    hip.set_option("csg_mesh_capacity", 8)                        # hit-list entries a mesh may add under CSG: these fans are crossed at most 4 times
    try:
        _check_random_scene(hip, seed)
    finally:
        hip.set_option("csg_mesh_capacity", 32)                   # (scene-affecting option: the next test commits its own scene)


def _check_random_scene(hip, seed):
    recipe = SceneRecipe(1000 + seed)
    orc = O.Oracle()
    recipe.build(orc)
    recipe.build(hip)
    o, d = H.random_rays(4000, seed=seed, origin_scale=4.0, toward=(0, 0, 0), spread=3.0)
    H.assert_hits_match(hip.closest(o, d), orc.closest(o, d), what=f"random scene {seed}")
    md = np.abs(np.random.default_rng(seed).normal(size=o.shape[0])) * 6.0
    assert np.array_equal(hip.blocked(o, d, md), orc.blocked(o, d, md)), f"random scene {seed}: lightIsBlocked differs"
    cam = ft.make_camera((1.0, 2.0, -9.0), (0, 0, 0), (0, 1, 0), H.deg(55.0))
    jit = ft.jitter_pattern(2)
    want, ost = orc.render(cam, 96, 64, 2, jit, seed=ft.DEFAULT_SEED)       # the soft-light streams are keyed by the seed: same one on both sides
    got, st = hip.render(cam, 96, 64, 2, jit, seed=ft.DEFAULT_SEED)
    # a negative base under a fractional shineyness is NaN in the reference too (Shading.fs:85: `**`): NaN must meet NaN
    assert np.array_equal(np.isnan(got), np.isnan(want)), f"random scene {seed}: NaN pixels differ"
    nan = np.isnan(want)
    worst = H.assert_frames_match(np.where(nan, 0.0, got), np.where(nan, 0.0, want), what=f"random scene {seed}")
    assert worst < 1e-6
    assert st["rays_reference_equivalent"] == ost["rays_traced"]


@pytest.mark.parametrize("seed", range(16))
def test_random_cameras_and_tiles_match_oracle(hip, seed):
    """The pixel-block classification and the wave-level culls bound rays by cones and pyramids built from the camera: random
    viewpoints (far, close, inside objects' bounding spheres), fields of view, aspect ratios, resolutions that are and are not
    multiples of 8, tiles, depth of field and corner sampling - every frame against the oracle, scenes WITHOUT a ground plane
    so that blocks really are dropped."""
    rng = np.random.default_rng(7000 + seed)
    recipe = SceneRecipe(3000 + seed, ground=False)
    hip.set_option("csg_mesh_capacity", 16)                       # a ray through the apex of a fan can cross it more than 8 times: the library refuses loudly (FT_ERR_OVERFLOW), never drops
    try:
        orc = O.Oracle()
        recipe.build(orc)
        recipe.build(hip)
        for k in range(3):
            dist = float(rng.choice([0.3, 1.5, 6.0, 25.0]))
            eye = rng.normal(size=3); eye = eye / np.linalg.norm(eye) * dist
            cam = ft.make_camera(tuple(eye), tuple(rng.normal(scale=0.5, size=3)), (0, 1, 0), H.deg(float(rng.uniform(15, 110))), float(rng.choice([1.0, 1.0, 1.6])))
            w, h = [(64, 64), (72, 40), (61, 37)][k]
            spp = int(rng.integers(1, 4))
            jit = ft.jitter_pattern(spp)
            tiles = None if k != 1 else [(0, 0, 32, 40), (32, 8, 40, 24)]
            want, ost = orc.render(cam, w, h, spp, jit, tiles=tiles, seed=ft.DEFAULT_SEED)
            got, st = hip.render(cam, w, h, spp, jit, tiles=tiles, seed=ft.DEFAULT_SEED)
            nan = np.isnan(want)
            assert np.array_equal(np.isnan(got), nan), f"seed {seed} view {k}: NaN pixels differ"
            assert H.assert_frames_match(np.where(nan, 0.0, got), np.where(nan, 0.0, want), what=f"seed {seed} view {k}") < 1e-6
            assert st["rays_reference_equivalent"] == ost["rays_traced"]
        cam.has_focus, cam.focal_length, cam.aperture_angular_size = 1, 6.0, 0.02          # depth of field: no classification, seeded streams
        want, _ = orc.render(cam, 48, 32, 2, ft.jitter_pattern(2), seed=ft.DEFAULT_SEED)
        got, _ = hip.render(cam, 48, 32, 2, ft.jitter_pattern(2), seed=ft.DEFAULT_SEED)
        nan = np.isnan(want)
        assert np.array_equal(np.isnan(got), nan) and H.assert_frames_match(np.where(nan, 0.0, got), np.where(nan, 0.0, want), what=f"seed {seed} depth of field") < 1e-6
        cam.has_focus = 0
        want, _ = orc.render(cam, 40, 24, 0, None, seed=ft.DEFAULT_SEED)                   # samples corner
        got, _ = hip.render(cam, 40, 24, 0, None, seed=ft.DEFAULT_SEED)
        nan = np.isnan(want)
        assert np.array_equal(np.isnan(got), nan) and H.assert_frames_match(np.where(nan, 0.0, got), np.where(nan, 0.0, want), what=f"seed {seed} corner sampling") < 1e-6
    finally:
        hip.set_option("csg_mesh_capacity", 32)
