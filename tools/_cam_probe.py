import os, sys
import numpy as np
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT)
import functracer_amd as ft
from oracle import ft_oracle_py as O
from tests import helpers as H
from tests.test_gpu_fuzz import SceneRecipe
seed = int(sys.argv[1])
rng = np.random.default_rng(7000 + seed)
recipe = SceneRecipe(3000 + seed, ground=False)
hip = ft.Context(0); hip.set_option("csg_mesh_capacity", 16)
orc = O.Oracle(); recipe.build(orc); recipe.build(hip)
print(hip.scene_info())
for k in range(3):
    dist = float(rng.choice([0.3, 1.5, 6.0, 25.0]))
    eye = rng.normal(size=3); eye = eye / np.linalg.norm(eye) * dist
    cam = ft.make_camera(tuple(eye), tuple(rng.normal(scale=0.5, size=3)), (0, 1, 0), H.deg(float(rng.uniform(15, 110))), float(rng.choice([1.0, 1.0, 1.6])))
    w, h = [(64, 64), (72, 40), (61, 37)][k]
    spp = int(rng.integers(1, 4)); jit = ft.jitter_pattern(spp)
    tiles = None if k != 1 else [(0, 0, 32, 40), (32, 8, 40, 24)]
    want, _ = orc.render(cam, w, h, spp, jit, tiles=tiles, seed=ft.DEFAULT_SEED)
    for name, opts in [("default", {}), ("no classify", {"classify_pixels": 0}), ("incoherent", {"coherent_waves": 0}), ("both off", {"classify_pixels": 0, "coherent_waves": 0}), ("depth 0", {"md": 0})]:
        hip.set_option("classify_pixels", 1); hip.set_option("coherent_waves", 1)
        md = 8
        for o, v in opts.items():
            if o == "md": md = v
            else: hip.set_option(o, v)
        w2 = want if md == 8 else orc.render(cam, w, h, spp, jit, tiles=tiles, seed=ft.DEFAULT_SEED, max_depth=md)[0]
        got, st = hip.render(cam, w, h, spp, jit, tiles=tiles, seed=ft.DEFAULT_SEED, max_depth=md)
        nan = np.isnan(w2)
        err = H.pixel_errors(np.where(nan, 0.0, got), np.where(nan, 0.0, w2)); bad = (~(err <= 1e-4)).any(-1); ys, xs = np.nonzero(bad)
        print(f"view {k} {name:12s} bad {int(bad.sum())}", [(int(x), int(y)) for x, y in zip(xs[:5], ys[:5])], "dist", dist, "spp", spp)
