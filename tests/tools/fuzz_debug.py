#!/usr/bin/env python3
"""Diagnose a failing random scene of tests/test_gpu_fuzz.py: python tests/tools/fuzz_debug.py <seed> ...  (GPU box)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import functracer_amd as ft  # noqa: E402
from oracle import ft_oracle_py as O  # noqa: E402
from tests import helpers as H  # noqa: E402
from tests.test_gpu_fuzz import SceneRecipe  # noqa: E402

hip = ft.Context(0)
for seed in [int(a) for a in sys.argv[1:]]:
    r = SceneRecipe(1000 + seed)
    orc = O.Oracle()
    r.build(orc)
    r.build(hip)
    cam = ft.make_camera((1.0, 2.0, -9.0), (0, 0, 0), (0, 1, 0), H.deg(55.0))
    jit = ft.jitter_pattern(2)
    want, ost = orc.render(cam, 96, 64, 2, jit, seed=ft.DEFAULT_SEED)
    kinds = [c[0] for c in r.calls]
    print(f"seed {seed}: lights", [k for k in kinds if k.startswith("add_")], "nan px oracle", int(np.isnan(want).any(-1).sum()))
    for name, opts in [("default", {}), ("incoherent", {"coherent_waves": 0}), ("no classify", {"classify_pixels": 0}), ("all levels", {"level_hint": 0}), ("depth0", {"max_depth": 0})]:
        for k in ("classify_pixels", "level_hint", "coherent_waves"):
            hip.set_option(k, 1)
        md = 8
        for k, v in opts.items():
            if k == "max_depth":
                md = v
            else:
                hip.set_option(k, v)
        w2 = want
        if md != 8:
            w2, _ = orc.render(cam, 96, 64, 2, jit, max_depth=md, seed=ft.DEFAULT_SEED)
        got, st = hip.render(cam, 96, 64, 2, jit, max_depth=md, seed=ft.DEFAULT_SEED)
        both_nan = np.isnan(got) & np.isnan(w2)
        err = np.where(both_nan, 0.0, H.pixel_errors(got, w2))
        bad = (~(err <= 1e-4)).any(-1)
        ys, xs = np.nonzero(bad)
        print(f"   {name:12s} bad px {int(bad.sum()):4d}  nan-mismatch {int((np.isnan(got) != np.isnan(w2)).any(-1).sum())}", [(int(x), int(y)) for x, y in zip(xs[:6], ys[:6])])
        if name == "default" and bad.any():
            x, y = int(xs[0]), int(ys[0])
            print("      first:", (x, y), "gpu", got[y, x], "oracle", w2[y, x])
    hip.set_option("classify_pixels", 1); hip.set_option("level_hint", 1); hip.set_option("coherent_waves", 1)
