#!/usr/bin/env python3
"""Writes tests/golden/frames.npz: small frames of the config scenes rendered by the CPU oracle
(oracle/ft_oracle.cpp) with the seeded jitter pattern.  The reference itself cannot be run here
(no F# toolchain), so these are restatement outputs, reviewed against the source."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import functracer_amd as ft  # noqa: E402
from oracle import ft_oracle_py as O  # noqa: E402

out = {}
for name, w, h, spp in [("hollow-sphere", 96, 54, 1), ("bunny", 96, 54, 2), ("night-house-det", 96, 54, 2), ("sample-det", 64, 64, 1)]:
    p = ft.parse_scene_file(os.path.join(ROOT, "scenes", name + ".scene"))
    orc = O.Oracle()
    p.lower(orc)
    jit = ft.jitter_pattern(spp)
    img, st = orc.render(p.camera, w, h, spp, jit)
    out[name] = img
    out[name + "_jitter"] = jit
    print(name, img.shape, st)
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "frames.npz"), **out)
