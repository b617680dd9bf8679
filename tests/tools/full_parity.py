#!/usr/bin/env python3
"""Full-size pixel parity of the HIP path against the CPU oracle on the BASELINE configurations
(the oracle uses every host core).  Writes a JSON summary: python tests/tools/full_parity.py > profiles/parity_full.json"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import functracer_amd as ft  # noqa: E402
from oracle import ft_oracle_py as O  # noqa: E402

CONFIGS = [  # BASELINE.json configs 1-4 (+ the as-written soft-light night-house); config 5 is the 8-GPU tiling of bunny
    ("sample-det", 256, 256, 1), ("hollow-sphere", 1920, 1080, 1), ("bunny", 1920, 1080, 4), ("bunny", 1920, 1080, 16),
    ("bunny-bsp12", 1920, 1080, 4), ("night-house-det", 1920, 1080, 16), ("night-house", 1920, 1080, 16), ("moon", 400, 400, 1)]
if len(sys.argv) > 1:
    CONFIGS = [c for c in CONFIGS if c[0] in sys.argv[1:]]
ctx = ft.Context(0)
out = []
for name, w, h, spp in CONFIGS:
    p = ft.parse_scene_file(os.path.join(ROOT, "scenes", name + ".scene"))
    p.lower(ctx)
    orc = O.Oracle()
    p.lower(orc)
    jit = ft.jitter_pattern(spp)
    got, st = ctx.render(p.camera, w, h, spp, jit, seed=ft.DEFAULT_SEED)
    t0 = time.time()
    want, ost = orc.render(p.camera, w, h, spp, jit, seed=ft.DEFAULT_SEED)
    err = np.abs(got - want) / np.maximum(np.abs(want), 1e-3)
    bad = int((err > 1e-4).any(axis=-1).sum())
    rec = {"scene": name, "res": [w, h], "spp": spp, "pixels": w * h, "max_rel_err": float(err.max()), "pixels_outside_1e-4": bad,
           "identical_pixels": int((got == want).all(axis=-1).sum()), "gpu_rays_traced": st["rays_traced"], "gpu_kernel_ms": round(st["kernel_ms"], 3),
           "oracle_rays_traced": ost["rays_traced"], "gpu_rays_reference_equivalent": st["rays_reference_equivalent"],
           "oracle_seconds": round(time.time() - t0, 1), "oracle_threads": ost["threads"]}
    out.append(rec)
    print(json.dumps(rec), file=sys.stderr, flush=True)
print(json.dumps(out, indent=1))
