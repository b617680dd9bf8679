#!/usr/bin/env python3
"""A/B of bounce 0: fused k_primary against k_closest + k_shade, over the config scenes at 1920x1080 (GPU box).
Prints kernel time per frame for both routes and whether the two frames are bit-identical."""
import os
import sys

import numpy as np

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import functracer_amd as ft

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CASES = [("bunny", 16), ("bunny", 4), ("hollow-sphere", 1), ("hollow-sphere", 16), ("night-house-det", 16), ("night-house", 16),
         ("bunny-bsp12", 16), ("sample-det", 16), ("moon", 16), ("repeat", 4)]
if len(sys.argv) > 1:
    CASES = [(a.split(":")[0], int(a.split(":")[1])) for a in sys.argv[1:]]
ctx = ft.Context(0)
for name, spp in CASES:
    p = ft.parse_scene_file(os.path.join(R, "scenes", name + ".scene"))
    p.lower(ctx)
    jit = ft.jitter_pattern(spp)
    frames, rows = {}, {}
    for fused in (0, 1):
        ctx.set_option("fused_primary", fused)
        best = None
        for _ in range(5):
            _, st = ctx.render(p.camera, 1920, 1080, spp, jit, fetch=False)
            if best is None or st["kernel_ms"] < best["kernel_ms"]:
                best, kt = st, ctx.kernel_times()
        f = np.zeros((1080, 1920, 3))
        ctx.fetch_frame(f)
        frames[fused], rows[fused] = f, (best, kt)
    same = np.array_equal(frames[0], frames[1])
    d = np.abs(frames[0] - frames[1]).max()
    s0, k0 = rows[0]; s1, k1 = rows[1]
    print(f"{name:18s} x{spp:<3d} split {s0['kernel_ms']:7.3f} ms (closest {k0['closest']['ms']:.3f} shade {k0['shade']['ms']:.3f} other {k0['other']['ms']:.3f}) | "
          f"fused {s1['kernel_ms']:7.3f} ms (primary {k1['primary']['ms']:.3f} closest {k1['closest']['ms']:.3f} shade {k1['shade']['ms']:.3f} other {k1['other']['ms']:.3f}) "
          f"identical={same} maxdiff={d:.2e} traced {s1['rays_traced']} stats_equal={all(s0[k] == s1[k] for k in ('rays_shadow', 'rays_reflect', 'hits_primary', 'hits_total', 'rays_reference_equivalent'))}", flush=True)
