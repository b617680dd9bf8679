#!/usr/bin/env python3
"""Frame time against the recursion limit: what each further bounce costs.  python tools/depth_sweep.py [scene spp]  (GPU box)"""
import os
import sys

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import functracer_amd as ft

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
name = sys.argv[1] if len(sys.argv) > 1 else "hollow-sphere"
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 1
ctx = ft.Context(0)
p = ft.parse_scene_file(os.path.join(R, "scenes", name + ".scene"))
p.lower(ctx)
jit = ft.jitter_pattern(spp)
for depth in range(0, 9):
    best = None
    for _ in range(4):
        _, st = ctx.render(p.camera, 1920, 1080, spp, jit, max_depth=depth, fetch=False)
        if best is None or st["kernel_ms"] < best["kernel_ms"]:
            best, kt = st, ctx.kernel_times()
    print(f"{name} x{spp} depth {depth}: {best['kernel_ms']:.3f} ms  primary {kt['primary']['ms']:.3f} bounces {kt['shade']['ms']:.3f}  reflect rays {best['rays_reflect']}", flush=True)
