#!/usr/bin/env python3
"""Frame time against the tail hand-over threshold (GPU box): python tools/tail_sweep.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import functracer_amd as ft  # noqa: E402

ctx = ft.Context(0)
for name, spp in [("hollow-sphere", 1), ("hollow-sphere", 16), ("night-house-det", 16), ("sample-det", 16), ("repeat", 4)]:
    p = ft.parse_scene_file(os.path.join(ROOT, "scenes", name + ".scene"))
    p.lower(ctx)
    jit = ft.jitter_pattern(spp)
    row = []
    for thr in (0, 4096, 16384, 65536, 262144, 1048576, 1 << 30):
        ctx.set_option("tail_rays", thr)
        best = 1e9
        for _ in range(4):
            _, st = ctx.render(p.camera, 1920, 1080, spp, jit, fetch=False)
            best = min(best, st["kernel_ms"])
        row.append(f"{thr}:{best:.3f}({st['rays_tail']})")
    print(f"{name} x{spp}: " + "  ".join(row), flush=True)
