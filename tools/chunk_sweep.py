#!/usr/bin/env python3
"""Frame time against chunk_samples (GPU box): python tools/chunk_sweep.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import functracer_amd as ft  # noqa: E402

ctx = ft.Context(0)
for name, spp in [("bunny", 16), ("hollow-sphere", 16), ("hollow-sphere", 1), ("night-house-det", 16), ("night-house", 16), ("sample-det", 16), ("bunny-bsp12", 16), ("bunny-full-bsp12", 16), ("moon", 16)]:
    p = ft.parse_scene_file(os.path.join(ROOT, "scenes", name + ".scene"))
    p.lower(ctx)
    jit = ft.jitter_pattern(spp)
    row = []
    for mi in (8, 16, 24, 32, 64):
        ctx.set_option("chunk_samples", mi << 20)
        best = 1e9
        for _ in range(4):
            _, st = ctx.render(p.camera, 1920, 1080, spp, jit, fetch=False)
            best = min(best, st["kernel_ms"])
        row.append(f"{mi}Mi:{best:.3f}")
    print(f"{name} x{spp}: " + "  ".join(row), flush=True)
