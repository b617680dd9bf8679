#!/usr/bin/env python3
"""Rays and kernel time per recursion limit (diagnostic): python tools/bounce_probe.py hollow-sphere"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import functracer_amd as ft  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "hollow-sphere"
p = ft.parse_scene_file(os.path.join(ROOT, "scenes", name + ".scene"))
ctx = ft.Context(0)
p.lower(ctx)
w, h = p.resolution
spp = p.samples
jit = ft.jitter_pattern(spp)
prev = 0
for depth in range(0, 9):
    for _ in range(2):
        _, st = ctx.render(p.camera, w, h, spp, jit, max_depth=depth, fetch=False)
    kt = ctx.kernel_times()
    print(f"depth {depth}: reflect rays {st['rays_reflect']:9d} (+{st['rays_reflect'] - prev:8d})  shadow {st['rays_shadow']:9d}  kernel_ms {st['kernel_ms']:.3f}  "
          f"closest {kt['closest']['ms']:.3f} shade {kt['shade']['ms']:.3f}")
    prev = st["rays_reflect"]
