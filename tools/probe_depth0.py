#!/usr/bin/env python3
"""Render one scene a few times at a fixed recursion limit (diagnostic, for rocprofv3 --pmc runs):
   python tools/probe_depth0.py <scene> <spp> <max_depth> [frames]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import functracer_amd as ft  # noqa: E402

name, spp, depth = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
frames = int(sys.argv[4]) if len(sys.argv) > 4 else 3
p = ft.parse_scene_file(os.path.join(ROOT, "scenes", name + ".scene"))
ctx = ft.Context(0)
p.lower(ctx)
jit = ft.jitter_pattern(spp)
for _ in range(frames):
    _, st = ctx.render(p.camera, 1920, 1080, spp, jit, max_depth=depth, fetch=False)
print(name, spp, depth, round(st["kernel_ms"], 3), ctx.kernel_times(), st["rays_traced"], st["hits_primary"])
