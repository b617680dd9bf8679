#!/usr/bin/env python3
"""Render one scene a few times (for rocprofv3 --kernel-trace): python tools/frame_trace.py scene spp [frames] [option=value ...]"""
import os
import sys

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import functracer_amd as ft

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
name, spp = sys.argv[1], int(sys.argv[2])
frames = int(sys.argv[3]) if len(sys.argv) > 3 else 4
ctx = ft.Context(0)
for kv in sys.argv[4:]:
    k, v = kv.split("=")
    ctx.set_option(k, int(v))
p = ft.parse_scene_file(os.path.join(R, "scenes", name + ".scene"))
p.lower(ctx)
jit = ft.jitter_pattern(spp)
for _ in range(frames):
    _, st = ctx.render(p.camera, 1920, 1080, spp, jit, fetch=False)
print({k: st[k] for k in ("kernel_ms", "rays_traced", "rays_shadow", "rays_reflect", "hits_total", "n_launches")}, ctx.kernel_times())
