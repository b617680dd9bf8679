"""Kernel time of a 1080p frame against the number of samples a bounce-0 wavefront takes (option wave_samples)."""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import functracer_amd as ft
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
ctx = ft.Context(0)
scenes = [("bunny", 16), ("bunny", 64), ("hollow-sphere", 16), ("night-house-det", 16), ("night-house", 16), ("bunny-bsp12", 16), ("sample-det", 16), ("moon", 16), ("repeat", 4)]
for name, spp in scenes:
    p = ft.parse_scene_file(os.path.join(R, "scenes", name + ".scene")); p.lower(ctx)
    jit = ft.jitter_pattern(spp)
    row, ref = [], None
    for g in (0, 1, 4, 16):
        ctx.set_option("wave_samples", g)
        frame, _ = ctx.render(p.camera, 1920, 1080, spp, jit)
        if ref is None: ref = frame
        same = np.array_equal(ref, frame)
        best = min(ctx.render(p.camera, 1920, 1080, spp, jit, fetch=False)[1]["kernel_ms"] for _ in range(5))
        row.append(f"{g}:{best:.3f}{'' if same else '!DIFF'}")
    print(f"{name}x{spp}", "  ".join(row), flush=True)
