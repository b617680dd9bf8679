"""Headline frame period against the workgroup slots k_primary leaves free (option primary_reserve): python tools/reserve_sweep.py   (GPU box)"""
import os, sys, time
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); sys.path.insert(0, R)
import functracer_amd as ft
ctx = ft.Context(0)
for name, W, H, SPP in (("bunny", 1920, 1080, 16), ("bunny", 1920, 1080, 4), ("bunny-bsp12", 1920, 1080, 16), ("moon", 1920, 1080, 16)):
    p = ft.parse_scene_file(os.path.join(R, "scenes", name + ".scene")); p.lower(ctx); jit = ft.jitter_pattern(SPP)
    for reserve in (0, 16, 32, 64, 128, 256, 0, 64):
        ctx.set_option("primary_reserve", reserve)
        best = 1e9
        for rep in range(3):
            for _ in range(200): ctx.render_enqueue(p.camera, W, H, SPP, jit)
            ctx.wait()
            t0 = time.perf_counter()
            for _ in range(200): ctx.render_enqueue(p.camera, W, H, SPP, jit)
            ctx.wait(); best = min(best, (time.perf_counter() - t0) / 200 * 1e3)
        print(f"{name} x{SPP} primary_reserve {reserve:4d}: {best:.4f} ms/frame", {k: round(v["ms"] / 200, 4) for k, v in ctx.kernel_times().items() if v["ms"]}, flush=True)
