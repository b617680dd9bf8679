import os, sys, time
import numpy as np
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); sys.path.insert(0, R)
import functracer_amd as ft
ctx = ft.Context(0)
for name, spp in (("bunny", 16), ("sample-det", 16), ("night-house-det", 16)):
    p = ft.parse_scene_file(os.path.join(R, "scenes", name + ".scene")); p.lower(ctx); jit = ft.jitter_pattern(spp)
    with ft.PinnedArray((1080, 1920, 3)) as buf:
        def t(fn, n=10):
            for _ in range(3): fn()
            t0 = time.perf_counter()
            for _ in range(n): fn()
            return round((time.perf_counter() - t0) / n * 1e3, 3)
        a = t(lambda: ctx.render(p.camera, 1920, 1080, spp, jit, out=buf))
        b = t(lambda: ctx.render(p.camera, 1920, 1080, spp, jit, fetch=False))
        c = t(lambda: (ctx.render(p.camera, 1920, 1080, spp, jit, fetch=False), ctx.fetch_frame(buf)))
        _, st = ctx.render(p.camera, 1920, 1080, spp, jit, out=buf)
        print(name, "render(out)", a, "render(no fetch)", b, "render + fetch_frame", c, "wall_ms", round(st["wall_ms"], 3), "kernel_ms", round(st["kernel_ms"], 3), flush=True)
