"""Unclassified frames: one chunk (a simple, pipelined frame) against two.  python tools/chunk_ab.py   (GPU box)"""
import os, sys, time
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); sys.path.insert(0, R)
import functracer_amd as ft
ctx = ft.Context(0)
for name, W, H, SPP in (("night-house-det", 1920, 1080, 16), ("night-house", 1920, 1080, 16), ("sample-det", 1920, 1080, 16), ("hollow-sphere", 1920, 1080, 16)):
    p = ft.parse_scene_file(os.path.join(R, "scenes", name + ".scene")); p.lower(ctx); jit = ft.jitter_pattern(SPP)
    for mi in (16, 34, 16, 34):
        ctx.set_option("chunk_samples", mi << 20)
        best = 1e9
        for rep in range(2):
            for _ in range(24): ctx.render_enqueue(p.camera, W, H, SPP, jit)
            ctx.wait()
            t0 = time.perf_counter()
            for _ in range(24): ctx.render_enqueue(p.camera, W, H, SPP, jit)
            ctx.wait(); best = min(best, (time.perf_counter() - t0) / 24 * 1e3)
        kt = ctx.kernel_times()
        print(f"{name:16s} chunk_samples {mi} Mi: {best:.4f} ms/frame, k_primary launches per frame {kt['primary']['launches'] / 24}", flush=True)
