"""ft_render_enqueue_into, FP64 frames: ms per frame against the number of page-locked host buffers in the ring and the main streams.  python tools/stream_ring_ab.py"""
import os, sys, time
import numpy as np
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); sys.path.insert(0, R)
import functracer_amd as ft
ctx = ft.Context(0)
p = ft.parse_scene_file(os.path.join(R, "scenes/bunny.scene")); p.lower(ctx); jit = ft.jitter_pattern(16)
for mains in (2, 1):
    ctx.set_option("mains", mains)
    for rgba8 in (False, True):
        for nbuf in (2, 5, 8):
            shape, dt = ((nbuf, 1080, 1920, 4), np.uint8) if rgba8 else ((nbuf, 1080, 1920, 3), np.float64)
            with ft.PinnedArray(shape, dtype=dt) as ring:
                best = 1e9
                for rep in range(3):
                    for k in range(8): ctx.render_enqueue(p.camera, 1920, 1080, 16, jit, rgba8=rgba8, out=ring[k % nbuf])
                    ctx.wait()
                    t0 = time.perf_counter()
                    for k in range(40): ctx.render_enqueue(p.camera, 1920, 1080, 16, jit, rgba8=rgba8, out=ring[k % nbuf])
                    ctx.wait(); best = min(best, (time.perf_counter() - t0) / 40 * 1e3)
            print(f"mains {mains} {'rgba8' if rgba8 else 'f64  '} ring of {nbuf}: {best:.4f} ms/frame", flush=True)
