"""The window hint (option window_hint / window_cap) on a rank's share of config 5 and on the whole frame: frames identical, ms per queued frame.
python tools/window_hint_ab.py   (GPU box)"""
import os, sys, time
import numpy as np
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); sys.path.insert(0, R)
import functracer_amd as ft
from functracer_amd import tiling
ctx = ft.Context(0)
p = ft.parse_scene_file(os.path.join(R, "scenes/bunny.scene")); p.lower(ctx)
W, H, SPP = 3840, 2160, 64
jit = ft.jitter_pattern(SPP)
for label, tiles in (("rank 0 of 8", tiling.bands_for_rank(W, H, 0, 8)), ("rank 3 of 4", tiling.bands_for_rank(W, H, 3, 4)), ("rank 1 of 2", tiling.bands_for_rank(W, H, 1, 2)), ("whole frame", None)):
    ctx.set_option("window_hint", 0)
    a, sa = ctx.render(p.camera, W, H, SPP, jit, tiles=tiles)
    ctx.set_option("window_hint", 1)
    b, sb = ctx.render(p.camera, W, H, SPP, jit, tiles=tiles)
    c, sc = ctx.render(p.camera, W, H, SPP, jit, tiles=tiles)
    assert np.array_equal(a, b) and np.array_equal(a, c) and sa["rays_traced"] == sc["rays_traced"]
    for hint in (0, 1, 0, 1):
        ctx.set_option("window_hint", hint)
        n = 24
        for _ in range(n): ctx.render_enqueue(p.camera, W, H, SPP, jit, tiles=tiles)
        ctx.wait()
        t0 = time.perf_counter()
        for _ in range(n): ctx.render_enqueue(p.camera, W, H, SPP, jit, tiles=tiles)
        ctx.wait(); ms = (time.perf_counter() - t0) / n * 1e3
        kt = ctx.kernel_times()
        print(f"{label:12s} window_hint {hint}: {ms:7.4f} ms/frame", {k: round(v["ms"] / n, 4) for k, v in kt.items() if v["ms"]}, "k_primary launches", kt["primary"]["launches"] / n, flush=True)
