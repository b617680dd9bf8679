"""A rank's share of config 5 (3840x2160x64 over N ranks), queued frames: ms per frame for a few option settings.  python tools/rank_share_ab.py   (GPU box)"""
import os, sys, time
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); sys.path.insert(0, R)
import functracer_amd as ft
from functracer_amd import tiling
ctx = ft.Context(0)
p = ft.parse_scene_file(os.path.join(R, "scenes/bunny.scene")); p.lower(ctx)
W, H, SPP = 3840, 2160, 64
jit = ft.jitter_pattern(SPP)
for world in (8, 4, 2):
    tiles = tiling.bands_for_rank(W, H, 0, world)
    for label, opts in (("default", {}), ("one window", {"chunk_samples": 40 << 20}), ("one window, one main", {"chunk_samples": 40 << 20, "two_mains": 0}), ("default, one main", {"two_mains": 0})):
        for k, v in {"chunk_samples": 16 << 20, "two_mains": 1, **opts}.items(): ctx.set_option(k, v)
        best = 1e9
        for rep in range(3):
            n = 48
            for _ in range(n): ctx.render_enqueue(p.camera, W, H, SPP, jit, tiles=tiles)
            ctx.wait()
            t0 = time.perf_counter()
            for _ in range(n): ctx.render_enqueue(p.camera, W, H, SPP, jit, tiles=tiles)
            ctx.wait(); best = min(best, (time.perf_counter() - t0) / n * 1e3)
        kt = ctx.kernel_times()
        print(f"rank 0 of {world}  {label:22s}: {best:7.4f} ms/frame", {k: round(v["ms"] / n, 4) for k, v in kt.items() if v["ms"]}, "k_primary launches", kt["primary"]["launches"] / n, flush=True)
