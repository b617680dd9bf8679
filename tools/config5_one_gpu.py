#!/usr/bin/env python3
"""BASELINE config 5 (bunny 3840x2160 x 64 spp) on ONE GPU, and the share one of 8 ranks renders (GPU box)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import functracer_amd as ft  # noqa: E402
from functracer_amd import tiling  # noqa: E402

p = ft.parse_scene_file(os.path.join(ROOT, "scenes", "bunny.scene"))
ctx = ft.Context(0)
p.lower(ctx)
jit = ft.jitter_pattern(64)
for _ in range(2):
    _, st = ctx.render(p.camera, 3840, 2160, 64, jit, fetch=False)
print("whole frame on one GPU:", round(st["kernel_ms"], 2), "ms", round(st["rays_traced"] / st["kernel_ms"] / 1e3), "Mrays/s", st["n_chunks"], "chunks", st["rays_traced"], "rays,",
      round(100.0 * st["rays_primary_culled"] / st["rays_primary"], 1), "% of the primary rays resolved per block")
bands = tiling.bands_for_rank(3840, 2160, 0, 8)
for _ in range(2):
    _, st = ctx.render(p.camera, 3840, 2160, 64, jit, tiles=bands, fetch=False)
print("rank 0 of 8:", round(st["kernel_ms"], 2), "ms", round(st["rays_traced"] / st["kernel_ms"] / 1e3), "Mrays/s", st["n_chunks"], "chunks")
# the bench's weak-scaling frames: 1920x1080 at 16 x N spp, the share of rank 0 of N (per-rank work as at N = 1)
for n in (1, 2, 4, 8):
    jit = ft.jitter_pattern(16 * n)
    bands = tiling.bands_for_rank(1920, 1080, 0, n) if n > 1 else None
    best = None
    for _ in range(4):
        _, st = ctx.render(p.camera, 1920, 1080, 16 * n, jit, tiles=bands, fetch=False)
        if best is None or st["kernel_ms"] < best["kernel_ms"]: best = st
    print(f"bench share, rank 0 of {n}: {best['kernel_ms']:.3f} ms, {best['rays_traced'] / best['kernel_ms'] / 1e3:.0f} Mrays/s, {best['n_chunks']} chunk(s), wall {best['wall_ms']:.3f} ms")
