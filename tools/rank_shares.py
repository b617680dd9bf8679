#!/usr/bin/env python3
"""Balance of the image-tile partition over N ranks, measured on ONE GPU: every rank's share of config 5 (bunny 3840x2160 x 64 spp)
rendered in turn.  python tools/rank_shares.py [N]
(Measured in round 3 beside a 2-D deal - every band cut into eight pieces, piece (i, j) to rank (i + j) % N: bands 1.042 max / mean at N = 8,
the 2-D deal no better (1.03 in rays, its rects cost launches): the bands stay.)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import functracer_amd as ft  # noqa: E402
from functracer_amd import tiling  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
p = ft.parse_scene_file(os.path.join(ROOT, "scenes", "bunny.scene"))
ctx = ft.Context(0)
p.lower(ctx)
w, h, spp = 3840, 2160, 64
jit = ft.jitter_pattern(spp)
for name, deal in (("8-row bands", tiling.bands_for_rank),):
    ms, rays = [], []
    for r in range(n):
        tiles = deal(w, h, r, n)
        best = None
        for _ in range(3):
            _, st = ctx.render(p.camera, w, h, spp, jit, tiles=tiles, fetch=False)
            if best is None or st["kernel_ms"] < best["kernel_ms"]:
                best = st
        ms.append(best["kernel_ms"]); rays.append(best["rays_traced"])
    print(f"{name:12s} N={n}: ms per rank {[round(x, 3) for x in ms]}  max/mean {max(ms) / (sum(ms) / n):.3f}  rays max/mean {max(rays) / (sum(rays) / n):.3f}  sum {sum(ms):.3f} ms")
