#!/usr/bin/env python3
"""VERDICT r2 item 6's rehearsal on a one-GPU box: config 5 (3840x2160x64) through ft_create([0, 0, 0]) - three sub-contexts on ONE device -
against three times a single context's third of the frame (the bands rank 0 of 3 gets).  python tools/multi_rehearsal.py [out.json]"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import functracer_amd as ft
from functracer_amd import tiling

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
W, H, SPP = 3840, 2160, 64
p = ft.parse_scene_file(os.path.join(R, "scenes", "bunny.scene"))
jit = ft.jitter_pattern(SPP)


def timed(ctx, n, **kw):
    for _ in range(2):
        ctx.render(p.camera, W, H, SPP, jit, fetch=False, **kw)
    t0 = time.perf_counter()
    for _ in range(n):
        ctx.render_enqueue(p.camera, W, H, SPP, jit, **kw)
    ctx.wait()
    return (time.perf_counter() - t0) / n * 1e3


single = ft.Context(0)
p.lower(single)
whole = timed(single, 8)
shares = [timed(single, 8, tiles=tiling.bands_for_rank(W, H, r, 3)) for r in range(3)]
single.close()
multi = ft.Context(device=[0, 0, 0])
p.lower(multi)
three = timed(multi, 8)
host = ft.PinnedArray((H, W, 3))
t0 = time.perf_counter()
for _ in range(4):
    multi.render(p.camera, W, H, SPP, jit, out=host.array)
blocking_pinned = (time.perf_counter() - t0) / 4 * 1e3
host.close()
multi.close()
out = {"workload": "bunny 3840x2160x64, one MI355X", "single_context_whole_frame_ms": round(whole, 3), "single_context_share_of_3_ms": [round(s, 3) for s in shares],
       "three_times_the_largest_share_ms": round(3 * max(shares), 3), "context_of_three_on_one_device_ms": round(three, 3),
       "ratio_to_3x_share": round(three / (3 * max(shares)), 3), "ratio_to_whole_frame": round(three / whole, 3),
       "blocking_into_pinned_host_frame_ms": round(blocking_pinned, 3), "frame_bytes": W * H * 24}
print(json.dumps(out, indent=1))
if len(sys.argv) > 1:
    json.dump(out, open(sys.argv[1], "w"), indent=1)
