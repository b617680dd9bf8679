#!/usr/bin/env python3
"""Kernel-time table over the config scenes at 1920x1080: python tools/bench_scenes.py [tag]  (GPU box)."""
import json
import os
import sys

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import functracer_amd as ft

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CASES = [("bunny", 16), ("bunny", 4), ("hollow-sphere", 1), ("hollow-sphere", 16), ("night-house-det", 16), ("night-house", 16),
         ("bunny-bsp12", 16), ("bunny-full-bsp12", 16), ("sample-det", 16), ("moon", 16), ("repeat", 4)]
out = {}
ctx = ft.Context(0)
for kv in filter(None, os.environ.get("FT_OPTS", "").split(",")):      # FT_OPTS="coherent_waves=0,follow_below=2048": A/B runs
    k, v = kv.split("=")
    ctx.set_option(k, int(v))
only = set(filter(None, os.environ.get("FT_SCENES", "").split(",")))   # FT_SCENES="bunny,hollow-sphere": a subset
for name, spp in CASES:
    if only and name not in only:
        continue
    path = os.path.join(R, "scenes", name + ".scene")
    if name == "bunny-full-bsp12" and not os.path.exists(os.path.join(R, "scenes", "meshes", "bunny_synth_full.ply")):
        continue
    p = ft.parse_scene_file(path)
    p.lower(ctx)
    jit = ft.jitter_pattern(spp)
    best = None
    for _ in range(4):
        _, st = ctx.render(p.camera, 1920, 1080, spp, jit, fetch=False)
        if best is None or st["kernel_ms"] < best["kernel_ms"]:
            best = st
    kt = ctx.kernel_times()
    row = {"ms": round(best["kernel_ms"], 3), "mrays_s": round(best["rays_traced"] / best["kernel_ms"] / 1e3, 1), "rays": best["rays_traced"],
           "primary_ms": round(kt["primary"]["ms"], 3), "closest_ms": round(kt["closest"]["ms"], 3), "shade_ms": round(kt["shade"]["ms"], 3), "other_ms": round(kt["other"]["ms"], 3)}
    out[f"{name}x{spp}"] = row
    print(f"{name:18s} x{spp:<3d} {row['ms']:9.3f} ms {row['mrays_s']:10.1f} Mrays/s  primary {row['primary_ms']:.3f} closest {row['closest_ms']:.3f} shade+tail {row['shade_ms']:.3f} other {row['other_ms']:.3f}", flush=True)
if len(sys.argv) > 1:
    os.makedirs(os.path.join(R, "gpurun_out"), exist_ok=True)
    json.dump(out, open(os.path.join(R, "gpurun_out", f"scenes_{sys.argv[1]}.json"), "w"), indent=1)
