#!/usr/bin/env python3
"""Host surface-area-sweep BVH against the device-built linear BVH (ft_bvh.hip): commit time, tree height, frame time at 1920x1080 and
bit-identity of the frames (both trees are exact stand-ins for the reference's linear scan).  GPU box: python tools/bvh_compare.py"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import functracer_amd as ft

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CASES = [("bunny", 16, 0), ("bunny-full-bsp12", 16, 1)]      # (scene, spp, mesh_unclipped_bvh): 980 triangles; 69.6 K triangles
out = {}
ctx = ft.Context(0)
for name, spp, unclipped in CASES:
    if not os.path.exists(os.path.join(R, "scenes", "meshes", "bunny_synth_full.ply")) and "full" in name:
        continue
    p = ft.parse_scene_file(os.path.join(R, "scenes", name + ".scene"))
    jit = ft.jitter_pattern(spp)
    frames = {}
    for builder in (0, 1, 3):
        ctx.set_option("mesh_unclipped_bvh", unclipped)
        ctx.set_option("bvh_builder", builder)
        commits = []
        for _ in range(3):
            t0 = time.perf_counter()
            p.lower(ctx)                                        # scene graph + ft_scene_commit
            commits.append(dict(ctx.commit_times(), lower_and_commit_ms=(time.perf_counter() - t0) * 1e3))
        best = None
        for _ in range(5):
            _, st = ctx.render(p.camera, 1920, 1080, spp, jit, fetch=False)
            if best is None or st["kernel_ms"] < best["kernel_ms"]:
                best, kt = st, ctx.kernel_times()
        f = np.zeros((1080, 1920, 3))
        ctx.fetch_frame(f)
        frames[builder] = f
        c = min(commits, key=lambda q: q["lower_and_commit_ms"])
        row = {"builder": {0: "host surface-area sweep", 1: "device linear BVH", 3: "device binned surface-area tree"}[builder], "triangles": ctx.scene_info()["triangles"],
               "commit": {k: round(v, 3) if isinstance(v, float) else v for k, v in c.items()}, "frame_kernel_ms": round(best["kernel_ms"], 3),
               "k_primary_ms": round(kt["primary"]["ms"], 3), "rays_traced": best["rays_traced"]}
        out[f"{name} builder={builder}"] = row
        print(name, json.dumps(row), flush=True)
    same = bool(np.array_equal(frames[0], frames[1]) and np.array_equal(frames[0], frames[3]))
    out[f"{name} frames_identical"] = same
    print(name, "frames identical across builders:", same, "max abs diff", float(max(np.abs(frames[0] - frames[1]).max(), np.abs(frames[0] - frames[3]).max())), flush=True)
ctx.set_option("mesh_unclipped_bvh", 0)
os.makedirs(os.path.join(R, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(R, "gpurun_out", "bvh_compare.json"), "w"), indent=1)
