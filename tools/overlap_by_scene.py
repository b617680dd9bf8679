"""Is the frame pipeline (classify_ahead / resolve_aside) worth it for a frame of this size?  Queued frames, wall ms per frame with each part
on / off, per scene and size.  python tools/overlap_by_scene.py [out.json]   (GPU box)"""
import json, os, sys, time
R = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
import functracer_amd as ft
CASES = [("bunny", 1920, 1080, 16), ("bunny", 1920, 1080, 4), ("bunny", 3840, 2160, 64), ("bunny", 3840, 2160, 16), ("hollow-sphere", 1920, 1080, 1), ("hollow-sphere", 1920, 1080, 16),
         ("night-house-det", 1920, 1080, 16), ("bunny-bsp12", 1920, 1080, 16), ("moon", 1920, 1080, 16), ("repeat", 1920, 1080, 4), ("sample-det", 1920, 1080, 16)]
SETTINGS = [(1, 1)] if os.environ.get("FT_ONLY_DEFAULT") else [(0, 0), (1, 0), (0, 1), (1, 1)]
ctx = ft.Context(0)
for kv in filter(None, os.environ.get("FT_OPTS", "").split(",")):
    k, v = kv.split("="); ctx.set_option(k, int(v))
out = {}
for name, w, h, spp in CASES:
    p = ft.parse_scene_file(os.path.join(R, "scenes", name + ".scene")); p.lower(ctx); jit = ft.jitter_pattern(spp)
    row = {}
    for rep in range(2):
        for ahead, aside in SETTINGS:
            ctx.set_option("classify_ahead", ahead); ctx.set_option("resolve_aside", aside)
            n = 0; t_end = time.perf_counter() + 0.15
            while time.perf_counter() < t_end:
                for _ in range(8): ctx.render_enqueue(p.camera, w, h, spp, jit)
                ctx.wait()
            n = max(8, int(0.25 / max(1e-5, (time.perf_counter() - t_end + 0.15) / 1e9 + 1e-3)))
            t0 = time.perf_counter(); k = 0
            while time.perf_counter() - t0 < 0.25:
                for _ in range(16): ctx.render_enqueue(p.camera, w, h, spp, jit)
                k += 16
            ctx.wait()
            ms = (time.perf_counter() - t0) / k * 1e3
            key = f"ahead{ahead}_aside{aside}"
            row[key] = min(row.get(key, 1e9), round(ms, 4))
    out[f"{name}_{w}x{h}x{spp}"] = row
    print(f"{name:16s} {w}x{h}x{spp:<3d}", row, flush=True)
if len(sys.argv) > 1:
    json.dump(out, open(sys.argv[1], "w"), indent=1)
