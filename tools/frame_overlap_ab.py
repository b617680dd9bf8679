"""A/B of the per-frame overheads on the headline (GPU box): queued frames, wall ms per frame for every setting of
classify_ahead / zero_fill_skip / resolve_blocks.  python tools/frame_overlap_ab.py"""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import functracer_amd as ft
ctx = ft.Context(0); p = ft.parse_scene_file(os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "scenes/bunny.scene")); p.lower(ctx); jit = ft.jitter_pattern(16)
for ahead, skip, rb, aside, after in ((0, 0, 0, 0, 0), (1, 1, 0, 0, 1), (1, 1, 0, 1, 0), (1, 1, 0, 1, 1), (1, 1, 2, 1, 0), (0, 1, 0, 1, 0), (1, 0, 0, 1, 0)):
    ctx.set_option("classify_ahead", ahead); ctx.set_option("zero_fill_skip", skip); ctx.set_option("resolve_blocks", rb)
    ctx.set_option("resolve_aside", aside); ctx.set_option("classify_after_trace", after)
    best = 1e9
    for rep in range(4):
        for _ in range(300): ctx.render_enqueue(p.camera, 1920, 1080, 16, jit)
        ctx.wait()
        t0 = time.perf_counter()
        for _ in range(200): ctx.render_enqueue(p.camera, 1920, 1080, 16, jit)
        ctx.wait(); best = min(best, (time.perf_counter() - t0) / 200 * 1e3)
    print(f"classify_ahead {ahead} zero_fill_skip {skip} resolve_blocks {rb} resolve_aside {aside} classify_after_trace {after}: {best:.4f} ms/frame", {k: round(v["ms"] / 200, 4) for k, v in ctx.kernel_times().items() if v["ms"]})
