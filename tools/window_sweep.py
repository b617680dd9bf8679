"""Frame time against the chunk window (option chunk_samples) on one frame: python tools/window_sweep.py [scene w h spp]   (GPU box)"""
import os, sys, time
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); sys.path.insert(0, R)
import functracer_amd as ft
name, W, H, SPP = (sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else ("bunny", 3840, 2160, 64)
ctx = ft.Context(0)
p = ft.parse_scene_file(os.path.join(R, "scenes", name + ".scene")); p.lower(ctx)
jit = ft.jitter_pattern(SPP)
ctx.set_option("window_hint", 0)
for mi in (2, 4, 8, 12, 16, 24, 32, 64, 128):
    ctx.set_option("chunk_samples", mi << 20)
    for _ in range(3): ctx.render(p.camera, W, H, SPP, jit, fetch=False)
    t0 = time.perf_counter()
    for _ in range(12): ctx.render_enqueue(p.camera, W, H, SPP, jit)
    st = ctx.wait(); ms = (time.perf_counter() - t0) / 12 * 1e3
    kt = ctx.kernel_times()
    print(f"chunk_samples {mi:4d} Mi: {ms:8.4f} ms/frame", {k: round(v["ms"] / 12, 4) for k, v in kt.items() if v["ms"]}, "k_primary launches", kt["primary"]["launches"] / 12, flush=True)
