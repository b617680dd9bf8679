"""What one ft_render_enqueue costs the HOST (the first four calls from an idle context never wait for a slot): python tools/enqueue_host_cost.py"""
import os, sys, time
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); sys.path.insert(0, R)
import functracer_amd as ft
ctx = ft.Context(0)
for name, spp in (("bunny", 16), ("hollow-sphere", 1), ("hollow-sphere", 16), ("sample-det", 16)):
    p = ft.parse_scene_file(os.path.join(R, "scenes", name + ".scene")); p.lower(ctx); jit = ft.jitter_pattern(spp)
    for _ in range(8): ctx.render_enqueue(p.camera, 1920, 1080, spp, jit)
    st = ctx.wait()
    best = 1e9
    for rep in range(5):
        t0 = time.perf_counter()
        for _ in range(4): ctx.render_enqueue(p.camera, 1920, 1080, spp, jit)
        t1 = time.perf_counter()
        ctx.wait()
        best = min(best, (t1 - t0) / 4 * 1e6)
    print(f"{name} x{spp}: {best:.1f} us of host time per enqueue, {st['n_launches'] if 'n_launches' in st else '?'} launches per frame", flush=True)
