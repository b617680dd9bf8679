"""Diagnostic build (make item-counts): top-level items a wave's query evaluates, per kind of query.  GPU box:
FT_HIP_LIB=build/libfunctracer_hip_counts.so python tools/item_counts.py"""
import ctypes as C, os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import functracer_amd as ft
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
lib = ft.hip_lib()
ctx = ft.Context(0)
buf = (C.c_ulonglong * 48)()
for kv in filter(None, os.environ.get("FT_OPTS", "").split(",")):
    k_, v_ = kv.split("="); ctx.set_option(k_, int(v_))
for name, spp in (("bunny", 16), ("bunny-bsp12", 16), ("hollow-sphere", 16), ("hollow-sphere", 1), ("night-house-det", 16), ("sample-det", 16)):
    p = ft.parse_scene_file(os.path.join(R, "scenes", name + ".scene")); p.lower(ctx)
    jit = ft.jitter_pattern(spp)
    ctx.render(p.camera, 1920, 1080, spp, jit, fetch=False)
    lib.ft_debug_item_counts(buf, 1)
    _, st = ctx.render(p.camera, 1920, 1080, spp, jit, fetch=False)
    lib.ft_debug_item_counts(buf, 1)
    v = list(buf)
    print(f"{name} x{spp}: {st['kernel_ms']:.3f} ms (instrumented)")
    for k, label in enumerate(("closest coherent", "closest incoherent", "any coherent", "any incoherent")):
        q, items, lanes, offered = v[4 * k:4 * k + 4]
        if q:
            cull, loop = v[16 + 4 * k], v[16 + 4 * k + 1]
            print(f"   {label:20s} wave-queries {q:9d}  items evaluated / query {items / q:6.2f}  offered by the mask {offered / q:6.2f}  cycles / query: cull {cull / q:8.0f} item loop {loop / q:8.0f}")
    clk = v[16:]
    if clk[26]:
        nb = clk[26]
        print(f"   k_primary per batch ({nb} batches): total {clk[25] / nb:9.0f} cycles = ray generation {clk[22] / nb:8.0f} + closest trace {clk[23] / nb:8.0f} + shadow queries {clk[24] / nb:8.0f} + surface / shading / store / spawn {(clk[25] - clk[22] - clk[23] - clk[24]) / nb:8.0f}")
    if clk[21]:
        nb = clk[21]
        print(f"   k_bounce per batch ({nb} batches): total {clk[20] / nb:9.0f} cycles = closest trace {clk[16] / nb:8.0f} + surface {(clk[17] - clk[16]) / nb:8.0f} + shadow queries {clk[18] / nb:8.0f} + shading / store / spawn {clk[19] / nb:8.0f}")
