"""Diagnostic build (make item-counts): top-level items a wave's query evaluates, per kind of query.  GPU box:
FT_HIP_LIB=build/libfunctracer_hip_counts.so python tools/item_counts.py"""
import ctypes as C, os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import functracer_amd as ft
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
lib = ft.hip_lib()
ctx = ft.Context(0)
buf = (C.c_ulonglong * 16)()
for name, spp in (("hollow-sphere", 16), ("hollow-sphere", 1), ("night-house-det", 16), ("sample-det", 16), ("bunny", 16)):
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
            print(f"   {label:20s} wave-queries {q:9d}  items evaluated / query {items / q:6.2f}  offered by the mask {offered / q:6.2f}  live lanes {lanes / q:5.1f}")
