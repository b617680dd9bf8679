import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import functracer_amd as ft
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
ctx = ft.Context(0)
out = []
for name, spp in (("hollow-sphere", 1), ("hollow-sphere", 16), ("night-house-det", 16), ("sample-det", 16)):
    p = ft.parse_scene_file(os.path.join(R, "scenes", name + ".scene")); p.lower(ctx)
    jit = ft.jitter_pattern(spp)
    best = min(ctx.render(p.camera, 1920, 1080, spp, jit, fetch=False)[1]["kernel_ms"] for _ in range(5))
    out.append(f"{name}x{spp} {best:.3f}")
print(os.environ.get("FT_HIP_LIB", "default").split("/")[-1], "  ".join(out))
