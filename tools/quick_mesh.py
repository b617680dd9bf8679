import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import functracer_amd as ft
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
ctx = ft.Context(0)
out = []
for name, spp, opt in (("bunny", 16, 0), ("bunny", 4, 0), ("bunny-bsp12", 16, 0), ("bunny-full-bsp12", 16, 1), ("moon", 16, 0)):
    ctx.set_option("mesh_unclipped_bvh", opt)
    p = ft.parse_scene_file(os.path.join(R, "scenes", name + ".scene")); p.lower(ctx)
    jit = ft.jitter_pattern(spp)
    best = min(ctx.render(p.camera, 1920, 1080, spp, jit, fetch=False)[1]["kernel_ms"] for _ in range(6))
    out.append(f"{name}{'+bvh' if opt else ''}x{spp} {best:.3f}")
print(os.environ.get("FT_HIP_LIB", "default").split("/")[-1], "  ".join(out))
