import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import functracer_amd as ft
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
txt = open(os.path.join(R, "scenes/bunny.scene")).read()
only = os.environ.get("FT_EXP_ONLY")
for name, t in [("normal", txt), ("away", txt.replace("lookat (0,0,3)", "lookat (0,0,-30)")), ("empty", txt.split("(material")[0] + "\ndirectional dir (-3,-2,3) colour (1,1,1)\n")]:
    if only and name != only:
        continue
    p = ft.parse_scene(t, base_dir=os.path.join(R, "scenes"))
    ctx = ft.Context(0); p.lower(ctx)
    jit = ft.jitter_pattern(16)
    for _ in range(3):
        _, st = ctx.render(p.camera, 1920, 1080, 16, jit, fetch=False)
    print(name, round(st["kernel_ms"], 3), ctx.kernel_times(), st["hits_primary"])
    ctx.close()
