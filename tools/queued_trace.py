"""Queue a few headline frames (for rocprofv3 --kernel-trace): python tools/queued_trace.py [frames]"""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import functracer_amd as ft
ctx = ft.Context(0); p = ft.parse_scene_file(os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "scenes/bunny.scene")); p.lower(ctx); jit = ft.jitter_pattern(16)
for kv in sys.argv[2:]:
    k, v = kv.split("="); ctx.set_option(k, int(v))
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 40): ctx.render_enqueue(p.camera, 1920, 1080, 16, jit)
ctx.wait()
