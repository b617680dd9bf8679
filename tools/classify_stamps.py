import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import functracer_amd as ft
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
ctx = ft.Context(0)
p = ft.parse_scene_file(os.path.join(R, "scenes", "bunny.scene")); p.lower(ctx)
jit = ft.jitter_pattern(16)
for _ in range(5): ctx.render(p.camera, 1920, 1080, 16, jit, fetch=False)
n = 127
buf = np.zeros((n, 8), dtype=np.uint64)
lib = ft.hip_lib(); lib.ft_debug_classify_stamps.argtypes = [C.c_void_p, C.c_void_p, C.c_int32]
assert lib.ft_debug_classify_stamps(ctx._ctx, buf.ctypes.data, n) == 0
t = buf[:, :6].astype(np.int64); t0 = t[:, 0].min()
rel = (t - t0) * 10  # ns
print("phase: start ticket loads classified lookback end  (ns after the first workgroup's start)")
for q in (0, 25, 50, 75, 100):
    print(q, np.percentile(rel, q, axis=0).astype(int))
d = np.diff(t, axis=1) * 10
print("phase durations ns (median, max):", np.median(d, axis=0).astype(int), d.max(axis=0))
print("kernel span ns:", (t[:, 5].max() - t0) * 10)
