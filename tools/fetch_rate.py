#!/usr/bin/env python3
"""PCIe-inclusive frame rate of the headline workload: render + ft_fetch_frame into one long-lived host array, pageable, as a
host program would hold it.  Prints one JSON line; run on the GPU box.  (Page-locking the array was measured in round 1 and gave
nothing: profiles/r01_z_fetch_rate_bunny.json holds both legs.)"""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import functracer_amd as ft

def main():
    frames = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    wl = ft.parse_scene_file(os.path.join(root, "scenes", "bunny.scene"))
    ctx = ft.Context(0)
    wl.lower(ctx)
    res_h, res_v, spp = 1920, 1080, 16
    jit = ft.jitter_pattern(spp)
    cam = wl.camera
    frame = np.zeros((res_v, res_h, 3))
    out = {"workload": "bunny %dx%dx%d" % (res_h, res_v, spp), "frames": frames}
    for pin in (0,):
        for _ in range(3):
            _, st = ctx.render(cam, res_h, res_v, spp, jit, fetch=False); ctx.fetch_frame(frame)
        t0 = time.perf_counter(); rays = 0; fetch = 0.0
        for _ in range(frames):
            _, st = ctx.render(cam, res_h, res_v, spp, jit, fetch=False)
            t1 = time.perf_counter(); ctx.fetch_frame(frame); fetch += time.perf_counter() - t1
            rays += st["rays_reference_equivalent"]
        wall = time.perf_counter() - t0
        out["pinned" if pin else "pageable"] = {"ms_per_frame": round(wall / frames * 1e3, 3), "fetch_ms": round(fetch / frames * 1e3, 3),
                                                 "fetch_GBps": round(frame.nbytes / (fetch / frames) / 1e9, 2), "Mrays_per_s": round(rays / wall / 1e6, 1)}
    print(json.dumps(out))

if __name__ == "__main__":
    main()
