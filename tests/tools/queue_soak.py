#!/usr/bin/env python3
"""Soak of the frame pipeline (GPU box): random sequences of queued frames - cameras, sizes, sample counts, tile lists, FP64 / RGBA8, with and
without a host buffer behind them, `mains` 1 .. 3 - against the same frames rendered one blocking call at a time.  Every delivered buffer and
the frame left on the device must equal its blocking twin bit for bit, and the last frame's ray counts must agree.
python tests/tools/queue_soak.py [sequences per scene] [seed] [big] [multi]   (multi: a context of two sub-contexts on the one device; big: frames of 960x544 .. 1920x1080 at 4 .. 16 samples: the kernels then run long enough to overlap)"""
import os
import sys

import numpy as np

ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
import functracer_amd as ft

n_seq = int(sys.argv[1]) if len(sys.argv) > 1 else 20
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 7)
BIG = "big" in sys.argv[3:]
MULTI = "multi" in sys.argv[3:]     # ft_create with the device listed twice: two sub-contexts take the bands of every frame, each with its own pipeline
ctx = ft.Context(device=[0, 0]) if MULTI else ft.Context(0)
bad = 0
total = 0
for name in ("bunny", "hollow-sphere", "sample-det", "night-house-det", "moon", "bunny-bsp12"):
    p = ft.parse_scene_file(os.path.join(ROOT, "scenes", name + ".scene"))
    p.lower(ctx)
    base = p.camera
    variants = []
    for k in range(6):
        w, h = ([(960, 544), (1280, 720), (1920, 1080)] if BIG else [(160, 96), (256, 192), (320, 200)])[k % 3]
        spp = ([4, 8, 16] if BIG else [1, 2, 4])[(k // 2) % 3]
        cam = ft.make_camera(tuple(np.array(base.o) + rng.normal(size=3) * 0.15), tuple(base.look_at), tuple(base.up), base.fov_y, w / h)
        tiles = None if k % 4 else [(8 * int(rng.integers(0, 4)), 8 * int(rng.integers(0, 3)), 64, 48)]
        jit = ft.jitter_pattern(spp, seed=int(rng.integers(1, 1000)))
        ctx.set_option("mains", 1)
        f64 = np.full((h, w, 3), -9.0); _, st = ctx.render(cam, w, h, spp, jit, tiles=tiles, out=f64)
        u8 = np.full((h, w, 4), 9, dtype=np.uint8); ctx.render_rgba8(cam, w, h, spp, jit, tiles=tiles, out=u8)
        variants.append((cam, w, h, spp, jit, tiles, f64, u8, st))
    for s in range(n_seq):
        mains = int(rng.integers(1, 4))
        ctx.set_option("mains", mains)
        plan = [(int(rng.integers(0, len(variants))), bool(rng.integers(0, 2)), bool(rng.integers(0, 3))) for _ in range(int(rng.integers(3, 13)))]
        bufs = []
        for v, rgba8, deliver in plan:
            cam, w, h, spp, jit, tiles, f64, u8, st = variants[v]
            pa = None
            if deliver:
                pa = ft.PinnedArray((h, w, 4), dtype=np.uint8) if rgba8 else ft.PinnedArray((h, w, 3))
                pa.array[...] = 9 if rgba8 else -9.0
            ctx.render_enqueue(cam, w, h, spp, jit, tiles=tiles, rgba8=rgba8, out=None if pa is None else pa.array)
            bufs.append(pa)
        stq = ctx.wait()
        ok = True
        for (v, rgba8, deliver), pa in zip(plan, bufs):
            if pa is not None:
                ok = ok and np.array_equal(pa.array, variants[v][7] if rgba8 else variants[v][6])
                pa.close()
        v, rgba8, _ = plan[-1]
        cam, w, h, spp, jit, tiles, f64, u8, st = variants[v]
        if tiles is None:                                           # (with tiles the rest of the device's frame buffer is whatever earlier frames left)
            last = ctx.fetch_frame_rgba8(np.zeros((h, w, 4), dtype=np.uint8)) if rgba8 else ctx.fetch_frame(np.zeros((h, w, 3)))
            ok = ok and np.array_equal(last, u8 if rgba8 else f64)
        ok = ok and all(stq[k] == st[k] for k in ("rays_traced", "rays_shadow", "rays_reflect", "rays_reference_equivalent"))
        total += 1
        if not ok:
            bad += 1
            print(f"MISMATCH scene {name} sequence {s} mains {mains} plan {plan}", flush=True)
    print(f"{name}: {n_seq} sequences done, {bad} mismatches so far", flush=True)
print(f"{total} sequences, {bad} mismatches")
sys.exit(1 if bad else 0)
