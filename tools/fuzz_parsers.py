#!/usr/bin/env python3
"""Mutation fuzz of the host-side parsers (scene grammar, PLY, PNG / PNM loader): every input must come back as a value or as a
Python exception - never a crash.  Meant to run under tools/sanitize_host.sh's libraries:
   FT_HIP_LIB=build/asan/libfunctracer_hip.so FT_HOST_LIB=build/asan/libfunctracer_host.so LD_PRELOAD="libasan.so libubsan.so" \
   python tools/fuzz_parsers.py [seconds]"""
import os, random, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import functracer_amd as ft  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 30.0
rng = random.Random(1234)
scenes = [open(os.path.join(ROOT, "scenes", n)).read() for n in sorted(os.listdir(os.path.join(ROOT, "scenes"))) if n.endswith(".scene") and "full" not in n]
ply = open(os.path.join(ROOT, "scenes", "meshes", "bunny_synth_res4.ply")).read()
png = open(os.path.join(ROOT, "scenes", "textures", "moon_synth_256x128.png"), "rb").read()
pnm = b"P6\n4 2\n255\n" + bytes(range(24))
tokens = ["(", ")", ",", ".", "{", "}", "0", "-1", "1e308", "1e-320", "nan", "repeat", "bspMesh", "subtract", "material", "scale", "\"x\"", "image", "grid", "\n", "#"]


def mutate_text(s):
    s = list(s)
    for _ in range(rng.randint(1, 6)):
        k = rng.randrange(5)
        i = rng.randrange(len(s) + 1)
        if k == 0 and s: del s[i % len(s):(i % len(s)) + rng.randint(1, 40)]
        elif k == 1: s[i:i] = list(rng.choice(tokens))
        elif k == 2 and s: s[i % len(s)] = chr(rng.randrange(1, 127))
        elif k == 3 and s: j = rng.randrange(len(s)); s[i:i] = s[j:j + rng.randint(1, 60)]
        else: s = s[:i]
    return "".join(s)


def mutate_bytes(b):
    b = bytearray(b)
    for _ in range(rng.randint(1, 8)):
        k = rng.randrange(4)
        i = rng.randrange(len(b) + 1)
        if k == 0 and b: b[i % len(b)] = rng.randrange(256)
        elif k == 1 and b: del b[i % len(b):(i % len(b)) + rng.randint(1, 64)]
        elif k == 2: b[i:i] = bytes(rng.randrange(256) for _ in range(rng.randint(1, 16)))
        else: b = b[:i]
    return bytes(b)


counts = {"scene": 0, "ply": 0, "image": 0, "lowered": 0}
ctx = ft.Context(host_only=True)
tmp = tempfile.mkdtemp(dir=os.path.join(ROOT, "build"))
t_end = time.time() + budget
while time.time() < t_end:
    try:
        p = ft.parse_scene(mutate_text(rng.choice(scenes)), base_dir=os.path.join(ROOT, "scenes"))
        counts["scene"] += 1
        if rng.random() < 0.3:
            try: p.lower(ctx); counts["lowered"] += 1
            except (ft.FtError, ValueError): pass
    except (ValueError, ft.FtError): counts["scene"] += 1
    try: ft.parse_ply(mutate_text(ply))
    except (ValueError, ft.FtError): pass
    counts["ply"] += 1
    path = os.path.join(tmp, "f.png" if rng.random() < 0.7 else "f.ppm")
    open(path, "wb").write(mutate_bytes(png if path.endswith("png") else pnm))
    try: ft.load_image(path)
    except (ValueError, ft.FtError, OSError): pass
    counts["image"] += 1
print("survived:", counts)
