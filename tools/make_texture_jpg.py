#!/usr/bin/env python3
"""Synthetic stand-in for the sky texture of the reference's Scenes/sample.scene:6 (`c:\\Temp\\env4.jpg`, a file on the author's disk):
an equirectangular sky - horizon glow, a sun, seeded clouds - written as a baseline 4:2:0 JPEG (needs Pillow; the committed file is what
the tests and scenes/sample.scene read, this script only documents how it was made).

    python tools/make_texture_jpg.py            # -> scenes/textures/env4_synth.jpg
"""
import os

import numpy as np
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
w, h = 384, 192
rng = np.random.default_rng(4)
lon = (np.arange(w) + 0.5) / w * 2 * np.pi
lat = (0.5 - (np.arange(h) + 0.5) / h) * np.pi
lon, lat = np.meshgrid(lon, lat)
p = np.stack([np.cos(lat) * np.cos(lon), np.sin(lat), np.cos(lat) * np.sin(lon)], -1)
up = np.clip(p[..., 1], -1, 1)
sky = np.stack([0.35 + 0.4 * (1 - up) ** 3, 0.55 + 0.3 * (1 - up) ** 3, 0.95 - 0.1 * (1 - up)], -1)
ground = np.stack([0.25 + 0.1 * np.sin(9 * lon), 0.22 + 0.08 * np.sin(7 * lon + 1), 0.18 + 0 * lon], -1)
img = np.where(up[..., None] > 0, sky, ground)
sun = np.array([0.5, 0.6, -0.62]); sun /= np.linalg.norm(sun)
img += np.exp((p @ sun - 1) * 180)[..., None] * np.array([1.0, 0.9, 0.6])
for _ in range(40):                                                   # clouds
    c = rng.normal(size=3); c[1] = abs(c[1]) * 0.5 + 0.1; c /= np.linalg.norm(c)
    img += (np.exp((p @ c - 1) * rng.uniform(60, 400)) * 0.35 * (up > 0))[..., None]
out = os.path.join(ROOT, "scenes", "textures", "env4_synth.jpg")
Image.fromarray((np.clip(img, 0, 1) * 255).astype(np.uint8)).save(out, quality=85, subsampling=2, optimize=False, restart_marker_rows=4)
print(out, os.path.getsize(out), "bytes")
