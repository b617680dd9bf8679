#!/usr/bin/env python3
"""Synthetic stand-in for the image textures the reference scenes fetch (Scenes/moon.scene:5 is an HTTP URL,
Scenes/sample.scene:6 a file on the author's disk): an equirectangular "moon" of seeded craters, written as an
8-bit RGB PNG with per-row adaptive filters so the loader's five filter paths are all exercised by a scene file.

    python tools/make_texture_png.py            # -> scenes/textures/moon_synth_256x128.png
"""
import argparse
import os
import struct
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def moon(width, height, seed):
    rng = np.random.default_rng(seed)
    lon = (np.arange(width) + 0.5) / width * 2 * np.pi
    lat = ((np.arange(height) + 0.5) / height - 0.5) * np.pi
    lon, lat = np.meshgrid(lon, lat)
    p = np.stack([np.cos(lat) * np.cos(lon), np.sin(lat), np.cos(lat) * np.sin(lon)], -1)
    shade = 0.62 + 0.10 * np.sin(3 * p[..., 0] + 1.3) * np.cos(2 * p[..., 1]) + 0.06 * np.sin(5 * p[..., 2])
    for _ in range(90):
        c = rng.normal(size=3)
        c /= np.linalg.norm(c)
        r = rng.uniform(0.03, 0.22)
        d = np.arccos(np.clip(p @ c, -1, 1)) / r
        shade = np.where(d < 0.8, shade * 0.78, np.where(d < 1.0, shade * 1.12, shade))
    g = np.clip(shade, 0, 1)
    g = np.round(g * 40) / 40                                          # few levels: compresses well
    rgb = np.stack([g, g * 0.97, g * 0.90], -1)
    rgb[:, : width // 32] *= (1.0, 0.55, 0.55)                        # a red seam at u ~ 0 makes orientation errors visible
    return (rgb * 255).astype(np.uint8)


def _filtered(rows):
    """Filter row y with type y % 5 (None, Sub, Up, Average, Paeth) - PNG spec section 9."""
    h, stride = rows.shape
    bpp = 3
    out = bytearray()
    prev = np.zeros(stride, dtype=np.int32)
    for y in range(h):
        cur = rows[y].astype(np.int32)
        a = np.concatenate([np.zeros(bpp, np.int32), cur[:-bpp]])
        c = np.concatenate([np.zeros(bpp, np.int32), prev[:-bpp]])
        b = prev
        f = y % 5
        if f == 0:
            pred = 0
        elif f == 1:
            pred = a
        elif f == 2:
            pred = b
        elif f == 3:
            pred = (a + b) >> 1
        else:
            p = a + b - c
            pa, pb, pc = abs(p - a), abs(p - b), abs(p - c)
            pred = np.where((pa <= pb) & (pa <= pc), a, np.where(pb <= pc, b, c))
        out.append(f)
        out += ((cur - pred) & 0xFF).astype(np.uint8).tobytes()
        prev = cur
    return bytes(out)


def write_png(path, rgb):
    h, w, _ = rgb.shape

    def chunk(kind, body):
        return struct.pack(">I", len(body)) + kind + body + struct.pack(">I", zlib.crc32(kind + body) & 0xFFFFFFFF)

    data = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 2, 0, 0, 0))
    data += chunk(b"IDAT", zlib.compress(_filtered(rgb.reshape(h, w * 3)), 9)) + chunk(b"IEND", b"")
    with open(path, "wb") as f:
        f.write(data)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--width", type=int, default=256)
    ap.add_argument("--height", type=int, default=128)
    ap.add_argument("--seed", type=int, default=1969)
    ap.add_argument("--out", default=os.path.join(ROOT, "scenes", "textures", "moon_synth_256x128.png"))
    a = ap.parse_args()
    write_png(a.out, moon(a.width, a.height, a.seed))
    print(a.out, os.path.getsize(a.out), "bytes")
