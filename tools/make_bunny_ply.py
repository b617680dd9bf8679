#!/usr/bin/env python3
"""Synthetic stand-in for the Stanford bunny meshes the reference scene points at
(Scenes/bunny.scene:7 -> bun_zipper_res4.ply, absent from the reference repo and this image).

A closed genus-0 mesh: geodesic icosphere of frequency n (20 n^2 faces, 10 n^2 + 2 vertices) with
a seeded smooth radial displacement, scaled to the bunny's ~0.15-unit bounding box and centred
where the bunny sits, written in the ASCII PLY layout PlyParser.fs:20-61 accepts
(`x y z confidence intensity` per vertex, `3 a b c` per face).  Faces wind counter-clockwise seen
from outside, so Triangle.fs:64's winding normal points outward.

  res4 stand-in: n = 7  ->  492 vertices,   980 faces  (bun_zipper_res4: 453 / 948)
  full stand-in: n = 59 -> 34812 vertices, 69620 faces (bun_zipper:    35947 / 69451)
"""
import argparse

import numpy as np

SEED = 948


def icosphere(n):
    t = (1.0 + 5.0 ** 0.5) / 2.0
    v = np.array([[-1, t, 0], [1, t, 0], [-1, -t, 0], [1, -t, 0], [0, -1, t], [0, 1, t], [0, -1, -t], [0, 1, -t],
                  [t, 0, -1], [t, 0, 1], [-t, 0, -1], [-t, 0, 1]], dtype=np.float64)
    f = [(0, 11, 5), (0, 5, 1), (0, 1, 7), (0, 7, 10), (0, 10, 11), (1, 5, 9), (5, 11, 4), (11, 10, 2), (10, 7, 6), (7, 1, 8),
         (3, 9, 4), (3, 4, 2), (3, 2, 6), (3, 6, 8), (3, 8, 9), (4, 9, 5), (2, 4, 11), (6, 2, 10), (8, 6, 7), (9, 8, 1)]
    verts, index, faces = [], {}, []

    def vid(p):
        p = p / np.linalg.norm(p)
        key = tuple(np.round(p, 9))
        if key not in index:
            index[key] = len(verts)
            verts.append(p)
        return index[key]

    for a, b, c in f:
        A, B, C = v[a], v[b], v[c]
        grid = {}
        for i in range(n + 1):
            for j in range(n + 1 - i):
                grid[(i, j)] = vid(A + (B - A) * (i / n) + (C - A) * (j / n))
        for i in range(n):
            for j in range(n - i):
                faces.append((grid[(i, j)], grid[(i + 1, j)], grid[(i, j + 1)]))
                if i + j < n - 1:
                    faces.append((grid[(i + 1, j)], grid[(i + 1, j + 1)], grid[(i, j + 1)]))
    return np.array(verts), np.array(faces, dtype=np.int64)


def displace(dirs, seed=SEED):
    rng = np.random.default_rng(seed)
    r = np.ones(len(dirs))
    for _ in range(9):                       # smooth low-frequency lobes
        k = rng.normal(size=3) * rng.uniform(1.0, 3.5)
        r += rng.uniform(0.03, 0.11) * np.sin(dirs @ k + rng.uniform(0, 2 * np.pi))
    ears = np.clip(dirs @ np.array([0.25, 0.9, -0.35]) - 0.82, 0, None) + np.clip(dirs @ np.array([-0.3, 0.9, -0.3]) - 0.85, 0, None)
    return r + 2.2 * ears


def build(n, seed=SEED):
    dirs, faces = icosphere(n)
    pts = dirs * displace(dirs, seed)[:, None]
    lo, hi = pts.min(0), pts.max(0)
    pts = (pts - (lo + hi) / 2) * (0.155 / (hi - lo).max())       # longest side 0.155, like the bunny
    pts += np.array([-0.0168, 0.1101, -0.0015])                   # where the bunny's box is centred
    # outward winding check
    c = pts.mean(0)
    a, b, cc = pts[faces[:, 0]], pts[faces[:, 1]], pts[faces[:, 2]]
    nrm = np.cross(b - a, cc - a)
    flip = np.einsum("ij,ij->i", nrm, (a + b + cc) / 3 - c) < 0
    faces[flip] = faces[flip][:, [0, 2, 1]]
    return pts, faces


def write_ply(path, pts, faces):
    with open(path, "w", newline="\n") as f:
        f.write("ply\nformat ascii 1.0\ncomment synthetic bunny stand-in (tools/make_bunny_ply.py, seed %d)\n" % SEED)
        f.write("element vertex %d\nproperty float x\nproperty float y\nproperty float z\nproperty float confidence\nproperty float intensity\n" % len(pts))
        f.write("element face %d\nproperty list uchar int vertex_indices\nend_header\n" % len(faces))
        for p in pts:
            f.write("%.9g %.9g %.9g 1 0.5\n" % (p[0], p[1], p[2]))
        for a, b, c in faces:
            f.write("3 %d %d %d\n" % (a, b, c))


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=7)
    ap.add_argument("--out", required=True)
    args = ap.parse_args()
    pts, faces = build(args.n)
    write_ply(args.out, pts, faces)
    print(f"{args.out}: {len(pts)} vertices, {len(faces)} faces")
