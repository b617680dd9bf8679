"""Shared test helpers: scene construction through either builder, ray bundles, comparisons."""
import math
import os

import numpy as np

import functracer_amd as ft

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PRIMS = {"circle": ft.CIRCLE, "square": ft.SQUARE, "cube": ft.CUBE, "sphere": ft.SPHERE, "plane": ft.PLANE, "cone": ft.CONE,
         "solidCylinder": ft.SOLID_CYLINDER, "cylinder": ft.CYLINDER}

# Relative tolerance of the pixel parity contract (BASELINE.json north_star): 1e-4.
PIXEL_RTOL = 1e-4
# What the device path actually achieves against the oracle (FMA contraction + pre-composed matrices): ~1e-12.
TIGHT = 1e-9


def scene_path(name):
    return os.path.join(ROOT, "scenes", name + ".scene")


def single_prim(b, prim, xf=None, lights=True):
    """A scene holding one primitive (optionally under a transform) in builder `b`."""
    b.clear()
    node = b.primitive(PRIMS[prim])
    if xf:
        node = b.transform(xf, node)
    b.set_objects(b.group([node]))
    if lights:
        b.add_directional((0, -1, 0.5), (1, 1, 1))
    b.commit()


def random_rays(n, seed, origin_scale=3.0, toward=(0, 0, 0), spread=1.2):
    """Rays from random origins roughly toward `toward` (so that many of them hit things near it)."""
    rng = np.random.default_rng(seed)
    o = rng.normal(size=(n, 3)) * origin_scale
    target = np.asarray(toward, dtype=np.float64) + rng.normal(size=(n, 3)) * spread
    d = target - o
    d *= rng.uniform(0.2, 3.0, size=(n, 1))          # directions are NOT normalised in the reference (Image.fs:88-89)
    return o, d


def assert_hits_match(got, want, rtol=TIGHT, what=""):
    ghit, gt, gp, gn, gc = got
    whit, wt, wp, wn, wc = want
    mism = np.nonzero(ghit != whit)[0]
    assert mism.size == 0, f"{what}: hit/miss differs on {mism.size} rays, first {mism[:5]}"
    m = whit.astype(bool)
    if not m.any():
        return
    scale = 1.0 + np.abs(wt[m])
    assert np.max(np.abs(gt[m] - wt[m]) / scale) <= rtol, f"{what}: t differs by {np.max(np.abs(gt[m] - wt[m]) / scale)}"
    assert np.max(np.abs(gp[m] - wp[m]) / (1.0 + np.abs(wp[m]))) <= rtol, f"{what}: p differs"
    assert np.max(np.abs(gn[m] - wn[m])) <= 1e-7, f"{what}: n differs by {np.max(np.abs(gn[m] - wn[m]))}"
    assert np.max(np.abs(gc[m] - wc[m])) <= 1e-12, f"{what}: material colour differs"


def pixel_errors(got, want):
    """Per-channel relative error as the parity contract defines it: |a-b| / max(|b|, floor)."""
    floor = 1e-3                                      # channels darker than 1/1000 are compared absolutely at 1e-7
    return np.abs(got - want) / np.maximum(np.abs(want), floor)


def assert_frames_match(got, want, rtol=PIXEL_RTOL, what=""):
    err = pixel_errors(got, want)
    bad = int((err > rtol).any(axis=-1).sum())
    assert bad == 0, f"{what}: {bad} of {err.shape[0] * err.shape[1]} pixels outside rtol {rtol}; max rel err {err.max():.3e}"
    return float(err.max())


def deg(x):
    return x * (math.pi / 180.0)
