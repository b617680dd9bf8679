#!/usr/bin/env python3
"""One-off soak of the random-scene parity checks over many more seeds than the test suite carries:
   python tests/tools/fuzz_soak.py <first> <count>      (GPU box; prints the seeds that fail)"""
import os
import sys
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import functracer_amd as ft  # noqa: E402
from tests import test_gpu_fuzz as F  # noqa: E402

first, count = int(sys.argv[1]), int(sys.argv[2])
hip = ft.Context(0)
bad = []
for seed in range(first, first + count):
    for name, fn in (("scene", F.test_random_scene_matches_oracle), ("camera", F.test_random_cameras_and_tiles_match_oracle)):
        try:
            fn.__wrapped__(hip, seed) if hasattr(fn, "__wrapped__") else fn(hip, seed)
        except ft.FtError as e:
            if "overflow" in str(e).lower() or "LDS" in str(e):
                print(f"seed {seed} {name}: refused loudly ({e})", flush=True)
            else:
                bad.append((seed, name)); print(f"seed {seed} {name}: FtError {e}", flush=True)
        except AssertionError as e:
            bad.append((seed, name)); print(f"seed {seed} {name}: MISMATCH {str(e)[:200]}", flush=True)
        except Exception:
            bad.append((seed, name)); traceback.print_exc()
    if (seed - first) % 20 == 19:
        print(f"... {seed - first + 1} seeds done, {len(bad)} failures", flush=True)
print("failures:", bad)
