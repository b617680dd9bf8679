#!/usr/bin/env python3
"""Per-kernel HBM traffic from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE), collected as
MI355X_MICROARCH.md §HBM prescribes: separate --pmc passes, values in KiB, and on gfx950 FETCH_SIZE
under-reports coalesced streaming reads (exactly 1/2 for 16 B/lane).  This path reads 8 B/lane, an
uncalibrated width; it was calibrated while k_blend still read every accumulator (exactly 24*spp B per pixel):
factor 1.979 and 1.9998 in two separate passes (profiles/r01_d_*, r01_j_pmc_traffic_bunny.json), WRITE_SIZE
exact (1.0004).  The read side is therefore doubled (--read-scale 2.0); k_blend's write (24 B per pixel) is
still checked in every run as `write_ratio`.

  python tools/pmc_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write --pixels 2073600 --spp 16
"""
import argparse
import csv
import glob
import json
import os
import re
from collections import defaultdict


def load(directory, counter):
    rows = defaultdict(list)
    for path in glob.glob(os.path.join(directory, "**", "*counter_collection.csv"), recursive=True):
        with open(path) as f:
            for r in csv.DictReader(f):
                if r["Counter_Name"] == counter:
                    m = re.search(r"\b(k_\w+)[<(]", r["Kernel_Name"])
                    name = m.group(1) if m else r["Kernel_Name"].split("(")[0]
                    rows[name].append(float(r["Counter_Value"]) * 1024.0)
    return rows


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("fetch_dir")
    ap.add_argument("write_dir")
    ap.add_argument("--pixels", type=int, required=True)
    ap.add_argument("--spp", type=int, required=True)
    ap.add_argument("--read-scale", type=float, default=2.0, help="FETCH_SIZE correction for this path's coalesced 8 B/lane reads")
    ap.add_argument("--active-pixels", type=int, default=0, help="pixels k_classify kept (ft_stats: (rays_primary - rays_primary_culled) / spp); they also cost 8 B of list each")
    ap.add_argument("--frames", type=int, default=7, help="frames rendered by the profiled command (steps + warmup)")
    args = ap.parse_args()
    fetch, write = load(args.fetch_dir, "FETCH_SIZE"), load(args.write_dir, "WRITE_SIZE")
    # Every pixel of a frame is written exactly once, 24 B, by k_classify (blocks that see nothing) or k_blend (the rest), and every
    # active pixel costs 8 B of list: the known byte count that checks WRITE_SIZE in every run.
    known_write = (24.0 * args.pixels + 8.0 * args.active_pixels) * args.frames
    out_write = sum(write.get("k_blend", [])) + sum(write.get("k_classify", []))
    read_scale = args.read_scale
    out = {"calibration": {"read_scale": read_scale, "read_scale_source": "k_blend full-accumulator passes r01_d / r01_j: 1.979, 1.9998",
                           "known_output_bytes_all_frames": known_write, "WRITE_SIZE_k_blend_plus_k_classify_all_frames": out_write,
                           "write_ratio": out_write / known_write}}
    for k in sorted(set(fetch) | set(write)):
        if not k.startswith("k_"):
            continue
        f, w = fetch.get(k, []), write.get(k, [])
        out[k] = {"launches": len(f), "fetch_raw_bytes_per_launch": sum(f) / max(1, len(f)), "fetch_corrected_bytes_per_launch": read_scale * sum(f) / max(1, len(f)),
                  "write_bytes_per_launch": sum(w) / max(1, len(w))}
        out[k]["traffic_bytes_per_launch"] = out[k]["fetch_corrected_bytes_per_launch"] + out[k]["write_bytes_per_launch"]
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
