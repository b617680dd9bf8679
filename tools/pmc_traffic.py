#!/usr/bin/env python3
"""Per-kernel HBM traffic from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE), collected as
MI355X_MICROARCH.md §HBM prescribes: separate --pmc passes, values in KiB, and on gfx950 FETCH_SIZE
under-reports coalesced streaming reads (exactly 1/2 for 16 B/lane).  This path reads 8 B/lane, an
uncalibrated width, so every run calibrates itself on the one kernel whose bytes are known exactly: k_resolve
reads 24 B per active sample + 4 B of pixel id per listed pixel + 4 B per block of the maps, and writes 24 B per pixel
of the frame.  Round 1 found the read factor 1.979 / 1.9998 the same way on its k_blend; the factor measured here is
applied to every kernel's FETCH_SIZE (`read_scale`), WRITE_SIZE is used as it is (`write_ratio` says how exact it is).

  python tools/pmc_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write --pixels 2073600 --spp 16 --active-pixels N --frames F
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
    ap.add_argument("--active-pixels", type=int, default=0, help="pixels k_classify kept (ft_stats: (rays_primary - rays_primary_culled) / spp)")
    ap.add_argument("--frames", type=int, default=7, help="frames rendered by the profiled command (every ft_render / enqueue)")
    ap.add_argument("--f64-frames", type=int, default=None, help="how many of them are FP64 frames (the others are RGBA8: 4 B per pixel out)")
    args = ap.parse_args()
    fetch, write = load(args.fetch_dir, "FETCH_SIZE"), load(args.write_dir, "WRITE_SIZE")
    f64 = args.frames if args.f64_frames is None else args.f64_frames
    known_write = args.pixels * (24.0 * f64 + 4.0 * (args.frames - f64))   # k_resolve writes every pixel of the frame once
    known_read = (24.0 * args.spp * args.active_pixels + 4.0 * args.pixels + 8.0 * args.pixels / 64.0) * args.frames
    got_write, got_read = sum(write.get("k_resolve", [])), sum(fetch.get("k_resolve", []))
    read_scale = known_read / got_read if got_read > 0 else args.read_scale
    note = "measured on k_resolve in this run"
    if not 1.5 <= read_scale <= 2.5:
        read_scale, note = args.read_scale, f"k_resolve gave {known_read / max(1.0, got_read):.3f}: outside 1.5..2.5, fell back to --read-scale"
    out = {"calibration": {"read_scale": read_scale, "read_scale_source": note, "k_resolve_known_read_bytes_all_frames": known_read, "k_resolve_FETCH_SIZE_all_frames": got_read,
                           "k_resolve_known_write_bytes_all_frames": known_write, "k_resolve_WRITE_SIZE_all_frames": got_write,
                           "write_ratio": got_write / known_write if known_write else None}}
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
