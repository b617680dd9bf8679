#!/usr/bin/env python3
"""Per-kernel HBM traffic from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE), collected as
MI355X_MICROARCH.md §HBM prescribes: separate --pmc passes, values in KiB, and on gfx950 FETCH_SIZE
under-reports coalesced streaming reads (exactly 1/2 for 16 B/lane).  This path reads 8 B/lane, an
uncalibrated width, so the read side is calibrated in the same pass on a kernel whose bytes are known:
k_blend reads exactly 24*spp B and writes 24 B per pixel.

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
    ap.add_argument("--chunks", type=int, default=0, help="launches of k_blend per frame (default: counted)")
    args = ap.parse_args()
    fetch, write = load(args.fetch_dir, "FETCH_SIZE"), load(args.write_dir, "WRITE_SIZE")
    n_blend = len(fetch.get("k_blend", [])) or 1
    known_read = 24.0 * args.spp * args.pixels / n_blend          # per k_blend launch
    known_write = 24.0 * args.pixels / n_blend
    blend_read = sum(fetch["k_blend"]) / n_blend
    blend_write = sum(write["k_blend"]) / max(1, len(write["k_blend"]))
    read_scale = known_read / blend_read                           # correction for 8 B/lane coalesced reads
    out = {"calibration": {"kernel": "k_blend", "known_read_bytes_per_launch": known_read, "FETCH_SIZE_bytes_per_launch": blend_read,
                           "read_scale": read_scale, "known_write_bytes_per_launch": known_write, "WRITE_SIZE_bytes_per_launch": blend_write,
                           "write_ratio": blend_write / known_write}}
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
