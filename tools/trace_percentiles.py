"""Percentiles of per-kernel durations in a rocprofv3 --kernel-trace CSV: python tools/trace_percentiles.py dir"""
import csv, glob, re, sys
import numpy as np
path = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[-1]
d = {}
for r in csv.DictReader(open(path)):
    m = re.search(r"(k_\w+)", r["Kernel_Name"])
    if m: d.setdefault(m.group(1), []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in d.items():
    v = np.array(v)
    print(f"{k:12s} n {len(v):5d} mean {v.mean():8.1f} median {np.median(v):8.1f} p10 {np.percentile(v, 10):8.1f} p90 {np.percentile(v, 90):8.1f} max {v.max():8.1f} us; calls above 2x median: {(v > 2 * np.median(v)).sum()}")
