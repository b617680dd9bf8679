#!/usr/bin/env python3
"""Per-launch durations from a rocprofv3 --kernel-trace CSV, in launch order (k_* kernels only)."""
import csv
import glob
import re
import sys

for path in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            m = re.search(r"\b(k_\w+)", r["Kernel_Name"])
            if m:
                rows.append((int(r["Start_Timestamp"]), m.group(1), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r["VGPR_Count"], r["LDS_Block_Size"], r["Grid_Size_X"]))
    rows.sort()
    t0 = rows[0][0]
    for t, name, us, vgpr, lds, grid in rows[: int(sys.argv[2]) if len(sys.argv) > 2 else 60]:
        print(f"{(t - t0) / 1e3:10.1f} us  {name:16s} {us:9.1f} us  vgpr {vgpr} lds {lds} grid {grid}")
