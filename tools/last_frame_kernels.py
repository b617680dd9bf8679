#!/usr/bin/env python3
"""Per-dispatch durations of the LAST frame in a rocprofv3 --kernel-trace CSV: python tools/last_frame_kernels.py dir"""
import csv
import glob
import os
import re
import sys

path = sorted(glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True))[-1]
rows = list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
starts = [i for i, r in enumerate(rows) if "fillBuffer" in r["Kernel_Name"] or "k_classify" in r["Kernel_Name"]]
# the last frame starts at the last fill that is followed by k_classify / k_primary
last = max(i for i in range(len(rows)) if "k_primary" in rows[i]["Kernel_Name"])
begin = last
while begin > 0 and ("k_classify" in rows[begin - 1]["Kernel_Name"] or "fillBuffer" in rows[begin - 1]["Kernel_Name"]):
    begin -= 1
t0 = int(rows[begin]["Start_Timestamp"])
for r in rows[begin:]:
    m = re.search(r"(k_\w+)", r["Kernel_Name"])
    name = m.group(1) if m else r["Kernel_Name"][:24]
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"{name:16s} start {(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:8.1f} us  grid {r.get('Grid_Size', '?'):>8} vgpr {r.get('VGPR_Count', '?')}")
