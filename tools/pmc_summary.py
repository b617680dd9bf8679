#!/usr/bin/env python3
"""Sum rocprofv3 --pmc counters per kernel: python tools/pmc_summary.py <dir>"""
import csv
import glob
import re
import sys
from collections import defaultdict

acc = defaultdict(lambda: defaultdict(float))
n = defaultdict(int)
for path in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    with open(path) as f:
        for r in csv.DictReader(f):
            m = re.search(r"\b(k_\w+)[<(]", r["Kernel_Name"])
            if not m:
                continue
            acc[m.group(1)][r["Counter_Name"]] += float(r["Counter_Value"])
            n[(m.group(1), r["Counter_Name"])] += 1
for k, d in acc.items():
    print(k, {c: f"{v:.4g}" for c, v in d.items()}, "launches", max(n[(k, c)] for c in d))
