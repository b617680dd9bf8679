#!/usr/bin/env python3
"""Registers, scratch and occupancy of every kernel of ft_kernels.hip as the compiler reports them (no GPU needed):
python tools/resource_usage.py > profiles/<tag>_resource_usage.json"""
import json
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = subprocess.run(["make", "-C", os.path.join(ROOT, "functracer_amd", "csrc"), "resource-usage"], capture_output=True, text=True).stderr
kernels, cur = {}, None
for line in out.splitlines():
    m = re.search(r"remark: Function Name: (\S+)", line)
    if m:
        name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
        name = re.sub(r"ftk::\(anonymous namespace\)::", "", name)
        name = re.sub(r"^void ", "", name).split("(")[0]
        cur = kernels.setdefault(name, {})
        continue
    m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\S+) \[-Rpass", line)
    if m and cur is not None:
        key = {"TotalSGPRs": "sgprs", "VGPRs": "vgprs", "ScratchSize": "scratch_bytes_per_lane", "Occupancy": "waves_per_simd", "SGPRs Spill": "sgpr_spills",
               "VGPRs Spill": "vgpr_spills", "LDS Size": "lds_static_bytes"}.get(m.group(1).strip())
        if key:
            cur[key] = int(m.group(2))
print(json.dumps(kernels, indent=1, sort_keys=True))
