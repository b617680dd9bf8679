#!/usr/bin/env python3
"""VALU occupancy and FP64 rate per kernel from two rocprofv3 PMC passes of the same command:
  pass A: SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64
  pass B: SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE
SQ_ACTIVE_INST_VALU counts quad-cycles over all SIMDs; GRBM_GUI_ACTIVE is summed over the 8 XCDs (checked against the kernel
trace: value / 8 = kernel duration in cycles).  valu_busy = 4 * ACTIVE_INST_VALU / (1024 SIMDs * GUI_ACTIVE / 8).

  python tools/pmc_valu.py gpurun_out/pmc_va gpurun_out/pmc_vb [--clock-ghz 2.4]
"""
import argparse
import csv
import glob
import json
import os
import re
from collections import defaultdict


def load(directory):
    acc = defaultdict(lambda: defaultdict(float))
    n = defaultdict(int)
    for path in glob.glob(os.path.join(directory, "**", "*counter_collection.csv"), recursive=True):
        with open(path) as f:
            for r in csv.DictReader(f):
                m = re.search(r"\b(k_\w+)[<(]", r["Kernel_Name"])
                if not m:
                    continue
                acc[m.group(1)][r["Counter_Name"]] += float(r["Counter_Value"])
                n[(m.group(1), r["Counter_Name"])] += 1
    return acc, n


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("dir_a")
    ap.add_argument("dir_b")
    ap.add_argument("--clock-ghz", type=float, default=2.4)
    args = ap.parse_args()
    a, na = load(args.dir_a)
    b, nb = load(args.dir_b)
    out = {"method": "valu_busy = 4*SQ_ACTIVE_INST_VALU / (1024 * GRBM_GUI_ACTIVE/8); fp64 = (ADD+MUL+TRANS+2*FMA)*64 lanes over the kernel's GUI time at the stated clock (upper bound: all lanes counted)",
           "clock_ghz": args.clock_ghz}
    for k in sorted(set(a) & set(b)):
        gui = b[k].get("GRBM_GUI_ACTIVE", 0.0) / 8.0
        if gui <= 0:
            continue
        valu = a[k].get("SQ_INSTS_VALU", 0.0)
        f64 = a[k].get("SQ_INSTS_VALU_ADD_F64", 0.0) + a[k].get("SQ_INSTS_VALU_MUL_F64", 0.0) + a[k].get("SQ_INSTS_VALU_TRANS_F64", 0.0)
        fma = a[k].get("SQ_INSTS_VALU_FMA_F64", 0.0)
        seconds = gui / (args.clock_ghz * 1e9)
        out[k] = {
            "launches": nb[(k, "GRBM_GUI_ACTIVE")],
            "valu_busy_frac": round(4.0 * b[k].get("SQ_ACTIVE_INST_VALU", 0.0) / (1024.0 * gui), 4),
            "wait_inst_frac_of_wave_cycles": round(b[k].get("SQ_WAIT_INST_ANY", 0.0) / max(1.0, b[k].get("SQ_WAVE_CYCLES", 0.0)), 4),
            "valu_insts": valu, "salu_insts": a[k].get("SQ_INSTS_SALU", 0.0), "smem_insts": a[k].get("SQ_INSTS_SMEM", 0.0),
            "fp64_arith_share_of_valu": round((f64 + fma) / max(1.0, valu), 4),
            "fp64_tflops_upper": round((f64 + 2.0 * fma) * 64.0 / seconds / 1e12, 3),
        }
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
