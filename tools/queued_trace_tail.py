"""Start / end of the last kernels in a rocprofv3 --kernel-trace CSV (queued frames): python tools/queued_trace_tail.py dir [n]"""
import csv, glob, re, sys
path = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[-1]
rows = sorted(csv.DictReader(open(path)), key=lambda r: int(r["Start_Timestamp"]))
rows = [r for r in rows if "k_" in r["Kernel_Name"]]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 12
t0 = int(rows[-n]["Start_Timestamp"])
for r in rows[-n:]:
    m = re.search(r"(k_\w+)", r["Kernel_Name"]); s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"{m.group(1):12s} start {(s - t0) / 1e3:8.1f} end {(e - t0) / 1e3:8.1f} dur {(e - s) / 1e3:7.1f} queue {r.get('Queue_Id', '?')}")
