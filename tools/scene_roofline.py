#!/usr/bin/env python3
"""One roofline block per profiled scene (tools/profile_scene.sh): python tools/scene_roofline.py gpurun_out/<tag>_<scene>x<spp>

Per tracing kernel of the scene: the SURVEY 8(d) fraction (192 B per ray the kernel traces / its mean launch time from the
un-profiled bench run, against 8 TB/s), the bytes this build's layout has to move, the HBM traffic the counters saw
(FETCH_SIZE calibrated on k_resolve, WRITE_SIZE as it is), and the FP64 vector fraction with VALU busy / wait shares."""
import csv
import json
import os
import re
import sys

HBM_PEAK_GBPS, FP64_PEAK_TFLOPS, SURVEY_BYTES_PER_RAY = 8000.0, 78.6, 192
T = sys.argv[1]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def last_json(path):
    with open(path) as f:
        return json.loads(f.read().strip().splitlines()[-1])


bench = last_json(T + "_bench.json")
traffic = json.load(open(T + "_pmc_traffic.json"))
valu = json.load(open(T + "_pmc_valu.json"))
stats = {}
with open(T + "_kernel_stats.csv") as f:
    for r in csv.DictReader(f):
        m = re.search(r"\b(k_\w+)(<[^>]*>)?", r["Name"])
        if m:
            s = stats.setdefault(m.group(1), {"calls": 0, "total_ns": 0.0, "variant": m.group(0)})
            s["calls"] += int(r["Calls"]); s["total_ns"] += float(r["TotalDurationNs"])
res = {}
for cand in sorted(f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("_resource_usage.json")):
    res = json.load(open(os.path.join(ROOT, "profiles", cand)))       # the newest committed one
rf = bench["rays_per_frame"]
steps = bench["steps"]
per_kernel = bench["per_kernel_ms_per_step"]
launches = bench.get("per_kernel_launches_per_step", {})
generated = rf["primary_listed_all_ranks"] - rf["primary_never_generated_all_ranks"]
rays = {"k_primary": generated + rf.get("shadow_primary_rank0", 0),
        "k_bounce": rf["reflect_rank0"] + rf["shadow_rank0"] - rf.get("shadow_primary_rank0", 0)}
layout = {"k_primary": bench.get("layout_bytes_per_frame", {}).get("primary"), "k_bounce": bench.get("layout_bytes_per_frame", {}).get("shade")}
ms = {"k_primary": per_kernel.get("primary", 0.0), "k_bounce": per_kernel.get("shade", 0.0)}
n_launch = {"k_primary": launches.get("primary", 1), "k_bounce": launches.get("shade", 1)}
queued = last_json(T + "_bench_queued.json") if os.path.exists(T + "_bench_queued.json") else bench
out = {"workload": bench["config"]["workload"], "ms_per_step": queued["ms_per_step"], "ms_per_step_one_stream": bench["ms_per_step"],
       "note": "kernel times: frames on ONE stream, every kernel alone on the device (FT_OPTS=mains=1,classify_ahead=0,resolve_aside=0); ms_per_step: the default, pipelined frame period", "kernel_ms_per_step": bench["kernel_ms_per_step"],
       "mrays_s_traced": queued["value"], "rays_traced_per_frame": rf["traced_all_ranks"], "per_kernel_ms_per_step": per_kernel,
       "hbm_calibration": traffic.get("calibration", {}), "kernels": {}}
for k in ("k_primary", "k_bounce"):
    if ms[k] <= 0 or rays[k] <= 0:
        continue
    sec = ms[k] * 1e-3                                               # per frame: all launches of the kernel together
    gbps = SURVEY_BYTES_PER_RAY * rays[k] / sec / 1e9
    row = {"ms_per_frame": round(ms[k], 4), "launches_per_frame": n_launch[k], "rays_per_frame": int(rays[k]), "grays_per_s": round(rays[k] / sec / 1e9, 3),
           "survey_8d": {"bytes_per_frame": SURVEY_BYTES_PER_RAY * int(rays[k]), "GBps": round(gbps, 1), "frac_of_hbm_peak": round(gbps / HBM_PEAK_GBPS, 4)}}
    if k in stats:
        row["rocprof"] = {"variant": stats[k]["variant"], "calls": stats[k]["calls"], "avg_us": round(stats[k]["total_ns"] / stats[k]["calls"] / 1e3, 2)}
        r = res.get(stats[k]["variant"].replace("true", "true").strip())
        if r:
            row["registers"] = r
    if layout[k]:
        g = layout[k] / sec / 1e9
        row["layout"] = {"bytes_per_frame": int(layout[k]), "GBps": round(g, 1), "frac_of_hbm_peak": round(g / HBM_PEAK_GBPS, 4)}
    t = traffic.get(k)
    if t:
        # the counter passes render 24 frames; launches counted there / 24 = launches per frame
        per_frame = t["traffic_bytes_per_launch"] * t["launches"] / 24.0
        row["hbm_traffic"] = {"bytes_per_frame": int(per_frame), "read": int(t["fetch_corrected_bytes_per_launch"] * t["launches"] / 24.0),
                              "written": int(t["write_bytes_per_launch"] * t["launches"] / 24.0), "GBps": round(per_frame / sec / 1e9, 1),
                              "frac_of_hbm_peak": round(per_frame / sec / 1e9 / HBM_PEAK_GBPS, 4)}
        if layout[k]:
            row["hbm_traffic"]["over_layout"] = round(per_frame / layout[k], 3)
            row["hbm_traffic"]["scratch_and_other_bytes"] = int(per_frame - layout[k])
    v = valu.get(k)
    if v:
        row["fp64_valu"] = {"valu_busy_frac": v["valu_busy_frac"], "wait_inst_frac_of_wave_cycles": v["wait_inst_frac_of_wave_cycles"],
                            "fp64_tflops_upper": v["fp64_tflops_upper"], "frac_of_fp64_peak": round(v["fp64_tflops_upper"] / FP64_PEAK_TFLOPS, 4),
                            "fp64_arith_share_of_valu": v["fp64_arith_share_of_valu"], "salu_per_valu_inst": round(v["salu_insts"] / max(1.0, v["valu_insts"]), 3)}
    out["kernels"][k] = row
print(json.dumps(out, indent=1))
