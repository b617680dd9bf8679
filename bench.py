#!/usr/bin/env python3
"""Headline benchmark: Mrays/s (primary + shadow + reflect) on scenes/bunny.scene at
1920x1080x16spp (BASELINE.json), one process per GPU, frame image-tiled over the ranks.

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...

Scaling is WEAK: the path shards into independent samples, so N ranks render a frame with N times the
samples per pixel (1920x1080 x 16*N spp; N = 1 is exactly the BASELINE configuration) and every rank keeps the
per-GPU work of the single-GPU run.  `--strong` keeps the frame fixed at 16 spp instead.

A step = one full frame: every rank renders its interleaved 8-row bands of the frame through the
C ABI with the frame left in HBM (out_rgb = NULL); scene, jitter pattern and pixel lists are
HBM-resident before the timed region.  The K steps are bracketed by barrier + device synchronise
and the slowest rank's wall time counts: `value` = rays traced by all ranks / that time.  After the
timed region every rank fetches its bands once (ft_fetch_frame) and rank 0 gathers them on the host;
the PCIe-inclusive frame time is reported as `frame_ms_incl_copy`, never as `value`.
Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0            # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md
RAY_REC, HIT_REC = 60, 16         # bytes, functracer_amd/csrc/ft_device.h
SURVEY_BYTES_PER_RAY = 2 * RAY_REC + 2 * HIT_REC   # SURVEY 8(d)'s stored-wavefront model restated with this build's sizeof


import contextlib


@contextlib.contextmanager
def stdout_to_stderr():
    """gloo announces its connections on stdout; rank 0's stdout must carry exactly one JSON line."""
    sys.stdout.flush()
    saved = os.dup(1)
    os.dup2(2, 1)
    try:
        yield
    finally:
        sys.stdout.flush()
        os.dup2(saved, 1)
        os.close(saved)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--scene", default="bunny")
    ap.add_argument("--res", type=int, nargs=2, default=None)
    ap.add_argument("--spp", type=int, default=None)
    ap.add_argument("--cpu-baseline-seconds", type=float, default=15.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--strong", action="store_true", help="keep the frame fixed as ranks are added (strong scaling)")
    ap.add_argument("--blocking", action="store_true", help="render the timed frames one synchronous ft_render at a time instead of queuing them")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")

    import numpy as np
    import torch
    import torch.distributed as dist

    import functracer_amd as ft
    from functracer_amd import tiling

    n_visible = max(1, torch.cuda.device_count())
    device = local_rank % n_visible               # more ranks than GPUs only happens in rehearsals on a 1-GPU box
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(device)
        # The data path has no exchange step, so no RCCL collective exists to run over xGMI; the process group only
        # carries the bench contract's barrier, three timing scalars and the host-side band gather: gloo on CPU tensors
        # (FT_BENCH_BACKEND=nccl switches the barrier / scalars to RCCL).
        backend = os.environ.get("FT_BENCH_BACKEND", "gloo")
        with stdout_to_stderr():
            dist.init_process_group(backend)
            host_group = dist.group.WORLD if backend == "gloo" else dist.new_group(backend="gloo")
            dist.barrier()

    def barrier_sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    scene = ft.parse_scene_file(os.path.join(ROOT, "scenes", args.scene + ".scene"))
    res_h, res_v = args.res if args.res else scene.resolution
    base_spp = args.spp if args.spp else scene.samples
    spp = base_spp if args.strong else base_spp * world          # weak scaling: per-GPU samples stay those of the N = 1 run
    jitter = ft.jitter_pattern(spp)
    ctx = ft.Context(device=device)
    scene.lower(ctx)
    bands = None if world == 1 else tiling.bands_for_rank(res_h, res_v, rank, world)
    frame = np.zeros((res_v, res_h, 3))

    def step():
        _, st = ctx.render(scene.camera, res_h, res_v, spp, jitter, tiles=bands, fetch=False)
        return st

    for _ in range(args.warmup):
        step()
    barrier_sync()
    t0 = time.perf_counter()
    rays = 0
    kernel_ms = trace_ms = 0.0
    k_times = {"other": 0.0, "closest": 0.0, "shade": 0.0, "blend": 0.0}   # "other": memsets, k_classify, k_blend, statistics (events bracket k_closest / k_shade only)
    k_launch = dict.fromkeys(k_times, 0)
    st = None
    if args.blocking:                                              # the reference's own flow: one synchronous frame after the other
        for _ in range(args.steps):
            st = step()
            rays += st["rays_traced"]
            kernel_ms += st["kernel_ms"]
            trace_ms += st["trace_kernel_ms"]
            for k, v in ctx.kernel_times().items():
                k_times[k] += v["ms"]
                k_launch[k] += v["launches"]
    else:                                                          # frames queued back to back (ft_render_enqueue): the host prepares frame
        for _ in range(args.steps):                                # k+1 while frame k runs; every frame is the same full frame as above
            ctx.render_enqueue(scene.camera, res_h, res_v, spp, jitter, tiles=bands)
        st = ctx.wait()                                            # statistics of the last frame; stage times summed over all of them
        rays = st["rays_traced"] * args.steps
        for k, v in ctx.kernel_times().items():
            k_times[k] += v["ms"]
            k_launch[k] += v["launches"]
        kernel_ms = sum(k_times.values())
        trace_ms = k_times["closest"] + k_times["shade"]
    barrier_sync()
    wall = time.perf_counter() - t0

    # slowest rank's times; total rays over ranks
    vals = torch.tensor([wall, kernel_ms], dtype=torch.float64)
    tot = torch.tensor([float(rays)], dtype=torch.float64)
    if world > 1:
        on_gpu = dist.get_backend() == "nccl"
        if on_gpu:
            vals = vals.cuda(); tot = tot.cuda()
        dist.all_reduce(vals, op=dist.ReduceOp.MAX)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        vals = vals.cpu(); tot = tot.cpu()
    wall_max, kernel_ms_max = float(vals[0]), float(vals[1])
    rays_total = float(tot[0])

    # D2H of this rank's bands + gather on the host of rank 0 (outside the timed region; reported separately)
    tg = time.perf_counter()
    ctx.fetch_frame(frame)
    copy_ms = (time.perf_counter() - tg) * 1e3
    full = tiling.gather_frame(frame, res_h, res_v, rank, world, group=host_group if world > 1 else None)
    gather_ms = (time.perf_counter() - tg) * 1e3
    if rank == 0 and os.environ.get("FT_BENCH_PNG"):
        ft.write_png(os.environ["FT_BENCH_PNG"], ft.quantise_rgba8(full))

    if rank == 0:
        steps = args.steps
        mrays_wall = rays_total / wall_max / 1e6                        # inputs and frame resident in HBM; barrier-to-barrier wall time
        mrays_kernel = rays_total / (kernel_ms_max * 1e-3) / 1e6       # same rays over the HIP-event kernel time of the slowest rank
        dom = max(("closest", "shade"), key=lambda k: k_times[k])
        dom_avg_ms = k_times[dom] / max(1, k_launch[dom])
        # Algorithmic bytes of the dominant kernel = what its launches have to move by construction of the pipeline (ft_stats,
        # DESIGN.md): primary rays are regenerated (4 B pixel id + 1 B flag), only hits and reflection rays have records in HBM.
        rays_rank = rays / steps
        launches_per_step = max(1, k_launch[dom] / steps)
        algo_bytes_per_launch = st["algorithmic_bytes_" + dom] / launches_per_step
        achieved = algo_bytes_per_launch / (dom_avg_ms * 1e-3) / 1e9
        rays_per_launch = (st["rays_primary"] + st["rays_reflect"] if dom == "closest" else st["rays_shadow"]) / launches_per_step
        survey_model = SURVEY_BYTES_PER_RAY * rays_per_launch / (dom_avg_ms * 1e-3) / 1e9
        out = {
            "metric": "Mrays/s (primary+shadow+reflect) at 1920x1080x16spp; frame ms",
            "value": round(mrays_wall, 3),
            "unit": "Mrays/s",
            "n_gpus": world,
            "steps": steps,
            "warmup": args.warmup,
            "ms_per_step": round(wall_max / steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "strong" if args.strong else "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": f"scenes/{args.scene}.scene {res_h}x{res_v}x{spp}spp ({base_spp} spp per GPU-share), depth 8, synthetic stand-in mesh, seeded jitter",
                       "parallelism": f"image-tiled, {world} rank(s) x interleaved {tiling.BAND_ROWS}-row bands, no collective on the data path"},
            "kernel_ms_per_step": round(kernel_ms_max / steps, 4),
            "value_kernel_only": round(mrays_kernel, 3),
            "frame_ms_incl_copy": round(wall_max / steps * 1e3 + gather_ms, 4),
            "d2h_copy_ms": round(copy_ms, 3),
            "gather_ms": round(gather_ms, 3),
            "roofline": {"bound": "hbm", "kernel": "k_" + dom, "achieved": round(achieved, 3), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBPS, 6), "traffic": None,
                         "avg_launch_ms": round(dom_avg_ms, 4), "algorithmic_bytes_per_launch": round(algo_bytes_per_launch),
                         "rays_per_launch": round(rays_per_launch),
                         "survey_stored_wavefront_model": {"bytes_per_ray": SURVEY_BYTES_PER_RAY, "GBps": round(survey_model, 1),
                                                           "note": "what a design that stores every ray and hit record would move; this build regenerates primary rays"}},
            "per_kernel_ms_per_step": {k: round(v / steps, 4) for k, v in k_times.items()},
        }
        traffic_file = os.path.join(ROOT, "profiles", f"pmc_traffic_{args.scene}.json")
        if os.path.exists(traffic_file) and world == 1 and not args.res and not args.spp:
            with open(traffic_file) as f:
                t = json.load(f).get("k_" + dom)
            if t:   # HBM bytes per launch from the committed rocprofv3 FETCH_SIZE / WRITE_SIZE passes of this same command
                out["roofline"]["traffic"] = round(t["traffic_bytes_per_launch"])
                out["roofline"]["traffic_source"] = os.path.relpath(traffic_file, ROOT)
        valu_file = os.path.join(ROOT, "profiles", f"pmc_valu_{args.scene}.json")
        if os.path.exists(valu_file) and world == 1 and not args.res and not args.spp:
            with open(valu_file) as f:
                v = json.load(f).get("k_" + dom)
            if v:   # informational: the path is FP64-VALU / scalar / latency bound, not HBM bound (tools/pmc_valu.py, same command)
                out["roofline"]["valu"] = {"busy_frac": v["valu_busy_frac"], "fp64_tflops_upper": v["fp64_tflops_upper"], "fp64_peak_tflops": 78.6,
                                           "salu_per_valu_inst": round(v["salu_insts"] / max(1.0, v["valu_insts"]), 3), "source": os.path.relpath(valu_file, ROOT)}
        out["rays_per_frame"] = {"traced_rank0": int(rays_rank), "traced_all_ranks": int(rays_total / steps),
                                 "reference_equivalent_rank0": st["rays_reference_equivalent"],
                                 "primary_resolved_per_64_pixel_block_rank0": int(st["rays_primary_culled"])}
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(scene, res_h, res_v, spp, jitter, args.cpu_baseline_seconds)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    ctx.close()


def cpu_baseline(scene, res_h, res_v, spp, jitter, budget_s):
    """The CPU oracle (a C++ port of the F# algorithm; the F# toolchain does not exist here) timed on
    this box's host cores on a bounded sample: interleaved 8-row bands of the same frame."""
    import numpy as np

    from functracer_amd import tiling
    from oracle import ft_oracle_py as O
    orc = O.Oracle()
    scene.lower(orc)
    cores = os.cpu_count() or 1
    # (always the N = 1 workload's spp) calibrate on a thin sample, then size the real sample to ~budget_s
    probe = tiling.bands_for_rank(res_h, res_v, 0, 64)
    frame = np.zeros((res_v, res_h, 3))
    _, st = orc.render(scene.camera, res_h, res_v, spp, jitter, tiles=probe, threads=cores, out=frame)
    rate = st["rays_traced"] / (st["wall_ms"] * 1e-3)
    frac = min(1.0, max(1.0 / 64, budget_s * rate / (st["rays_traced"] * 64)))
    stride = max(1, int(round(1.0 / frac)))
    sample = tiling.bands_for_rank(res_h, res_v, 0, stride)
    _, st = orc.render(scene.camera, res_h, res_v, spp, jitter, tiles=sample, threads=cores, out=frame)
    return {"value": round(st["rays_traced"] / (st["wall_ms"] * 1e-3) / 1e6, 4), "unit": "Mrays/s", "cores": int(st["threads"]), "kind": "port",
            "sample": f"every {stride}th 8-row band of the same {res_h}x{res_v}x{spp}spp frame ({st['rays_traced']} rays, {st['wall_ms'] / 1e3:.1f} s), "
                      "C++ restatement of the F# algorithm (oracle/ft_oracle.cpp); the F# toolchain is unavailable"}


if __name__ == "__main__":
    main()
