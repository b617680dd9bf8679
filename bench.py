#!/usr/bin/env python3
"""Headline benchmark: Mrays/s (primary + shadow + reflect, rays the GPU really traced) on scenes/bunny.scene,
one process per GPU, frame image-tiled over the ranks (no collective on the data path).

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...

Workloads (BASELINE.json):
  headline  1920x1080 x 16 spp   configs[3]'s frame, the one `metric` is quoted on; the timed workload at N = 1
  config5   3840x2160 x 64 spp   configs[4], the frame north_star names for 8 GPUs; the timed workload at N > 1 (STRONG scaling:
                                 the frame is fixed, every rank renders its interleaved 8-row bands of it)
  weak      1920x1080 x 16*N spp per-GPU work of the N = 1 run (reported beside the others, never as `value` unless --weak)
Every line carries all of them (`workloads`), each timed the same way, so any two N can be compared like for like.

A step = one full frame of the timed workload: every rank renders its bands through the C ABI with the frame left in HBM
(out_rgb = NULL); scene, jitter pattern and pixel lists are HBM-resident before the timed region.  The K steps are bracketed
by barrier + device synchronise and the slowest rank's wall time counts.  Before the W warm-up steps untimed frames are queued for
--prewarm-ms (200): a fresh box idles at low clocks, and K = 20 steps are 6 ms.

`value` = rays the devices TRACED (ft_stats.rays_traced: generated primaries + shadow + reflection rays; primaries of 64-pixel
blocks that k_classify proves empty are never generated and are NOT counted) / that time.  `value_reference_equivalent` is the
rate in rays the F# recursion would have traced for the same frame (Shading.fs:109-147: every listed pixel x spp primaries, L
shadow rays per hit, L reflection rays per reflective hit) - the number to hold against the reference's own clock.

At N = 1 rank 0 also renders a bounded band sample of the same frame with the CPU oracle (`cpu_baseline`) and the line reports
the parity of the timed GPU frame against it (`parity`).  Rank 0 prints ONE JSON line.
"""
import argparse
import contextlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0            # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md
FP64_VECTOR_PEAK_TFLOPS = 78.6    # MI355X vector FP64; the path is FP64 VALU work, no MFMA
SURVEY_BYTES_PER_RAY = 192        # RayRec 80 B + HitRec 16 B, each written once and read once
PIXEL_BYTES = 24                  # one FP64 RGB store per pixel


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
    ap.add_argument("--cpu-baseline-seconds", type=float, default=8.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--workload", choices=["auto", "headline", "config5", "weak"], default="auto",
                    help="which workload the K timed steps (and `value`) are: auto = headline at N = 1, config5 (strong) at N > 1")
    ap.add_argument("--strong", action="store_true", help="same as --workload headline: the 1080p x 16 frame strong-scaled")
    ap.add_argument("--weak", action="store_true", help="same as --workload weak")
    ap.add_argument("--no-alone", action="store_true", help="skip the untimed frames that measure each kernel with the frame pipeline off (roofline.alone); the counter passes use it: a known number of frames")
    ap.add_argument("--side-steps", type=int, default=5, help="frames timed for each workload other than the primary one (0 = skip them)")
    ap.add_argument("--prewarm-ms", type=float, default=200.0, help="untimed frames queued for this long before the W warm-up steps of every workload (clock ramp); 0 = none")
    ap.add_argument("--no-boundary", action="store_true", help="skip the boundary legs (frames copied to the host): the counter passes of the profile scripts want a known number of frames")
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
    host_group = None
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(device)
        # The data path has no exchange step, so no RCCL collective exists to run over xGMI; the process group only
        # carries the bench contract's barrier, the timing scalars and the host-side band gather: gloo on CPU tensors
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

    def reduce_max_sum(maxes, sums):
        vals = torch.tensor(maxes, dtype=torch.float64)
        tot = torch.tensor(sums, dtype=torch.float64)
        if world > 1:
            on_gpu = dist.get_backend() == "nccl"
            if on_gpu:
                vals = vals.cuda(); tot = tot.cuda()
            dist.all_reduce(vals, op=dist.ReduceOp.MAX)
            dist.all_reduce(tot, op=dist.ReduceOp.SUM)
            vals = vals.cpu(); tot = tot.cpu()
        return [float(v) for v in vals], [float(v) for v in tot]

    scene = ft.parse_scene_file(os.path.join(ROOT, "scenes", args.scene + ".scene"))
    base_res = tuple(args.res) if args.res else tuple(scene.resolution)
    base_spp = args.spp if args.spp else scene.samples
    ctx = ft.Context(device=device)
    for kv in filter(None, os.environ.get("FT_OPTS", "").split(",")):      # FT_OPTS="zero_fill_skip=0,...": A/B and counter runs
        k, v = kv.split("=")
        ctx.set_option(k, int(v))
    scene.lower(ctx)

    primary = args.workload
    if args.strong:
        primary = "headline"
    if args.weak:
        primary = "weak"
    if primary == "auto":
        primary = "headline" if world == 1 else "config5"
    custom = bool(args.res or args.spp or args.scene != "bunny")
    specs = {"headline": (base_res, base_spp, "strong"),
             "config5": ((3840, 2160), 64, "strong"),
             "weak": (base_res, base_spp * world, "weak")}
    if custom and primary == "config5" and world == 1:
        primary = "headline"

    def run(name, steps, warmup):
        """Time `steps` frames of workload `name`; returns the all-rank figures (identical on every rank) + rank-local details."""
        (res_h, res_v), spp, scaling = specs[name]
        jitter = ft.jitter_pattern(spp)
        bands = None if world == 1 else tiling.bands_for_rank(res_h, res_v, rank, world)

        def step():
            _, st = ctx.render(scene.camera, res_h, res_v, spp, jitter, tiles=bands, fetch=False)
            return st

        prewarm_frames = 0
        if args.prewarm_ms > 0:                                    # clocks: a fresh box idles low and 20 frames are 7 ms - queue frames for a while first,
            t_end = time.perf_counter() + args.prewarm_ms / 1e3    # untimed like the W warm-up steps that follow, so the K steps run at the clocks a stream of frames sees
            while time.perf_counter() < t_end:
                for _ in range(32):
                    ctx.render_enqueue(scene.camera, res_h, res_v, spp, jitter, tiles=bands)
                ctx.wait()
                prewarm_frames += 32
        for _ in range(warmup):
            step()
        barrier_sync()
        t0 = time.perf_counter()
        k_times, k_launch = {}, {}
        st = None
        if args.blocking:                                          # the reference's own flow: one synchronous frame after the other
            for _ in range(steps):
                st = step()
                for k, v in ctx.kernel_times().items():
                    k_times[k] = k_times.get(k, 0.0) + v["ms"]
                    k_launch[k] = k_launch.get(k, 0) + v["launches"]
        else:                                                      # frames queued back to back (ft_render_enqueue): the host prepares frame
            for _ in range(steps):                                 # k+1 while frame k runs; every frame is the same full frame as above
                ctx.render_enqueue(scene.camera, res_h, res_v, spp, jitter, tiles=bands)
            st = ctx.wait()                                        # statistics of the last frame; stage times summed over all of them
            for k, v in ctx.kernel_times().items():
                k_times[k] = v["ms"]
                k_launch[k] = v["launches"]
        barrier_sync()
        wall = time.perf_counter() - t0
        kernel_ms = sum(k_times.values())
        (wall_max, kernel_ms_max), (traced, ref_equiv, culled, listed) = reduce_max_sum(
            [wall, kernel_ms], [float(st["rays_traced"]), float(st["rays_reference_equivalent"]), float(st["rays_primary_culled"]), float(st["rays_primary"])])
        return {"name": name, "res": (res_h, res_v), "spp": spp, "scaling": scaling, "steps": steps, "warmup": warmup, "bands": bands, "jitter": jitter, "prewarm_frames": prewarm_frames,
                "wall_max": wall_max, "kernel_ms_max": kernel_ms_max, "rays_traced_frame": traced, "rays_ref_equiv_frame": ref_equiv,
                "rays_culled_frame": culled, "rays_listed_frame": listed, "st": st, "k_times": k_times, "k_launch": k_launch}

    def summary(r):
        ms = r["wall_max"] / r["steps"] * 1e3
        return {"workload": f"{r['res'][0]}x{r['res'][1]}x{r['spp']}spp", "scaling": r["scaling"], "steps": r["steps"], "ms_per_frame": round(ms, 4),
                "mrays_s_traced": round(r["rays_traced_frame"] / ms / 1e3, 3), "mrays_s_reference_equivalent": round(r["rays_ref_equiv_frame"] / ms / 1e3, 3),
                "rays_traced_per_frame": int(r["rays_traced_frame"]), "kernel_ms_per_frame": round(r["kernel_ms_max"] / r["steps"], 4),
                "timed_region": "the contract's K steps" if r["name"] == primary else f"{r['steps']} frames after 1 warm-up frame, same bracketing"}

    results = {primary: run(primary, args.steps, args.warmup)}
    if args.side_steps > 0 and not custom:
        for name in ("headline", "config5", "weak"):
            if name in results or (name == "weak" and world == 1):
                continue
            results[name] = run(name, args.side_steps, 1)
    P = results[primary]
    res_h, res_v = P["res"]
    spp = P["spp"]

    # The same frames with the frame pipeline switched off (every kernel alone on the device, one stream): what a launch of each kernel
    # takes when nothing runs beside it - the figure a serial rocprofv3 pass shows (profiles/*_serial_kernel_stats.csv).  Not the timed region.
    alone = {}
    if world == 1 and not args.no_alone:
        for k in ("classify_ahead", "resolve_aside"):
            ctx.set_option(k, 0)
        for _ in range(2):
            for _ in range(P["steps"]):
                ctx.render_enqueue(scene.camera, res_h, res_v, spp, P["jitter"], tiles=P["bands"])
            ctx.wait()
            alone = {k: v["ms"] / max(1, v["launches"]) for k, v in ctx.kernel_times().items() if v["launches"]}
        for k, v in {"classify_ahead": 1, "resolve_aside": 1, **{kv.split("=")[0]: int(kv.split("=")[1]) for kv in filter(None, os.environ.get("FT_OPTS", "").split(","))}}.items():
            if k in ("classify_ahead", "resolve_aside"):
                ctx.set_option(k, v)

    # The primary workload's frame: this rank's bands D2H, gathered on the host of rank 0 (outside the timed region).
    _, _ = ctx.render(scene.camera, res_h, res_v, spp, P["jitter"], tiles=P["bands"], fetch=False)
    frame = np.zeros((res_v, res_h, 3))
    ctx.fetch_frame(frame)                                          # first fetch also first-touches the host array: not timed
    tg = time.perf_counter()
    ctx.fetch_frame(frame)
    copy_ms = (time.perf_counter() - tg) * 1e3
    full = tiling.gather_frame(frame, res_h, res_v, rank, world, group=host_group)
    gather_ms = (time.perf_counter() - tg) * 1e3
    # The boundary as the reference uses it (Program.fs:63-68): one blocking call that hands the frame back in host memory.
    boundary = {}
    n_b = 10
    boundary_modes = () if args.no_boundary else ("f64", "rgba8", "f64_pinned", "rgba8_pinned")
    pinned = {} if args.no_boundary else {"f64_pinned": ft.PinnedArray((res_v, res_h, 3)), "rgba8_pinned": ft.PinnedArray((res_v, res_h, 4), dtype=np.uint8)}   # ft_host_alloc: one DMA at link rate
    for mode in boundary_modes:
        buf = pinned[mode].array if mode in pinned else (frame if mode == "f64" else np.zeros((res_v, res_h, 4), dtype=np.uint8))
        call = (lambda: ctx.render(scene.camera, res_h, res_v, spp, P["jitter"], tiles=P["bands"], out=buf)) if mode.startswith("f64") else \
               (lambda: ctx.render_rgba8(scene.camera, res_h, res_v, spp, P["jitter"], tiles=P["bands"], out=buf))
        call()
        barrier_sync()
        tb = time.perf_counter()
        for _ in range(n_b):
            call()
        barrier_sync()
        (b_wall,), _ = reduce_max_sum([time.perf_counter() - tb], [0.0])
        boundary[mode] = round(b_wall / n_b * 1e3, 4)
    # ... and as a host that renders frame after frame would use it: frames queued, each with its copy into page-locked memory behind it
    # (ft_render_enqueue_into, two buffers in turn): the copy of frame N travels while frame N + 1 is traced.
    streaming = {}
    for mode in (() if args.no_boundary else ("f64", "rgba8")):
        shape, dt = ((5, res_v, res_h, 3), np.float64) if mode == "f64" else ((5, res_v, res_h, 4), np.uint8)   # four frames in flight + the one the host would be reading
        with ft.PinnedArray(shape, dtype=dt) as ring:
            for k in range(5):
                ctx.render_enqueue(scene.camera, res_h, res_v, spp, P["jitter"], tiles=P["bands"], rgba8=mode == "rgba8", out=ring[k % 5])
            ctx.wait()
            barrier_sync()
            ts = time.perf_counter()
            n_s = 30
            for k in range(n_s):
                ctx.render_enqueue(scene.camera, res_h, res_v, spp, P["jitter"], tiles=P["bands"], rgba8=mode == "rgba8", out=ring[k % 5])
            ctx.wait()
            barrier_sync()
            (s_wall,), _ = reduce_max_sum([time.perf_counter() - ts], [0.0])
            streaming[mode] = round(s_wall / n_s * 1e3, 4)
    for pa in pinned.values():
        pa.close()
    if rank == 0 and os.environ.get("FT_BENCH_PNG"):
        ft.write_png(os.environ["FT_BENCH_PNG"], ft.quantise_rgba8(full))

    if rank == 0:
        steps = P["steps"]
        st = P["st"]
        k_times, k_launch = P["k_times"], P["k_launch"]
        ms_per_step = P["wall_max"] / steps * 1e3
        mrays_wall = P["rays_traced_frame"] / ms_per_step / 1e3
        # Dominant kernel of rank 0: per-launch figures from the HIP events the library records on its own stream.
        traced_kernels = [k for k in k_times if k not in ("other", "blend")]
        dom = max(traced_kernels, key=lambda k: k_times[k])
        dom_launches = max(1, k_launch[dom])
        dom_avg_ms = k_times[dom] / dom_launches
        launches_per_step = dom_launches / steps
        rays_dom = ft.rays_handled_by(dom, st) / launches_per_step     # rays one launch of the dominant kernel traces
        survey_bytes = SURVEY_BYTES_PER_RAY * rays_dom
        survey_gbps = survey_bytes / (dom_avg_ms * 1e-3) / 1e9
        layout_bytes = st["algorithmic_bytes_" + dom] / launches_per_step if ("algorithmic_bytes_" + dom) in st else None
        frame_bytes = SURVEY_BYTES_PER_RAY * st["rays_traced"] + PIXEL_BYTES * (st["rays_primary"] // max(1, spp))
        kernel_s = P["kernel_ms_max"] / steps * 1e-3
        # Queued frames overlap - a frame's k_primary is dispatched on a second stream while its predecessor's drains, the two share the CUs for
        # a stretch - so the dominant kernel's rate is its bytes over the time the timed region gave each launch (the frame period); the
        # duration of a launch by itself (HIP events, `per_launch`) counts the shared stretch twice, `alone` is the launch with nothing beside it.
        agg_gbps = survey_bytes * launches_per_step / (ms_per_step * 1e-3) / 1e9
        roof = {"bound": "hbm", "kernel": "k_" + dom,
                "achieved": round(agg_gbps, 3), "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(agg_gbps / HBM_PEAK_GBPS, 6), "traffic": None,
                "model": f"SURVEY 8(d): {SURVEY_BYTES_PER_RAY} B per ray x the rays one launch of this kernel traces x its launches in the timed region / the timed region "
                         "(launches of consecutive frames overlap on two streams: see per_launch and alone for one launch's own duration)",
                "avg_launch_ms": round(dom_avg_ms, 4), "rays_per_launch": round(rays_dom), "algorithmic_bytes_per_launch": round(survey_bytes),
                "per_launch": {"GBps": round(survey_gbps, 3), "frac": round(survey_gbps / HBM_PEAK_GBPS, 6),
                               "note": "algorithmic bytes of one launch / its mean duration by HIP events on its own stream in the timed region (avg_launch_ms): with two launches sharing the CUs each takes longer than the period"},
                "frame_model": {"bytes_per_frame": int(frame_bytes), "GBps": round(frame_bytes / (ms_per_step * 1e-3) / 1e9, 2), "frac": round(frame_bytes / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBPS, 6),
                                "note": "8(d)'s whole-frame form: (192 x rays_traced + 24 x pixels) / the frame period"},
                "note": "the path is FP64 vector-ALU / latency bound, not HBM bound (DESIGN.md 5): `real_bound` is the fraction that says how well the kernel uses the chip"}
        if alone.get(dom):
            roof["alone"] = {"avg_launch_ms": round(alone[dom], 4), "GBps": round(survey_bytes / (alone[dom] * 1e-3) / 1e9, 2),
                             "frac": round(survey_bytes / (alone[dom] * 1e-3) / 1e9 / HBM_PEAK_GBPS, 6),
                             "note": "the same launch with the frame pipeline off (options classify_ahead = resolve_aside = 0; frames after the timed region): in the timed region "
                                     "the next frame's k_classify and the previous frame's k_resolve share the CUs with this kernel, which lengthens each launch and shortens the frame period"}
        if layout_bytes is not None:
            g = layout_bytes / (dom_avg_ms * 1e-3) / 1e9
            roof["layout_model"] = {"bytes_per_launch": round(layout_bytes), "GBps": round(g, 2), "frac": round(g / HBM_PEAK_GBPS, 6),
                                    "note": "bytes this build's data layout has to move per launch (ft_stats; DESIGN.md 4): primaries are regenerated, shadow rays stay in registers"}
        # Counter-derived fields are NOT measurements of this run: they come from the committed rocprofv3 PMC passes of this same command.
        prof = {}
        for kind in ("traffic", "valu"):
            path = os.path.join(ROOT, "profiles", f"pmc_{kind}_{args.scene}.json")
            if os.path.exists(path) and world == 1 and not custom and primary == "headline":
                with open(path) as f:
                    prof[kind] = (json.load(f).get("k_" + dom), os.path.relpath(path, ROOT))
        if prof.get("traffic") and prof["traffic"][0]:
            t, src = prof["traffic"]
            roof["traffic"] = round(t["traffic_bytes_per_launch"])
            roof["traffic_source"] = src + " (from_committed_profile: separate --pmc FETCH_SIZE / WRITE_SIZE passes, not this run)"
        if prof.get("valu") and prof["valu"][0]:
            v, src = prof["valu"]
            roof["real_bound"] = {"name": "fp64_valu", "busy_frac": v["valu_busy_frac"], "fp64_tflops_upper": v["fp64_tflops_upper"], "peak_tflops": FP64_VECTOR_PEAK_TFLOPS,
                                  "frac": round(v["fp64_tflops_upper"] / FP64_VECTOR_PEAK_TFLOPS, 4), "salu_per_valu_inst": round(v["salu_insts"] / max(1.0, v["valu_insts"]), 3),
                                  "source": src + " (from_committed_profile)"}
        out = {
            "metric": f"Mrays/s (primary+shadow+reflect) at {res_h}x{res_v}x{spp}spp; frame ms",
            "value": round(mrays_wall, 3),
            "unit": "Mrays/s",
            "n_gpus": world,
            "steps": steps,
            "warmup": P["warmup"],
            "prewarm_ms": args.prewarm_ms,
            "prewarm_frames": P["prewarm_frames"],
            "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True,
            "scaling": P["scaling"],
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": f"scenes/{args.scene}.scene {res_h}x{res_v}x{spp}spp ({primary}), depth 8, synthetic stand-in mesh, seeded jitter",
                       "parallelism": f"image-tiled, {world} rank(s) x interleaved {tiling.BAND_ROWS}-row bands, no collective on the data path"},
            "value_counts": "rays traced on the device: generated primaries + shadow + reflection rays (primaries of 64-pixel blocks proven empty are not generated and not counted)",
            "value_reference_equivalent": round(P["rays_ref_equiv_frame"] / ms_per_step / 1e3, 3),
            "kernel_ms_per_step": round(P["kernel_ms_max"] / steps, 4),
            "value_kernel_only": round(P["rays_traced_frame"] / (P["kernel_ms_max"] / steps) / 1e3, 3),
            "boundary_ms_per_frame": {**boundary, "note": "blocking ft_render with a caller host buffer (Program.fs:63-68), PCIe copy included; steady state over 10 frames into one reused array; *_pinned: the buffer comes from ft_host_alloc"},
            "boundary_streaming_ms_per_frame": {**streaming, "note": "frames queued with ft_render_enqueue_into: each frame's copy into page-locked host memory rides beside the next frame's tracing; 30 frames, PCIe included"},
            "frame_ms_incl_copy": boundary.get("f64"),
            "d2h_copy_ms": round(copy_ms, 3),
            "gather_ms": round(gather_ms, 3),
            "roofline": roof,
            "per_kernel_ms_per_step": {k: round(v / steps, 4) for k, v in k_times.items()},
            "per_kernel_note": "HIP-event brackets on the frame's main stream; queued frames overlap (the next frames' k_classify and k_primary and the last frame's k_resolve run on other streams beside k_primary), "
                               "so `other` is what of a frame's own span its tracing kernels do not cover, not time added to the frame period: ms_per_step minus the dominant kernel is that",
            "frame_period_minus_dominant_kernel_ms": round(ms_per_step - dom_avg_ms * launches_per_step, 4),
            "per_kernel_launches_per_step": {k: round(v / steps, 3) for k, v in k_launch.items()},
            "layout_bytes_per_frame": {k[len("algorithmic_bytes_"):]: int(v) for k, v in st.items() if k.startswith("algorithmic_bytes_")},
            "rays_per_frame": {"traced_all_ranks": int(P["rays_traced_frame"]), "reference_equivalent_all_ranks": int(P["rays_ref_equiv_frame"]),
                               "primary_listed_all_ranks": int(P["rays_listed_frame"]), "primary_never_generated_all_ranks": int(P["rays_culled_frame"]),
                               "shadow_rank0": int(st["rays_shadow"]), "reflect_rank0": int(st["rays_reflect"]),
                               "shadow_primary_rank0": int(st["rays_shadow_primary"]), "reflect_primary_rank0": int(st["rays_reflect_primary"])},
            "workloads": {k: summary(r) for k, r in results.items()},
        }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"], out["parity"] = cpu_baseline(scene, res_h, res_v, spp, P["jitter"], args.cpu_baseline_seconds, full)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    ctx.close()


def effective_cpus():
    """Host cores this process may really use: its affinity mask, capped by the cgroup's CPU quota (a GPU box of this pool shows 256
    logical CPUs and grants 16 CPUs' worth of time; 256 threads on it are 16 cores' work and were reported as 256 in round 2)."""
    import math
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()
        if quota != "max":
            n = min(n, max(1, int(math.ceil(int(quota) / int(period)))))
    except (OSError, ValueError):
        pass
    return max(1, n)


def cpu_baseline(scene, res_h, res_v, spp, jitter, budget_s, gpu_frame):
    """The CPU oracle (a C++ port of the F# algorithm; the F# toolchain does not exist here) timed on this box's host cores on
    a bounded sample: interleaved 8-row bands of the same frame.  The bands it renders double as the parity check of the GPU
    frame the bench has just timed (the oracle is the checker here, never the thing measured as the product)."""
    import numpy as np

    from functracer_amd import tiling
    from oracle import ft_oracle_py as O
    orc = O.Oracle()
    scene.lower(orc)
    cores = effective_cpus()
    probe = tiling.bands_for_rank(res_h, res_v, 0, 64)             # calibrate on a thin sample, then size the real one to ~budget_s
    frame = np.zeros((res_v, res_h, 3))
    _, st = orc.render(scene.camera, res_h, res_v, spp, jitter, tiles=probe, threads=cores, out=frame)
    rate = st["rays_traced"] / (st["wall_ms"] * 1e-3)
    frac = min(1.0, max(1.0 / 64, budget_s * rate / (st["rays_traced"] * 64)))
    stride = max(1, int(round(1.0 / frac)))
    sample = tiling.bands_for_rank(res_h, res_v, 0, stride)
    _, st = orc.render(scene.camera, res_h, res_v, spp, jitter, tiles=sample, threads=cores, out=frame)
    base = {"value": round(st["rays_traced"] / (st["wall_ms"] * 1e-3) / 1e6, 4), "unit": "Mrays/s", "cores": int(st["threads"]), "logical_cpus_visible": os.cpu_count(), "kind": "port",
            "counts": "reference-equivalent rays (the oracle traces every ray of the F# recursion): compare with value_reference_equivalent",
            "sample": f"every {stride}th 8-row band of the same {res_h}x{res_v}x{spp}spp frame ({st['rays_traced']} rays, {st['wall_ms'] / 1e3:.1f} s), "
                      "C++ restatement of the F# algorithm (oracle/ft_oracle.cpp); the F# toolchain is unavailable"}
    want = tiling.pack_bands(frame, sample)
    got = tiling.pack_bands(gpu_frame, sample)
    err = np.abs(got - want) / np.maximum(np.abs(want), 1e-3)      # the contract's relative error (tests/helpers.py: pixel_errors)
    parity = {"max_rel_err": float(err.max()), "pixels_outside_1e-4": int((err > 1e-4).any(axis=-1).sum()), "pixels_compared": int(err.shape[0] * err.shape[1]),
              "pixels_bit_identical": int((got == want).all(axis=-1).sum()), "against": "oracle/ft_oracle.cpp on the cpu_baseline band sample of the timed frame",
              "tolerance": "1e-4 relative per channel (floor 1e-3), BASELINE.json north_star"}
    return base, parity


if __name__ == "__main__":
    main()
