"""The N > 1 path: world_size-2 gloo processes each render their interleaved row bands and rank 0
assembles the frame with functracer_amd.tiling.gather_frame.  Without a GPU the per-rank renderer is
the CPU oracle (tests may use it); the -m gpu tests run the same partition and gather around the HIP
path (two ranks sharing the box's one GPU) and bench.py --gpus 2 as the driver launches it."""
import os
import socket
import sys

import numpy as np
import pytest

from . import helpers as H

W, HH, SPP = 96, 50, 2


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, out_path):
    sys.path.insert(0, H.ROOT)
    import torch.distributed as dist

    import functracer_amd as ft
    from functracer_amd import tiling
    from oracle import ft_oracle_py as O
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    scene = ft.parse_scene_file(H.scene_path("night-house-det"))
    orc = O.Oracle()
    scene.lower(orc)
    jit = ft.jitter_pattern(SPP)
    bands = tiling.bands_for_rank(W, HH, rank, world, band_rows=4)
    local, _ = orc.render(scene.camera, W, HH, SPP, jit, tiles=bands, threads=1)
    frame = tiling.gather_frame(local, W, HH, rank, world, band_rows=4)
    if rank == 0:
        np.save(out_path, frame)
    dist.barrier()
    dist.destroy_process_group()


def test_band_partition_covers_the_frame_once():
    from functracer_amd import tiling
    for world in (1, 2, 3, 4, 8):
        seen = np.zeros((1080,), dtype=int)
        for r in range(world):
            for (x0, y, w, h) in tiling.bands_for_rank(1920, 1080, r, world):
                assert x0 == 0 and w == 1920
                seen[y:y + h] += 1
        assert (seen == 1).all()
        sizes = [sum(h for (_, _, _, h) in tiling.bands_for_rank(1920, 1080, r, world)) for r in range(world)]
        assert max(sizes) - min(sizes) <= tiling.BAND_ROWS


@pytest.mark.timeout(300)
def test_two_rank_tiled_frame_equals_single_process_frame(tmp_path):
    import torch.multiprocessing as mp

    import functracer_amd as ft
    from oracle import ft_oracle_py as O
    out = str(tmp_path / "frame.npy")
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    tiled = np.load(out)
    scene = ft.parse_scene_file(H.scene_path("night-house-det"))
    orc = O.Oracle()
    scene.lower(orc)
    full, _ = orc.render(scene.camera, W, HH, SPP, ft.jitter_pattern(SPP))
    assert np.array_equal(tiled, full)          # tiles are independent: bit-identical union


def _worker_hip(rank, world, port, out_path):
    sys.path.insert(0, H.ROOT)
    import torch.distributed as dist

    import functracer_amd as ft
    from functracer_amd import tiling
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    scene = ft.parse_scene_file(H.scene_path("night-house"))
    ctx = ft.Context(0)                                           # both ranks on the box's one GPU; on an 8-GPU node: LOCAL_RANK
    scene.lower(ctx)
    w, h, spp = 640, 360, 3
    jit = ft.jitter_pattern(spp)
    local = np.zeros((h, w, 3))
    ctx.render(scene.camera, w, h, spp, jit, tiles=tiling.bands_for_rank(w, h, rank, world), out=local)
    frame = tiling.gather_frame(local, w, h, rank, world)
    if rank == 0:
        np.save(out_path, frame)
    dist.barrier()
    dist.destroy_process_group()
    ctx.close()


@pytest.mark.gpu
@pytest.mark.timeout(600)
def test_two_rank_tiled_frame_on_the_hip_path(tmp_path, hip):
    """Each rank renders its bands through the C ABI on the device and the gathered frame equals the single-context frame bit
    for bit (seeded soft light included: the streams are keyed by pixel and sample, not by the tiling)."""
    import torch.multiprocessing as mp

    import functracer_amd as ft
    out = str(tmp_path / "frame.npy")
    mp.spawn(_worker_hip, args=(2, _free_port(), out), nprocs=2, join=True)
    scene = ft.parse_scene_file(H.scene_path("night-house"))
    scene.lower(hip)
    full, _ = hip.render(scene.camera, 640, 360, 3, ft.jitter_pattern(3))
    assert np.array_equal(np.load(out), full)


@pytest.mark.gpu
@pytest.mark.timeout(900)
def test_bench_two_ranks_as_the_driver_launches_it():
    """python -m torch.distributed.run --nproc-per-node 2 bench.py --gpus 2: one JSON line from rank 0, the timed workload is the
    strong-scaled config 5 frame, and every workload's ray count is the single-context count (the bands partition the frame)."""
    import json
    import subprocess
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(H.ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--side-steps", "2"]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=800, cwd=H.ROOT)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, res.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and "3840x2160x64spp" in d["config"]["workload"] and d["value"] > 0
    assert set(d["workloads"]) == {"config5", "headline", "weak"}
    one = subprocess.run([sys.executable, os.path.join(H.ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--side-steps", "1", "--no-cpu-baseline"],
                         capture_output=True, text=True, timeout=800, cwd=H.ROOT)
    assert one.returncode == 0, one.stderr[-2000:]
    d1 = json.loads([l for l in one.stdout.splitlines() if l.startswith("{")][0])
    for name in ("headline", "config5"):                            # same frame, same rays, whatever the number of ranks
        assert d["workloads"][name]["rays_traced_per_frame"] == d1["workloads"][name]["rays_traced_per_frame"], name
    assert d1["n_gpus"] == 1 and "1920x1080x16spp" in d1["config"]["workload"]
