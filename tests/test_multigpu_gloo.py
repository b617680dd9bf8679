"""The N > 1 path on CPU: world_size-2 gloo processes each render their interleaved row bands and
rank 0 assembles the frame with functracer_amd.tiling.gather_frame.  Without a GPU the per-rank
renderer is the CPU oracle (tests may use it); on the GPU box bench.py runs the same partition and
gather around the HIP path."""
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
