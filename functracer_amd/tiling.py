"""Image-tile partitioning of one frame over N ranks (one process per GPU).

Rays are independent (Shading.fs:141-147), so the frame shards with no exchange step: every rank
holds the whole scene, renders an interleaved set of row bands (cost varies strongly over the
image, so bands are dealt round-robin) and the bands are gathered on the host.  No RCCL collective
is involved; the gather below uses whatever process group it is given (gloo on CPU tensors).
"""
import numpy as np

BAND_ROWS = 8


def bands_for_rank(res_h, res_v, rank, world, band_rows=BAND_ROWS):
    """Rects (x0, y0, w, h) of the row bands owned by `rank`: band b belongs to rank b % world."""
    bands = [(0, y, res_h, min(band_rows, res_v - y)) for y in range(0, res_v, band_rows)]
    return bands[rank::world]


def pack_bands(frame, bands):
    """Rows of `frame` (res_v, res_h, 3) covered by `bands`, concatenated in band order."""
    if not bands:
        return np.zeros((0,) + frame.shape[1:], dtype=frame.dtype)
    return np.concatenate([frame[y:y + h] for (_, y, _, h) in bands], axis=0)


def unpack_bands(frame, bands, rows):
    k = 0
    for (_, y, _, h) in bands:
        frame[y:y + h] = rows[k:k + h]
        k += h
    return frame


def gather_frame(local_frame, res_h, res_v, rank, world, group=None, band_rows=BAND_ROWS):
    """Assemble the full frame on rank 0 from every rank's bands (host-side gather).  Returns the
    frame on rank 0 and None elsewhere."""
    if world == 1:
        return local_frame
    import torch
    import torch.distributed as dist
    rows_of = [sum(h for (_, _, _, h) in bands_for_rank(res_h, res_v, r, world, band_rows)) for r in range(world)]
    max_rows = max(rows_of)                      # gather needs equal shapes: pad every rank's rows to the largest share
    mine = np.zeros((max_rows, res_h, 3), dtype=local_frame.dtype)
    mine[:rows_of[rank]] = pack_bands(local_frame, bands_for_rank(res_h, res_v, rank, world, band_rows))
    mine = torch.from_numpy(mine)
    if rank == 0:
        parts = [torch.empty_like(mine) for _ in range(world)]
        dist.gather(mine, gather_list=parts, dst=0, group=group)
        frame = np.zeros((res_v, res_h, 3), dtype=local_frame.dtype)
        for r in range(world):
            unpack_bands(frame, bands_for_rank(res_h, res_v, r, world, band_rows), parts[r].numpy()[:rows_of[r]])
        return frame
    dist.gather(mine, gather_list=None, dst=0, group=group)
    return None
