"""Multi-GPU sharding of the image: one process per GPU, interleaved row strips, one exchange
step - the per-rank bands are gathered TO THE ROOT with torch.distributed (backend "nccl" = RCCL over xGMI: every peer
sends over its own link to rank 0; the ring all-gather, 8 x the bytes at 8 ranks, stays selectable) and scattered
back into frame order by kernel K4 (mirt_deinterleave_device).  The scene is replicated on every rank; the
counter-based RNG makes the image independent of the number of ranks.

The bookkeeping is plain integer arithmetic and is tested on CPU (gloo, world_size 2) with synthetic bands.
"""
from __future__ import annotations

import numpy as np

DEFAULT_STRIP_ROWS = 8


def owned_rows(height: int, strip_rows: int, part_index: int, part_count: int) -> np.ndarray:
    """Global row numbers rendered by `part_index`, in band (local) order."""
    s = strip_rows or DEFAULT_STRIP_ROWS
    rows = np.arange(height)
    return rows[(rows // s) % max(part_count, 1) == part_index]


def part_rows(height: int, strip_rows: int, part_index: int, part_count: int) -> int:
    return int(len(owned_rows(height, strip_rows, part_index, part_count)))


def deinterleave_host(bands: np.ndarray, height: int, strip_rows: int) -> np.ndarray:
    """Host restatement of K4 for tests: bands[r, local_row] -> frame[global_row]."""
    n = bands.shape[0]
    frame = np.empty((height,) + bands.shape[2:], bands.dtype)
    for r in range(n):
        rows = owned_rows(height, strip_rows, r, n)
        frame[rows] = bands[r, : len(rows)]
    return frame


def gather_bands(band, world_size: int):
    """All ranks contribute an equally sized band tensor (rank 0's size is the largest: pad to it); returns the
    stacked (world_size, ...) tensor on every rank.  On ROCm the 'nccl' backend is RCCL."""
    import torch
    import torch.distributed as td

    if world_size == 1:
        return band.unsqueeze(0)
    # concatenated-along-dim-0 output: the one layout both RCCL and gloo accept
    out = torch.empty((world_size * band.shape[0],) + tuple(band.shape[1:]), dtype=band.dtype, device=band.device)
    td.all_gather_into_tensor(out, band.contiguous())
    return out.view((world_size,) + tuple(band.shape))


def gather_bands_into(out, band) -> None:
    """The same exchange into a caller-owned (world_size * rows, ...) tensor, on torch's current stream: what bench.py
    calls once per frame and ring slot, so that nothing is allocated while frames are in flight."""
    import torch.distributed as td

    if out.shape[0] % band.shape[0] != 0 or tuple(out.shape[1:]) != tuple(band.shape[1:]) or out.dtype != band.dtype:
        raise ValueError(f"gather buffer {tuple(out.shape)} {out.dtype} does not hold whole bands of {tuple(band.shape)} {band.dtype}")
    td.all_gather_into_tensor(out, band)


def gather_bands_to_root_into(out, band, rank: int, world_size: int, root: int = 0) -> None:
    """Gather-to-root: every rank sends its band to `root`, which receives band r in rows [r * rows, (r + 1) * rows) of its
    caller-owned `out` (None on the other ranks).  xGMI is point to point, so each peer's band travels over its own link to
    the root - (N - 1) bands arrive in parallel - where the ring all-gather moves N - 1 bands over every link and leaves N
    copies of a frame only rank 0 uses (1080p f32 at 8 ranks: 3.1 MB per rank against 24.9 MB).  torch.distributed.gather:
    grouped ncclSend / ncclRecv on the nccl (= RCCL) backend, on torch's current stream; plain gather on gloo."""
    import torch.distributed as td

    if rank == root:
        if out is None or out.shape[0] != world_size * band.shape[0] or tuple(out.shape[1:]) != tuple(band.shape[1:]) or out.dtype != band.dtype:
            raise ValueError(f"gather buffer {None if out is None else tuple(out.shape)} does not hold {world_size} bands of {tuple(band.shape)} {band.dtype}")
        slots = list(out.view((world_size,) + tuple(band.shape)).unbind(0))
        td.gather(band, gather_list=slots, dst=root)
    else:
        td.gather(band, gather_list=None, dst=root)
