"""The N>1 path on CPU: two gloo ranks each produce the band of rows they own (synthetic pixels standing in for
mirt_render_device's compact output), gather the bands with the same call bench.py uses, and scatter them into
frame order.  The result must equal the frame a single rank would produce.  No GPU, no oracle."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _frame(height, width):
    r = np.arange(height, dtype=np.float32)[:, None, None]
    c = np.arange(width, dtype=np.float32)[None, :, None]
    ch = np.arange(3, dtype=np.float32)[None, None, :]
    return r * 1000.0 + c + ch * 0.25


def _worker(rank, world, port, height, width, strip, q):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as td

    from raytracer_amd import dist

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    td.init_process_group("gloo", rank=rank, world_size=world)
    try:
        full = _frame(height, width)
        rows = dist.owned_rows(height, strip, rank, world)
        max_rows = dist.part_rows(height, strip, 0, world)  # rank 0 owns the most rows
        band = np.zeros((max_rows, width, 3), np.float32)
        band[: len(rows)] = full[rows]
        bands = dist.gather_bands(torch.from_numpy(band), world)
        frame = dist.deinterleave_host(bands.numpy(), height, strip)
        # bench.py's form: a ring of caller-owned gather buffers, one frame per slot (frames differ by a constant here)
        ring = [torch.zeros((world * max_rows, width, 3), dtype=torch.float32) for _ in range(3)]
        for f in range(5):
            dist.gather_bands_into(ring[f % 3], torch.from_numpy(band + np.float32(f)))
        for f in (2, 3, 4):
            got = dist.deinterleave_host(ring[f % 3].numpy().reshape(world, max_rows, width, 3), height, strip)
            assert np.array_equal(got, full + np.float32(f))
        with pytest.raises(ValueError):
            dist.gather_bands_into(torch.zeros((world * max_rows + 1, width, 3)), torch.from_numpy(band))
        # gather-to-root (what bench.py --gpus N does by default): only rank 0 holds a buffer, and it gets every band in its slot
        root_buf = torch.full((world * max_rows, width, 3), -1.0) if rank == 0 else None
        for f in range(3):
            dist.gather_bands_to_root_into(root_buf, torch.from_numpy(band + np.float32(10 * f)), rank, world)
            if rank == 0:
                got = dist.deinterleave_host(root_buf.numpy().reshape(world, max_rows, width, 3), height, strip)
                assert np.array_equal(got, full + np.float32(10 * f))
        if rank == 0:
            with pytest.raises(ValueError):
                dist.gather_bands_to_root_into(torch.zeros((world * max_rows + 2, width, 3)), torch.from_numpy(band), rank, world)
        # the timing reduction bench.py does: max over ranks
        t = torch.tensor([float(rank + 1)], dtype=torch.float64)
        td.all_reduce(t, op=td.ReduceOp.MAX)
        q.put((rank, bool(np.array_equal(frame, full)), float(t.item()), len(rows)))
    except Exception as e:  # report instead of leaving the parent to time out
        q.put((rank, False, repr(e), 0))
        raise
    finally:
        td.destroy_process_group()


@pytest.mark.parametrize("height,width,strip", [(36, 8, 8), (1080, 4, 8), (17, 5, 4), (3, 7, 8)])
def test_two_rank_gather_reassembles_frame(height, width, strip):
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, height, width, strip, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert [r[1] for r in res] == [True, True], res
    assert [r[2] for r in res] == [2.0, 2.0]
    assert sum(r[3] for r in res) == height


def test_band_bookkeeping_matches_the_library():
    """dist.part_rows / owned_rows (Python) against mirt_part_rows (the C ABI helper the kernels share)."""
    from raytracer_amd import abi, dist

    lib = abi.load()
    for height in (1, 7, 8, 9, 225, 1080, 2160):
        for strip in (0, 1, 4, 8, 16):
            for cnt in (1, 2, 3, 8):
                got = [lib.mirt_part_rows(height, strip or 8, i, cnt) for i in range(cnt)]
                want = [dist.part_rows(height, strip, i, cnt) for i in range(cnt)]
                assert got == want and sum(got) == height
                assert want[0] == max(want)  # gather_bands pads every band to rank 0's size
