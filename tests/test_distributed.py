"""world_size-2 gloo test of the sharding/gather layout used by the multi-GPU path (CPU only)."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from pdn_jpegxl_amd.distributed import band_rows, gather_bands, shard_images


def test_band_rows_cover_exactly():
    for rows in (1, 9, 64, 65):
        for world in (1, 2, 4, 8):
            spans = [band_rows(rows, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == rows
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def test_shard_images_partition():
    for n in (1, 7, 64):
        for world in (1, 2, 8):
            got = sorted(i for r in range(world) for i in shard_images(n, r, world))
            assert got == list(range(n))


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    H, W, C = 9 * 16, 40, 4
    full = torch.from_numpy(np.random.default_rng(3).integers(0, 256, (H, W, C), dtype=np.uint8))
    spans = [band_rows(9, r, world) for r in range(world)]
    r0, r1 = spans[rank]
    band = full[r0 * 16:r1 * 16].contiguous()
    out = gather_bands(band, [(b - a) * 16 for a, b in spans], W * C, dst=0)
    if rank == 0:
        q.put(bool((out.reshape(H, W, C) == full).all()))
    dist.destroy_process_group()


def test_gather_bands_gloo_world2():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert q.get(timeout=5) is True
