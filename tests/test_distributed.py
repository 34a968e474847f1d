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


# ---------------------------------------------------------------- BandDecoder (bench.py --workload 16k-bands) over gloo, CPU only
class _StubDecoder:
    """Stands in for api.Decoder on a box without a GPU: 'decodes' a band by copying the rows of a known image."""

    def __init__(self, full):
        self.full = full
        self.opts = {}

    def set_option(self, name, value):
        self.opts[name] = value
        return 1

    def decode_batch(self, files, outs, dev_in=None, synchronize=True):
        import ctypes
        r0, n = self.opts["band_first_row"], self.opts["band_rows"]
        h = self.full.shape[0]
        y0, y1 = min(r0 * 256, h), min((r0 + n) * 256, h)
        rows = np.ascontiguousarray(self.full[y0:y1])
        ctypes.memmove(outs[0], rows.ctypes.data, rows.nbytes)
        return [0]


def _band_worker(rank, world, port, q, data, shape):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from pdn_jpegxl_amd.distributed import BandDecoder
    full = np.random.default_rng(9).integers(0, 256, shape, dtype=np.uint8)
    bd = BandDecoder(_StubDecoder(full), data, rank, world, gather_device="cpu", device="cpu")
    bd.step()
    bd.step()   # buffers are reused between steps
    if rank == 0:
        img = bd.image().numpy().reshape(shape)
        q.put(bool((img == full).all()) and bd.rows == bd.rows_per_rank[0])
    dist.destroy_process_group()


def _spawn(target, world, extra):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=target, args=(r, world, port, q) + extra) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    return q.get(timeout=5)


def test_band_decoder_gathers_uneven_bands_gloo():
    """A frame of 3 group rows (the last one short) over 2 ranks: 2 + 1 group rows, padded all_gather, exact reassembly."""
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    # any real file gives BandDecoder its geometry through jxlhip_peek (host only); the pixels come from the stub
    from pdn_jpegxl_amd import api
    import oracle_lib as O
    from pdn_jpegxl_amd.synth import synth
    data = O.encode(synth(300, 600, 31), distance=2.0)
    info = api.peek(data)
    assert (info.height + 255) // 256 == 3
    assert _spawn(_band_worker, 2, (data, (info.height, info.width, info.num_channels))) is True


def test_bench_gpus_flag_starts_ranks_or_fails_loudly():
    """`bench.py --gpus 2` must never silently run one rank: without two visible GPUs it exits non-zero before touching a device."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1"], env=env, capture_output=True, text=True, timeout=300)
    if torch.cuda.device_count() < 2:
        assert r.returncode == 2, (r.returncode, r.stderr[-400:])
        assert "only" in r.stderr and "GPU" in r.stderr
        assert '"n_gpus": 1' not in r.stdout
