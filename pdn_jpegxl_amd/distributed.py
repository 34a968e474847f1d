"""Multi-GPU layout of the decode path (one process per GPU, torch.distributed over RCCL/xGMI).

JPEG XL groups are independently decodable once the LF image exists, so a large frame is sharded by contiguous
group rows ("bands") and independent images are sharded round-robin.  No collective is on the data path of the
decode itself; the only exchange is the final gather of the RGBA8 bands / per-rank outputs.
"""
import torch
import torch.distributed as dist


def shard_images(n_images, rank, world):
    """Indices of the images this rank decodes (round-robin keeps per-rank work within one image of equal)."""
    return list(range(rank, n_images, world))


def band_rows(n_group_rows, rank, world):
    """[r0, r1) group rows owned by `rank`: contiguous, sizes differ by at most one, earlier ranks get the extra row."""
    base, extra = divmod(n_group_rows, world)
    r0 = rank * base + min(rank, extra)
    return r0, r0 + base + (1 if rank < extra else 0)


def gather_bands(band, rows_per_rank, row_bytes, dst=0):
    """Gathers variable-height bands (uint8 tensors of rows*row_bytes) into one image tensor on `dst`.
    Uses all_gather on a padded buffer (RCCL all-gather over xGMI on GPU, gloo on CPU)."""
    world = dist.get_world_size()
    rank = dist.get_rank()
    max_rows = max(rows_per_rank)
    buf = torch.zeros(max_rows * row_bytes, dtype=torch.uint8, device=band.device)
    buf[: band.numel()] = band.reshape(-1)
    parts = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(parts, buf)
    if rank != dst:
        return None
    return torch.cat([p[: r * row_bytes] for p, r in zip(parts, rows_per_rank)])


def decode_frame_band(dec, data, rank, world, device="cuda"):
    """Band-sharded decode of ONE large lossy frame: this rank decodes the group rows band_rows() assigns to it (plus one halo
    row each side, redundantly, so the loop filters need no exchange) and returns (uint8 tensor of its pixel rows, (y0, y1)).
    `dec` is a pdn_jpegxl_amd.api.Decoder bound to this rank's GPU."""
    from . import api
    info = api.peek(data)
    n_rows = (info.height + 255) // 256
    r0, r1 = band_rows(n_rows, rank, world)
    y0, y1 = min(r0 * 256, info.height), min(r1 * 256, info.height)
    row_bytes = info.width * info.num_channels * info.bytes_per_sample   # u16 samples for streams of more than 8 bits
    out = torch.empty(max(1, (y1 - y0) * row_bytes), dtype=torch.uint8, device=device)
    if r1 > r0:
        dec.set_option("band_first_row", r0)
        dec.set_option("band_rows", r1 - r0)
        try:
            st = dec.decode_batch([data], [out.data_ptr()], None, synchronize=True)
        finally:
            dec.set_option("band_rows", 0)
            dec.set_option("band_first_row", 0)
        if st[0] != 0:
            raise RuntimeError("band decode failed with status %d" % st[0])
    return out[: (y1 - y0) * row_bytes], (y0, y1)


def decode_frame_sharded(dec, data, dst=0):
    """All ranks decode their band of `data`; the bands are gathered (RCCL all-gather over xGMI) and rank `dst` gets the image."""
    from . import api
    rank, world = dist.get_rank(), dist.get_world_size()
    info = api.peek(data)
    band, _ = decode_frame_band(dec, data, rank, world)
    n_rows = (info.height + 255) // 256
    spans = [band_rows(n_rows, r, world) for r in range(world)]
    rows = [min(b * 256, info.height) - min(a * 256, info.height) for a, b in spans]
    img = gather_bands(band, rows, info.width * info.num_channels * info.bytes_per_sample, dst=dst)
    if img is None:
        return None
    if info.bytes_per_sample == 2:
        img = img.view(torch.float16 if info.reserved else torch.int16)   # torch has no uint16 arithmetic; callers reinterpret
    elif info.bytes_per_sample == 4:
        img = img.view(torch.float32)
    return img.reshape(info.height, info.width, info.num_channels)


class BandDecoder:
    """One rank's share of a band-sharded frame, set up once and stepped many times (bench.py --workload 16k-bands): the file goes
    to HBM once, the band and gather buffers are allocated once, and a step is decode(band) + one all_gather.
    gather_device "cuda" = RCCL over xGMI; "cpu" = gloo (one-GPU rehearsals and the CPU tests)."""

    def __init__(self, dec, data, rank, world, gather_device="cuda", device="cuda"):
        from . import api
        self.dec, self.data, self.rank, self.world = dec, data, rank, world
        self.info = info = api.peek(data)
        n_rows = (info.height + 255) // 256
        self.spans = [band_rows(n_rows, r, world) for r in range(world)]
        self.row_bytes = info.width * info.num_channels * info.bytes_per_sample
        self.rows_per_rank = [min(b * 256, info.height) - min(a * 256, info.height) for a, b in self.spans]
        self.r0, self.r1 = self.spans[rank]
        self.rows = self.rows_per_rank[rank]
        self.max_bytes = max(1, max(self.rows_per_rank) * self.row_bytes)
        self.src = torch.zeros(len(data) + 64, dtype=torch.uint8, device=device)
        self.src[: len(data)] = torch.frombuffer(bytearray(data), dtype=torch.uint8).to(device)
        self.band = torch.zeros(self.max_bytes, dtype=torch.uint8, device=device)   # padded to the tallest band: equal all_gather shares
        self.gather_device = gather_device
        self.gathered = torch.empty(world * self.max_bytes, dtype=torch.uint8, device=gather_device) if world > 1 else None

    def decode(self):
        if self.r1 <= self.r0:
            return
        self.dec.set_option("band_first_row", self.r0)
        self.dec.set_option("band_rows", self.r1 - self.r0)
        try:
            st = self.dec.decode_batch([self.data], [self.band.data_ptr()], [self.src.data_ptr()], synchronize=True)
        finally:
            self.dec.set_option("band_rows", 0)
            self.dec.set_option("band_first_row", 0)
        if st[0] != 0:
            raise RuntimeError("band decode failed with status %d" % st[0])

    def gather(self):
        if self.world == 1:
            return
        share = self.band if self.gather_device == self.band.device.type else self.band.to(self.gather_device)
        dist.all_gather_into_tensor(self.gathered, share)

    def step(self):
        self.decode()
        self.gather()

    def image(self):
        """The whole frame as an (H, W, C) uint8 tensor (u8 streams) assembled from the gathered bands; call after step()."""
        info = self.info
        if self.world == 1:
            flat = self.band[: self.rows * self.row_bytes]
        else:
            flat = torch.cat([self.gathered[r * self.max_bytes: r * self.max_bytes + n * self.row_bytes] for r, n in enumerate(self.rows_per_rank)])
        return flat.reshape(info.height, info.width * info.num_channels * info.bytes_per_sample)
