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
