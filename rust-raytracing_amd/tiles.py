"""Framebuffer partition across the GPUs of one node and the single gather that reassembles it.

The path shards by construction (pixels are independent, scene.rs:149-160): rank r of n renders the
interleaved row band {y : y % n == r} -- interleaving balances the very uneven per-pixel cost -- into a
compact local buffer; there is no mid-render exchange.  One `torch.distributed.gather` (RCCL over xGMI
on GPUs, gloo on CPU tensors in the tests) brings the bands to rank 0, which de-interleaves them.
Pixel values do not depend on the partition: a pixel's RNG key is its index in the full image.
"""
import torch
import torch.distributed as dist


def rows_for_rank(height, rank, world):
    """(row_begin, row_stride, n_rows) of rank's band."""
    if not 0 <= rank < world:
        raise ValueError("rank %d outside world %d" % (rank, world))
    n_rows = (height - rank + world - 1) // world if rank < height else 0
    return rank, world, n_rows


def band_capacity(height, world):
    """Rows every rank's gather buffer holds (bands are padded to equal size for the collective)."""
    return (height + world - 1) // world


def alloc_band(height, width, world, device, dtype=torch.float64):
    return torch.zeros((band_capacity(height, world), width, 3), dtype=dtype, device=device)


def gather_bands(band, height, width, rank, world, dst=0, group=None):
    """One gather to `dst`; returns the full (height, width, 3) image there, None elsewhere."""
    if world == 1:
        return band[:height]
    cap = band_capacity(height, world)
    assert band.shape[0] == cap and band.shape[1] == width
    if rank == dst:
        parts = [torch.empty_like(band) for _ in range(world)]
        dist.gather(band, gather_list=parts, dst=dst, group=group)
        return deinterleave(parts, height, width)
    dist.gather(band, gather_list=None, dst=dst, group=group)
    return None


def deinterleave(parts, height, width):
    """parts[r][k] is image row r + k*world."""
    world = len(parts)
    full = torch.empty((height, width, 3), dtype=parts[0].dtype, device=parts[0].device)
    for r, p in enumerate(parts):
        _, _, n = rows_for_rank(height, r, world)
        full[r::world] = p[:n]
    return full
