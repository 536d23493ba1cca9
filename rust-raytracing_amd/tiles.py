"""Framebuffer partition across the GPUs of one node and the single gather that reassembles it.

The path shards by construction (pixels are independent, scene.rs:149-160): the frame is cut into blocks of
ROW_BLOCK = 8 rows dealt out round-robin -- rank r of n renders blocks r, r + n, ... (rtx_render_blocks) -- into a
compact local band; there is no mid-render exchange.  Interleaving balances the very uneven per-pixel cost; blocks of
8 rows keep the BVH kernels' 8x8 ray tiles whole (with single interleaved rows a tile of rank r is 8 columns x 8n image
rows, and the packet walks of the primary rays lose a quarter to a third of their rate).  One
`torch.distributed.gather` (RCCL over xGMI on GPUs, gloo on CPU tensors in the tests) brings the bands to rank 0, which
puts the rows in place with one index_select.  Pixel values do not depend on the partition: a pixel's RNG key is its
index in the full image.
"""
import numpy as np
import torch
import torch.distributed as dist

ROW_BLOCK = 8


def rows_of_part(height, part, world, block=ROW_BLOCK):
    """Image rows of part `part` of `world`, in band order (numpy int64)."""
    if not 0 <= part < world:
        raise ValueError("part %d outside world %d" % (part, world))
    y = np.arange(height, dtype=np.int64)
    return y[(y // block) % world == part]


def rows_for_rank(height, rank, world):
    """(row_begin, row_stride, n_rows) of the single-row band (block = 1): rtx_render_rows' form."""
    if not 0 <= rank < world:
        raise ValueError("rank %d outside world %d" % (rank, world))
    n_rows = (height - rank + world - 1) // world if rank < height else 0
    return rank, world, n_rows


def band_capacity(height, world, block=1):
    """Rows every rank's gather buffer holds (bands are padded to equal size for the collective): part 0's count."""
    return len(rows_of_part(height, 0, world, block))


def alloc_band(height, width, world, device, dtype=torch.float64, block=1):
    return torch.zeros((band_capacity(height, world, block), width, 3), dtype=dtype, device=device)


class Partition:
    """Rank `rank` of `world`'s share of a `height`-row frame and the gather that reassembles the frame on rank `dst`.

    Everything the timed loop needs is allocated here, once: the band, rank dst's receive buffer (one tensor, the gather
    list are views of it) and the row permutation, so that a step is render + one collective + one index_select."""

    def __init__(self, height, rank, world, block=ROW_BLOCK):
        self.height, self.rank, self.world, self.block = int(height), int(rank), int(world), int(block)
        self.rows = rows_of_part(self.height, self.rank, self.world, self.block)
        self.n_rows = len(self.rows)
        self.cap_rows = band_capacity(self.height, self.world, self.block)
        self._recv = None
        self._perm = None

    def alloc_band(self, width, device, dtype=torch.float64):
        return torch.zeros((self.cap_rows, int(width), 3), dtype=dtype, device=device)

    def render(self, handle, width, band, stream=None, want_stats=False):
        """The band through the C ABI (rtx_render_blocks); asynchronous unless stats are asked for."""
        return handle.render_blocks(width, self.height, self.block, self.rank, self.world, band.data_ptr(), stream=stream,
                                    want_stats=want_stats)

    def row_permutation(self, device):
        """perm[y] = index of image row y in the stacked (world * cap_rows) receive buffer."""
        perm = np.empty(self.height, dtype=np.int64)
        for p in range(self.world):
            rows = rows_of_part(self.height, p, self.world, self.block)
            perm[rows] = p * self.cap_rows + np.arange(len(rows))
        return torch.from_numpy(perm).to(device)

    def gather(self, band, dst=0, group=None):
        """One gather to `dst`; returns the full (height, width, 3) image there, None elsewhere."""
        if self.world == 1:
            return band[:self.height]
        assert band.shape[0] == self.cap_rows
        if self.rank != dst:
            dist.gather(band, gather_list=None, dst=dst, group=group)
            return None
        if self._recv is None or self._recv.shape[2:] != band.shape[1:] or self._recv.dtype != band.dtype or self._recv.device != band.device:
            self._recv = torch.empty((self.world,) + tuple(band.shape), dtype=band.dtype, device=band.device)
            self._perm = self.row_permutation(band.device)
        dist.gather(band, gather_list=list(self._recv.unbind(0)), dst=dst, group=group)
        return self._recv.view((self.world * self.cap_rows,) + tuple(band.shape[1:])).index_select(0, self._perm)


def gather_bands(band, height, width, rank, world, dst=0, group=None, block=1):
    """One-off form of Partition.gather (allocates its receive buffer per call: tests, not the timed loop)."""
    assert band.shape[1] == width
    return Partition(height, rank, world, block).gather(band, dst=dst, group=group)


def deinterleave(parts, height, width, block=1):
    """parts[p] is band p (rows_of_part order); returns the (height, width, 3) frame."""
    world = len(parts)
    full = torch.empty((height, width, 3), dtype=parts[0].dtype, device=parts[0].device)
    for p, band in enumerate(parts):
        rows = rows_of_part(height, p, world, block)
        full[torch.from_numpy(rows).to(full.device)] = band[:len(rows)]
    return full
