"""The N>1 path on CPU: world_size-2 (and 3) gloo process groups run the product's band partition + single
gather (rust-raytracing_amd/tiles.py); each rank fills its band with the oracle (the checker stands in for
the GPU renderer here), rank 0 compares the reassembled image with a full oracle render."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, width, height, block, out_path):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import rtx_oracle as oracle
        from rust_raytracing_amd import scenes, tiles
        sc = oracle.make_scene(scenes.three_spheres(), scenes.CAMERA, rays_per_pixel=2, seed=42)
        part = tiles.Partition(height, rank, world, block)
        band = part.alloc_band(width, "cpu")
        for k, y in enumerate(part.rows):                      # one oracle row at a time: the band's rows in band order
            img = oracle.render(sc, width, height, n_threads=1, row_begin=int(y), row_stride=height)
            band[k] = torch.from_numpy(img[int(y)])
        full = part.gather(band, dst=0)
        again = part.gather(band, dst=0)                        # the pre-allocated receive buffer is reused
        if rank == 0:
            assert torch.equal(full, again)
        if rank == 0:
            np.save(out_path, full.numpy())
        else:
            assert full is None
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,width,height,block", [(2, 24, 15, 1), (2, 24, 21, 8), (3, 16, 20, 8), (2, 8, 1, 8), (3, 8, 10, 4)])
def test_band_partition_and_gather_gloo(tmp_path, oracle, world, width, height, block):
    from rust_raytracing_amd import scenes
    out = str(tmp_path / "full.npy")
    mp.spawn(_worker, args=(world, _free_port(), width, height, block, out), nprocs=world, join=True)
    got = np.load(out)
    ref = oracle.render(oracle.make_scene(scenes.three_spheres(), scenes.CAMERA, rays_per_pixel=2, seed=42), width, height)
    assert got.shape == ref.shape
    assert np.array_equal(got, ref)


def test_rows_for_rank_covers_every_row_once():
    from rust_raytracing_amd import tiles
    for height in (0, 1, 7, 8, 1080):
        for world in (1, 2, 3, 8):
            seen = []
            for r in range(world):
                rb, rs, n = tiles.rows_for_rank(height, r, world)
                assert n <= tiles.band_capacity(height, world)
                seen += [rb + k * rs for k in range(n)]
            assert sorted(seen) == list(range(height))


def test_block_partition_covers_every_row_once_and_matches_the_library(rtx):
    """tiles.rows_of_part (the Python side of the partition) against rtx_blocks_row_count (the C ABI's), every row in exactly
    one part, bands in increasing row order, whole blocks except the frame's last, and the load balance the docs promise."""
    from rust_raytracing_amd import tiles
    lib = rtx.load_library()
    for height in (0, 1, 7, 8, 9, 45, 1080, 2160):
        for world in (1, 2, 3, 8):
            for block in (1, 4, 8):
                seen = []
                for r in range(world):
                    rows = tiles.rows_of_part(height, r, world, block)
                    assert lib.rtx_blocks_row_count(height, block, r, world) == len(rows)
                    assert len(rows) <= tiles.band_capacity(height, world, block)
                    assert list(rows) == sorted(rows)
                    seen += list(rows)
                assert sorted(seen) == list(range(height))
    counts = [len(tiles.rows_of_part(1080, r, 8, 8)) for r in range(8)]
    assert max(counts) == 136 and min(counts) == 128                     # 17 or 16 blocks of 8 rows
    perm = tiles.Partition(45, 0, 3, 8).row_permutation("cpu")
    assert sorted(perm.tolist()) == sorted(set(perm.tolist())) and len(perm) == 45
