"""The N > 1 path on CPU: two `gloo` ranks partition one frame into interleaved 8-row bands
(raytracer_amd/partition.py — the code bench.py runs over RCCL), each renders ITS bands, rank 0
gathers and re-interleaves, and the result must equal the single-rank frame bit for bit.

No GPU here, so the per-band renderer is the test-only FP32 host simulator (same rt_options band
semantics as rt_render_device; the `-m gpu` suite checks the real kernel's bands against the full
frame)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, name, w, h, out_path):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import helpers as H
    from raytracer_amd import ExampleScene, RtOptions, partition

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        base = RtOptions()
        base.jitter_mode, base.seed = 1, 5  # the jitter hash is keyed on the GLOBAL pixel index
        ex = ExampleScene(name, w, h)
        opt = partition.rank_options(base, rank, world)
        pad = partition.padded_rows(h, world)
        mine = H.hostsim_render(ex, opt)
        local = torch.zeros((pad, w, 3), dtype=torch.float32)
        local[: mine.shape[0]] = torch.from_numpy(mine)
        gathered = [torch.empty_like(local) for _ in range(world)] if rank == 0 else None
        dist.gather(local, gathered, dst=0)
        if rank == 0:
            perm = torch.from_numpy(partition.gather_permutation(h, world))
            final = torch.index_select(torch.cat(gathered, 0), 0, perm)
            np.save(out_path, final.numpy())
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("h", [54, 61])  # 61: ragged last band, ranks get different row counts
def test_two_ranks_reassemble_the_single_rank_frame(tmp_path, h):
    name, w, world = "checkered_walls_metallic_sphere", 96, 2
    out = str(tmp_path / "final.npy")
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(world, port, name, w, h, out), nprocs=world, join=True)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import helpers as H
    from raytracer_amd import ExampleScene, RtOptions

    base = RtOptions()
    base.jitter_mode, base.seed = 1, 5
    full = H.hostsim_render(ExampleScene(name, w, h), base)
    assert np.array_equal(np.load(out), full)


def test_partition_index_arithmetic():
    from raytracer_amd import band_rows, partition

    for h in (1, 7, 8, 9, 720, 1080, 2160):
        for world in (1, 2, 3, 4, 8):
            perm = partition.gather_permutation(h, world)
            pad = partition.padded_rows(h, world)
            assert sorted(perm.tolist()) == sorted(set(perm.tolist())) and len(perm) == h
            assert perm.max() < world * pad
            seen = np.concatenate([band_rows(h, partition.BAND_ROWS, r, world) for r in range(world)]) if world > 1 \
                else np.arange(h)
            assert sorted(seen.tolist()) == list(range(h))


@pytest.mark.parametrize("height,world", [(1080, 2), (1084, 2), (1084, 3), (540, 4), (37, 8), (8, 3), (5, 2)])
def test_band_copy_plan_puts_every_row_where_the_gather_puts_it(height, world):
    """bench.py's other way to assemble the frame: each rank moves its packed bands into the shared frame
    with strided copies (hipMemcpy2D semantics, emulated here on byte arrays).  The frame must equal the
    one the gather + permutation path builds, for heights with and without a partial last band."""
    sys.path.insert(0, ROOT)
    from raytracer_amd import partition
    from raytracer_amd.capi import band_rows

    width, bpp = 5, 12
    rng = np.random.default_rng(height * 31 + world)
    full = rng.integers(0, 255, size=(height, width * bpp), dtype=np.uint8)  # the frame, one row per image row
    frame = np.zeros(height * width * bpp, dtype=np.uint8)
    for rank in range(world):
        rows = band_rows(height, partition.BAND_ROWS, rank, world)
        packed = full[rows].reshape(-1)  # what this rank's kernel leaves in its own buffer
        for dst, dpitch, src, spitch, wbytes, n in partition.band_copy_plan(width, rank, world, len(rows), bpp):
            for k in range(n):
                frame[dst + k * dpitch: dst + k * dpitch + wbytes] = packed[src + k * spitch: src + k * spitch + wbytes]
    assert np.array_equal(frame.reshape(height, width * bpp), full)
