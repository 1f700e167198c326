"""Row-band partition of one frame over the GPUs of a node, and its re-assembly (index arithmetic).

The reference parallelises Camera::render by image row (camera.rs:183-205); here a frame is cut
into bands of BAND_ROWS pixel rows, band b goes to rank b mod world (interleaved, because cost per
row is very uneven: glass regions are ~60x floor regions), every rank renders its bands packed one
after another (rt_options tile_rows/tile_first/tile_stride) and the packed buffers are gathered to
rank 0.  There is no other exchange step: no all-reduce, no halo.
"""
from __future__ import annotations

import numpy as np

from .capi import RtOptions, band_rows

BAND_ROWS = 8


def rank_options(base: RtOptions, rank: int, world: int) -> RtOptions:
    o = RtOptions()
    o.max_depth, o.jitter_mode, o.seed, o.flags = base.max_depth, base.jitter_mode, base.seed, base.flags
    if world > 1:
        o.tile_rows, o.tile_first, o.tile_stride = BAND_ROWS, rank, world
    return o


def padded_rows(height: int, world: int) -> int:
    """Rows of the largest per-rank buffer (rank 0 always has the most bands)."""
    return height if world == 1 else len(band_rows(height, BAND_ROWS, 0, world))


def gather_permutation(height: int, world: int) -> np.ndarray:
    """perm[y] = row index of global image row y inside concat(rank buffers, each padded_rows long)."""
    pad = padded_rows(height, world)
    perm = np.full(height, -1, dtype=np.int64)
    for r in range(world):
        rows = band_rows(height, BAND_ROWS, r, world) if world > 1 else np.arange(height)
        perm[rows] = r * pad + np.arange(len(rows))
    assert (perm >= 0).all()
    return perm


def band_copy_plan(width: int, rank: int, world: int, my_rows: int, bytes_per_pixel: int = 12):
    """Strided copies that move one rank's PACKED bands to their rows of a full frame.

    Returns a list of (dst_offset, dst_pitch, src_offset, src_pitch, width_bytes, height) in the sense
    of hipMemcpy2D: `height` runs of `width_bytes` bytes, `src_pitch` apart in the packed buffer and
    `dst_pitch` apart in the frame.  Full bands go in one copy (they lie `world` bands apart in the
    frame); a last partial band (image height not a multiple of BAND_ROWS) in a second one.
    """
    band_bytes = BAND_ROWS * width * bytes_per_pixel
    full, tail = divmod(my_rows, BAND_ROWS)
    plan = []
    dst0 = rank * band_bytes
    if full:
        plan.append((dst0, world * band_bytes, 0, band_bytes, band_bytes, full))
    if tail:
        n = tail * width * bytes_per_pixel
        plan.append((dst0 + full * world * band_bytes, n, full * band_bytes, n, n, 1))
    return plan
