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
