"""Interleaved row bands: how one frame is split over the GPUs of a node.

Band b (rows [b*band_rows, (b+1)*band_rows)) belongs to rank b % world; a rank
stores its bands back to back in a compact band-major frame.  The same layout
is implemented on the device by pt_set_row_bands / pt_assemble_bands
(include/pt_hip.h); these helpers are the host-side statement of it.
"""
from __future__ import annotations

from typing import List, Tuple

import numpy as np

DEFAULT_BAND_ROWS = 16


def num_bands(height: int, band_rows: int) -> int:
    return (height + band_rows - 1) // band_rows


def rank_bands(height: int, band_rows: int, rank: int, world: int) -> List[Tuple[int, int]]:
    """Global row ranges [y0, y1) of the bands `rank` owns, in order."""
    return [(b * band_rows, min((b + 1) * band_rows, height)) for b in range(rank, num_bands(height, band_rows), world)]


def local_rows(height: int, band_rows: int, rank: int, world: int) -> int:
    return sum(y1 - y0 for y0, y1 in rank_bands(height, band_rows, rank, world))


def stride_rows(height: int, band_rows: int, world: int) -> int:
    """Rows reserved per rank in the gather buffer (the largest compact frame, whole bands)."""
    return ((num_bands(height, band_rows) + world - 1) // world) * band_rows


def local_to_global_row(ly: int, band_rows: int, rank: int, world: int) -> int:
    band = ly // band_rows
    return (band * world + rank) * band_rows + ly % band_rows


def extract(frame: np.ndarray, band_rows: int, rank: int, world: int) -> np.ndarray:
    """Compact band-major frame of `rank` cut out of a full frame [H, ...]."""
    parts = [frame[y0:y1] for y0, y1 in rank_bands(frame.shape[0], band_rows, rank, world)]
    return np.concatenate(parts, axis=0) if parts else frame[:0]


def assemble(gathered: np.ndarray, height: int, band_rows: int) -> np.ndarray:
    """Inverse of the split: gathered[world, stride_rows, ...] -> full frame [height, ...]."""
    world = gathered.shape[0]
    out = np.zeros((height,) + gathered.shape[2:], dtype=gathered.dtype)
    for rank in range(world):
        ly = 0
        for y0, y1 in rank_bands(height, band_rows, rank, world):
            out[y0:y1] = gathered[rank, ly:ly + (y1 - y0)]
            ly += y1 - y0
    return out
