"""Multi-GPU frame split: one process per GPU, interleaved row bands, one
gather of the compact band-major frames to rank 0 (RCCL when the tensors live
on GPUs — torch's "nccl" backend — gloo on CPU), then de-interleave.

The reference is single-GPU; pixels are independent
(reference shaders/raytrace.comp:686-705), so the only exchange is this gather.
"""
from __future__ import annotations

from typing import Optional

import numpy as np

from . import bands


class BandGather:
    """Buffers and the collective for one frame size.

    ``local`` is the tensor a rank renders its compact frame into
    ([stride_rows, width, 4] uint8; rows past ``local_rows`` are padding so
    every rank sends the same amount).  ``gather()`` leaves all ranks' frames
    in ``gathered`` on rank 0.
    """

    def __init__(self, torch_mod, dist, width: int, height: int, rank: int, world: int, device: str,
                 band_rows: int = bands.DEFAULT_BAND_ROWS, channels: int = 4, dtype=None):
        self.torch, self.dist = torch_mod, dist
        self.width, self.height, self.rank, self.world, self.band_rows = width, height, rank, world, band_rows
        self.local_rows = bands.local_rows(height, band_rows, rank, world)
        self.stride_rows = bands.stride_rows(height, band_rows, world)
        dtype = torch_mod.uint8 if dtype is None else dtype
        self.local = torch_mod.zeros((self.stride_rows, width, channels), dtype=dtype, device=device)
        self.gathered = None
        if rank == 0:
            self.gathered = torch_mod.zeros((world, self.stride_rows, width, channels), dtype=dtype, device=device)

    def gather(self) -> None:
        if self.rank == 0:
            self.dist.gather(self.local, list(self.gathered.unbind(0)), dst=0)
        else:
            self.dist.gather(self.local, None, dst=0)

    def assemble_host(self) -> Optional[np.ndarray]:
        """Rank 0: the full frame as a numpy array (de-interleaved on the host)."""
        if self.rank != 0:
            return None
        return bands.assemble(self.gathered.cpu().numpy(), self.height, self.band_rows)
