"""The N > 1 path on CPU: two gloo ranks each render their interleaved row
bands (with the CPU oracle standing in for the device kernel), gather to rank 0
and de-interleave; the result must equal the single-process frame.

What this covers: the band arithmetic (which rows a rank owns, the compact
band-major layout, the padded stride of the gather buffer), the gather through
torch.distributed and the host-side de-interleave, with two real processes.
What it does not cover: the device kernel drawing a rank's share, the device
de-interleave kernel, the RCCL gather inside the library and the frame slots
and streams of the bench; tests/test_gpu_multigpu.py runs those on one GPU
(every rank's share of 2, 3 and 8 ranks in turn, RCCL and torch-nccl with a
world of one)."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, out_path):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist

    from oracle import oracle_py
    from path_tracer_amd import bands, scenes
    from path_tracer_amd.distributed import BandGather

    dist.init_process_group("gloo", rank=rank, world_size=world)
    sc = scenes.config2(96, 70, n_spheres=10, n_ellipsoids=4)  # 70 rows: ragged last band
    bg = BandGather(torch, dist, sc.width, sc.height, rank, world, "cpu", band_rows=16)
    # render only this rank's rows, band by band, into the compact frame
    ly = 0
    for y0, y1 in bands.rank_bands(sc.height, 16, rank, world):
        rgba, _, _ = oracle_py.render(sc, rows=(y0, y1), nthreads=2, want_f32=False)
        bg.local[ly:ly + (y1 - y0)] = torch.from_numpy(rgba[y0:y1])
        ly += y1 - y0
    assert ly == bg.local_rows
    bg.gather()
    if rank == 0:
        np.save(out_path, bg.assemble_host())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_band_split_gathers_to_the_single_process_frame(tmp_path):
    import torch.multiprocessing as mp

    from oracle import oracle_py
    from path_tracer_amd import scenes

    out_path = str(tmp_path / "frame.npy")
    mp.spawn(_worker, args=(2, _free_port(), out_path), nprocs=2, join=True)
    got = np.load(out_path)
    sc = scenes.config2(96, 70, n_spheres=10, n_ellipsoids=4)
    full, _, _ = oracle_py.render(sc, nthreads=2, want_f32=False)
    np.testing.assert_array_equal(got, full)
