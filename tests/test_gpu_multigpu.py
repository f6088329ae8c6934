"""The device side of the multi-GPU split on one GPU: what each rank does (pt_set_row_bands, render into a
compact band-major frame, frame slots), the rank-0 de-interleave kernel (pt_assemble_bands) and the gather
entry points (pt_comm_* / pt_gather_bands over RCCL, and torch.distributed's nccl backend) with one rank.
The shares of 2, 3 and 8 ranks are rendered one after the other by one renderer into the buffer a gather
would fill; the assembled frame must equal the frame drawn without a split, for rgba8 and float4 frames,
with a ragged last band.  (The exchange between real ranks needs several GPUs; the 2-rank CPU test covers
the band arithmetic on the host side with gloo.)"""
import os

import numpy as np
import pytest

from path_tracer_amd import HipRenderer, bands, scenes

pytestmark = pytest.mark.gpu
BAND = 16


def _plain(sc):
    r = HipRenderer()
    assert r.initialize(sc.width, sc.height)
    r.setScene(sc, honour_max_depth=True)
    r.render(sc.push_constants())
    r.present()
    out = r.readbackRGBA8(), r.readbackF32()
    r.shutdown()
    return out


@pytest.mark.parametrize("world", [2, 3, 8])
def test_assembled_shares_equal_the_plain_frame(world):
    torch = pytest.importorskip("torch")
    sc = scenes.config2(328, 203, n_spheres=18, n_ellipsoids=5, max_depth=5)  # 203 rows: 12 bands of 16 + one of 11
    want8, wantf = _plain(sc)
    stride = bands.stride_rows(sc.height, BAND, world)
    g8 = torch.zeros((world, stride, sc.width, 4), dtype=torch.uint8, device="cuda:0")
    gf = torch.zeros((world, stride, sc.width, 4), dtype=torch.float32, device="cuda:0")
    r = HipRenderer()
    assert r.initialize(sc.width, sc.height)
    r.setScene(sc, honour_max_depth=True)
    pc = sc.push_constants()
    for rank in range(world):  # every rank's share in turn, straight into its place in the gather buffer
        r.setRowBands(BAND, rank, world)
        assert r.localRows() == bands.local_rows(sc.height, BAND, rank, world)
        r.setSlot(0, None, g8[rank].data_ptr(), gf[rank].data_ptr())
        r.render(pc)
        r.present()
    out8 = torch.zeros((sc.height, sc.width, 4), dtype=torch.uint8, device="cuda:0")
    outf = torch.zeros((sc.height, sc.width, 4), dtype=torch.float32, device="cuda:0")
    r.assembleBands(g8.data_ptr(), out8.data_ptr(), 4, BAND, world, stride)
    r.assembleBands(gf.data_ptr(), outf.data_ptr(), 16, BAND, world, stride)
    r.present()
    torch.cuda.synchronize()
    np.testing.assert_array_equal(out8.cpu().numpy(), want8)
    np.testing.assert_array_equal(outf.cpu().numpy().view(np.uint32), wantf.view(np.uint32))
    r.shutdown()


def test_rccl_gather_entry_points_with_one_rank():
    """pt_comm_unique_id / pt_comm_create / pt_gather_bands with world = 1: the communicator is real (RCCL is
    loaded and initialised), the gather is a copy plus the de-interleave; eight frames in flight."""
    sc = scenes.config2(320, 180, n_spheres=14, n_ellipsoids=4, max_depth=4)
    want8, wantf = _plain(sc)
    r = HipRenderer()
    assert r.initialize(sc.width, sc.height)
    r.setOption(6, 8)  # PT_OPT_FRAMES_IN_FLIGHT
    r.setScene(sc, honour_max_depth=True)
    comm = r.commCreate(HipRenderer.commUniqueId(), 0, 1, 0)
    r.setRowBands(BAND, 0, 1)
    pc = sc.push_constants()
    for _ in range(10):  # goes round the slots; every frame is gathered on its own stream
        r.render(pc)
        r.gatherBands(comm, 4)
    r.present()
    np.testing.assert_array_equal(r.readbackAssembled(4), want8)
    r.setOption(6, 1)
    r.render(pc)
    r.gatherBands(comm, 16)
    r.present()
    np.testing.assert_array_equal(r.readbackAssembled(16).view(np.uint32), wantf.view(np.uint32))
    r.commDestroy(comm)
    r.shutdown()


def test_torch_nccl_gather_with_one_rank():
    """The torch.distributed path (backend "nccl" = RCCL) the Python BandGather uses, world size 1:
    pt_set_row_bands + pt_set_slot + dist.gather + pt_assemble_bands against the plain frame."""
    torch = pytest.importorskip("torch")
    import torch.distributed as dist

    from path_tracer_amd.distributed import BandGather

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29571")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    sc = scenes.config3(256, 150, n_spheres=300, max_depth=4)  # the pooled variant of the kernel
    want8, _ = _plain(sc)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
    try:
        r = HipRenderer()
        assert r.initialize(sc.width, sc.height)
        r.setOption(1, 0)
        r.setOption(6, 2)
        r.setScene(sc, honour_max_depth=True)
        r.setRowBands(BAND, 0, 1)
        bgs, streams, frames = [], [], []
        for slot in range(2):
            bg = BandGather(torch, dist, sc.width, sc.height, 0, 1, "cuda:0", band_rows=BAND)
            ts = torch.cuda.Stream(device="cuda:0")
            r.setSlot(slot, ts.cuda_stream, bg.local.data_ptr(), None)
            bgs.append(bg); streams.append(ts)
            frames.append(torch.zeros((sc.height, sc.width, 4), dtype=torch.uint8, device="cuda:0"))
        pc = sc.push_constants()
        for _ in range(5):
            r.render(pc)
            s = r.lastSlot()
            with torch.cuda.stream(streams[s]):
                bgs[s].gather()
            r.assembleBands(bgs[s].gathered.data_ptr(), frames[s].data_ptr(), 4, BAND, 1, bgs[s].stride_rows)
        r.present()
        torch.cuda.synchronize()
        np.testing.assert_array_equal(frames[r.lastSlot()].cpu().numpy(), want8)
        r.shutdown()
    finally:
        dist.destroy_process_group()


def test_eight_slots_run_side_by_side_next_to_an_rccl_communicator():
    """Eight frames in flight need eight hardware queues for the slots' streams, and an RCCL communicator in the process
    takes queues of its own: with GPU_MAX_HW_QUEUES = 16 (what bench.py sets when the frame is split) the library must
    measure 8 streams side by side although the communicator was there first.  The variable is read when HIP starts, so
    the check runs in a process of its own."""
    import subprocess
    import sys

    code = r'''
import sys
sys.path.insert(0, ".")
from path_tracer_amd import HipRenderer, scenes
sc = scenes.config3(960, 540, n_spheres=300, max_depth=3)
r = HipRenderer(); assert r.initialize(sc.width, sc.height)
comm = r.commCreate(HipRenderer.commUniqueId(), 0, 1, 0)   # before the eight slots' streams exist
r.setOption(1, 0); r.setOption(6, 8); r.setScene(sc, honour_max_depth=True); r.setRowBands(16, 0, 1)
pc = sc.push_constants()
for i in range(16): r.render(pc); r.gatherBands(comm, 4)
r.present()
print("side_by_side", r.streamConcurrency())
r.commDestroy(comm); r.shutdown()
'''
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, GPU_MAX_HW_QUEUES="16", HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, "-c", code], cwd=root, env=env, capture_output=True, text=True, timeout=170)
    assert out.returncode == 0, out.stdout + out.stderr
    line = [l for l in out.stdout.splitlines() if l.startswith("side_by_side")]
    assert line and int(line[0].split()[1]) == 8, out.stdout + out.stderr
