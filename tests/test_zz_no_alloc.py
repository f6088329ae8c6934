"""pt_render_static (the entry point that allocates nothing: its scratch is a static device array)
against pt_render on the same renderer: bit-identical frames, two calls in a row, the second on a
smaller frame with fewer objects, both sharing the GPU's one static scratch array."""
import numpy as np
import pytest

from path_tracer_amd import HipRenderer, scenes

pytestmark = pytest.mark.gpu


def _both(sc, torch, with_float):
    r = HipRenderer()
    assert r.initialize(sc.width, sc.height)
    try:
        r.setScene(sc, honour_max_depth=True)
        pc = sc.push_constants()
        r.render(pc)
        r.present()
        want8, wantf = r.readbackRGBA8(), r.readbackF32()
        out8 = torch.full((sc.height, sc.width, 4), 7, dtype=torch.uint8, device="cuda:0")
        outf = torch.full((sc.height, sc.width, 4), -1.0, dtype=torch.float32, device="cuda:0") if with_float else None
        s = torch.cuda.Stream(device="cuda:0")
        torch.cuda.synchronize()
        # first calls: the HIP runtime sets up what a stream's first launch needs (queue, signals, kernel arguments)
        r.renderStatic(pc, out8.data_ptr(), None if outf is None else outf.data_ptr())
        r.present()  # a null stream means the stream of frame slot 0, which present() waits for
        r.renderStatic(pc, out8.data_ptr(), None, s.cuda_stream)
        s.synchronize()
        # after that nothing may be allocated by the static calls: the device's free memory does not shrink across
        # them (this sees allocations the driver serves with new pages: anything of the size of a frame or a scratch block)
        free_before = torch.cuda.mem_get_info(0)[0]
        for _ in range(3):
            r.renderStatic(pc, out8.data_ptr(), None, s.cuda_stream)
            s.synchronize()
            r.renderStatic(pc, out8.data_ptr(), None if outf is None else outf.data_ptr())
            r.present()
        free_after = torch.cuda.mem_get_info(0)[0]
        assert free_after >= free_before, (free_before, free_after)
        got8 = out8.cpu().numpy()
        np.testing.assert_array_equal(got8, want8)
        if with_float:
            np.testing.assert_array_equal(outf.cpu().numpy().view(np.uint32), wantf.view(np.uint32))
        # the renderer's own frame slots are untouched by the static call
        np.testing.assert_array_equal(r.readbackRGBA8(), want8)
        # on a caller's stream
        out8.fill_(3)
        torch.cuda.synchronize()
        r.renderStatic(pc, out8.data_ptr(), None, s.cuda_stream)
        s.synchronize()
        np.testing.assert_array_equal(out8.cpu().numpy(), want8)
    finally:
        r.shutdown()


def test_static_entry_point_matches_pt_render_twice_in_a_row():
    torch = pytest.importorskip("torch")
    assert torch.cuda.is_available()
    # first call: mirrors and transparency -> the tile-order lists in the static scratch are used
    _both(scenes.config2(448, 252, n_spheres=30, n_ellipsoids=8, max_depth=6), torch, with_float=True)
    # second call: a smaller frame with fewer objects; the scratch still holds the first call's contents
    _both(scenes.config2(160, 90, n_spheres=9, n_ellipsoids=3, max_depth=4), torch, with_float=False)
    # and a scene that takes the pooled variant of the kernel plus the volume march
    _both(scenes.config3(192, 108, n_spheres=300, max_depth=4), torch, with_float=True)
    # a frame without solids: the volume kernel, which uses none of the scratch
    res = (40, 30, 36)
    _both(scenes.config4(200, 120, voxels=(scenes.synthetic_walnut(res) // 8).astype(np.uint8), resolution=res), torch, with_float=True)


def test_static_entry_point_rejects_what_it_cannot_hold():
    from path_tracer_amd._lib import PtError

    sc = scenes.config1(64, 48)
    r = HipRenderer()
    assert r.initialize(8192, 4096)  # 524 288 tiles > 262 144
    try:
        r.setScene(sc, honour_max_depth=True)
        with pytest.raises(PtError, match="static scratch"):
            r.renderStatic(sc.push_constants(), 0x1000)
        with pytest.raises(PtError, match="required"):
            r.renderStatic(sc.push_constants(), None)
    finally:
        r.shutdown()
