"""Modes of the HIP renderer that have no counterpart in the reference, each checked
against the oracle's statement of the same mode (through the C-ABI), plus the
invariance of the frame under the scheduling options.

* bricked volume copy vs the linear array: the same voxels, so the frame must be
  bit-identical;
* trilinear volume filter vs the oracle's trilinear mode (same operations in the same
  order: bit-exact on a volume-only scene);
* accumulation of sub-pixel samples vs the oracle's mean of the same samples
  (tolerance as for single frames; sample 0 alone must equal the plain frame);
* tile drawing order on/off: bit-identical frames.
"""
import numpy as np
import pytest

from path_tracer_amd import scenes
from path_tracer_amd import types as T

pytestmark = pytest.mark.gpu


def _renderer(sc, options=None):
    from path_tracer_amd import HipRenderer

    r = HipRenderer()
    assert r.initialize(sc.width, sc.height)
    for k, v in (options or {}).items():
        r.setOption(k, v)
    r.setScene(sc, honour_max_depth=True)
    return r


def _frame(r, pc):
    r.render(pc)
    r.present()
    return r.readbackRGBA8(), r.readbackF32()


def _small_volume_scene(w=400, h=225, divide=8):
    res = (70, 50, 61)  # not multiples of the 8-voxel brick edge
    vox = (scenes.synthetic_walnut(res) // divide).astype(np.uint8)
    return scenes.config4(w, h, voxels=vox, resolution=res)


def test_bricked_and_linear_volume_give_the_same_frame():
    sc = _small_volume_scene()
    pc = sc.push_constants()
    frames = []
    for bricks in (0, 1):
        r = _renderer(sc, {T.OPT_VOLUME_BRICKS: bricks})
        frames.append(_frame(r, pc))
        r.shutdown()
    np.testing.assert_array_equal(frames[0][1].view(np.uint32), frames[1][1].view(np.uint32))
    np.testing.assert_array_equal(frames[0][0], frames[1][0])
    assert frames[0][0][..., :3].std() > 0  # the volume is visible


def test_trilinear_filter_matches_oracle_trilinear(oracle):
    sc = _small_volume_scene()
    pc = sc.push_constants()
    ref8, reff, refc = oracle.render(sc, nthreads=16, volume_filter=1)
    near8, _, _ = oracle.render(sc, nthreads=16)
    for bricks in (0, 1):
        r = _renderer(sc, {T.OPT_VOLUME_FILTER: 1, T.OPT_VOLUME_BRICKS: bricks})
        g8, gf = _frame(r, pc)
        cnt = r.renderCounted(pc)
        r.shutdown()
        # same operations in the same order on both sides, no pow/cos on this path
        np.testing.assert_array_equal(gf.view(np.uint32), reff.view(np.uint32))
        np.testing.assert_array_equal(g8, ref8)
        np.testing.assert_array_equal(cnt[:T.N_ALGORITHM_COUNTERS], refc[:T.N_ALGORITHM_COUNTERS])
    assert (ref8 != near8).any()  # the filter changes the picture


@pytest.mark.parametrize("box", [((0.0, 0.0, 0.0), (2.0, 2.0, 2.0)),        # the common volume: the kernel variants compiled for it
                                 ((0.25, -0.5, 0.125), (2.25, 1.5, 2.125)),  # box off the volume's position, extents 2
                                 ((0.0, 0.0, 0.0), (1.7, 2.3, 1.9)),         # extents that are no powers of two
                                 ((-0.3, 0.1, 0.0), (1.5, 2.0, 2.4))])
@pytest.mark.parametrize("filt", [0, 1])
def test_every_variant_of_the_volume_kernel_against_the_oracle(oracle, box, filt):
    """The kernel for frames without solids is compiled for the common volume (bricked, power-of-two extents, box at
    v0 = 0), once per filter, and once with everything decided at run time; a linear copy takes the last one too.
    Each must give the oracle's frame bit for bit (no pow / cos on this path), from outside and from inside the box."""
    res = (70, 50, 61)
    vox = (scenes.synthetic_walnut(res) // 6).astype(np.uint8)
    for cam in ((5.0, 1.5, 6.0), (2.4, 1.9, 0.3)):  # the second camera stands inside the box
        sc = scenes.config4(320, 180, voxels=vox, resolution=res)
        sc.volumes["v0"][0][:3] = box[0]
        sc.volumes["v1"][0][:3] = box[1]
        sc.camera_pos = cam
        pc = sc.push_constants()
        ref8, reff, refc = oracle.render(sc, nthreads=16, volume_filter=filt)
        assert int(refc[6]) > 0  # voxel samples: the rays do meet the volume
        for bricks in (1, 0):
            r = _renderer(sc, {T.OPT_VOLUME_FILTER: filt, T.OPT_VOLUME_BRICKS: bricks})
            g8, gf = _frame(r, pc)
            assert r.lastLaunch()[1] == 256 and r.lastLaunch()[2] == 1024  # the volume kernel's launch
            r.shutdown()
            np.testing.assert_array_equal(gf.view(np.uint32), reff.view(np.uint32))
            np.testing.assert_array_equal(g8, ref8)


def test_trilinear_with_solids_within_tolerance(oracle):
    res = (40, 30, 36)
    vox = scenes.synthetic_walnut(res)
    sc = scenes.demo_scene(320, 240, frame=3, with_volume=True, voxels=vox, volume_res=res)
    pc = sc.push_constants()
    ref8, reff, _ = oracle.render(sc, nthreads=16, volume_filter=1)
    r = _renderer(sc, {T.OPT_VOLUME_FILTER: 1})
    g8, gf = _frame(r, pc)
    r.shutdown()
    d8 = np.abs(g8.astype(np.int16) - ref8.astype(np.int16)).max(axis=-1)
    assert float((d8 <= 1).mean()) >= 0.999


@pytest.mark.parametrize("kind", ["objects", "pooled", "objects_and_volume", "bands", "volume", "volume_trilinear"])
def test_samples_per_launch_do_not_change_the_frames(kind):
    """pt_render_samples draws several samples per launch (PT_OPT_SAMPLE_BATCH): (tile, sample) tickets, one plane of
    clamped colours per sample, the planes added up in sample order.  The frames must be the ones that one launch per
    sample gives, bit for bit, whatever the batch: all samples in one launch, batches that do not divide the count,
    a sum carried over calls (keep)."""
    from path_tracer_amd import types as T

    if kind == "objects":
        sc = scenes.config2(328, 188, n_spheres=20, n_ellipsoids=6, max_depth=5)  # not multiples of the 8-pixel tile
    elif kind == "pooled":
        sc = scenes.config3(200, 120, n_spheres=300, max_depth=3)
    elif kind == "objects_and_volume":
        res = (70, 50, 61)
        sc = scenes.config5(240, 136, voxels=(scenes.synthetic_walnut(res) // 8).astype(np.uint8), resolution=res)
    elif kind in ("volume", "volume_trilinear"):  # frames without solids: the volume kernel, a wave per (tile, sample)
        sc = _small_volume_scene(w=328, h=188)
    else:
        sc = scenes.config2(320, 180, n_spheres=12, n_ellipsoids=4, max_depth=4)
    pc = sc.push_constants()
    n = 7

    def draw(batch, calls):
        r = _renderer(sc, {T.OPT_SAMPLE_BATCH: batch, T.OPT_VOLUME_FILTER: 1 if kind == "volume_trilinear" else 0})
        if kind == "bands":
            r.setRowBands(8, 1, 3)
        first, keep = 0, False
        for c in calls:
            r.renderSamples(pc, first, c, keep=keep)
            first, keep = first + c, True
        r.present()
        out = r.readbackRGBA8().copy(), r.readbackF32().copy()
        r.shutdown()
        return out

    want8, wantf = draw(1, [n])
    for batch, calls in ((0, [n]), (3, [n]), (2, [4, 3]), (0, [1, n - 1]), (64, [2, 2, 3])):
        got8, gotf = draw(batch, calls)
        np.testing.assert_array_equal(gotf.view(np.uint32), wantf.view(np.uint32), err_msg=f"{kind} batch {batch} calls {calls}")
        np.testing.assert_array_equal(got8, want8, err_msg=f"{kind} batch {batch} calls {calls}")


def test_sample_zero_is_the_plain_frame_and_mean_matches_oracle(oracle):
    sc = scenes.config2(320, 180, n_spheres=20, n_ellipsoids=6, max_depth=4)
    pc = sc.push_constants()
    r = _renderer(sc)
    plain8, plainf = _frame(r, pc)
    r.renderSamples(pc, 0, 1)
    r.present()
    s0 = r.readbackF32()
    np.testing.assert_array_equal(s0.view(np.uint32), plainf.view(np.uint32))

    n = 4
    r.renderSamples(pc, 0, n)
    r.present()
    g8, gf = r.readbackRGBA8(), r.readbackF32()
    # the same mean, built progressively over two calls
    r.renderSamples(pc, 0, 1)
    r.renderSamples(pc, 1, n - 1, keep=True)
    r.present()
    gf2 = r.readbackF32()
    r.shutdown()
    np.testing.assert_array_equal(gf.view(np.uint32), gf2.view(np.uint32))

    ref8, reff, _ = oracle.render(sc, nthreads=16, num_samples=n)
    d8 = np.abs(g8.astype(np.int16) - ref8.astype(np.int16)).max(axis=-1)
    frac = float((d8 <= 1).mean())
    err = np.abs(gf - reff)
    ok = (err <= 1e-4 * np.maximum(np.abs(reff), 1e-2)).all(axis=-1)
    print(f"[accumulate x{n}] rgba8<=1LSB {frac:.6f} float<=1e-4 {float(ok.mean()):.6f}")
    assert frac >= 0.999 and float(ok.mean()) >= 0.999
    assert (g8 != plain8).any()  # edges are smoothed


def test_tile_order_and_object_source_do_not_change_the_frame():
    sc = scenes.config2(480, 270, max_depth=6)
    pc = sc.push_constants()
    frames = []
    for order in (0, 1, 2):
        r = _renderer(sc, {T.OPT_TILE_ORDER: order})
        frames.append(_frame(r, pc)[1])
        r.shutdown()
    for f in frames[1:]:
        np.testing.assert_array_equal(frames[0].view(np.uint32), f.view(np.uint32))


@pytest.mark.parametrize("in_flight", [2, 3, 4, 8])
def test_frames_in_flight_show_the_latest_frame(in_flight):
    """Consecutive pt_render calls go round the slots; what readback shows is always the
    most recent frame, bit-identical to the same frame drawn one at a time."""
    import copy

    sc = scenes.config2(320, 180, n_spheres=16, n_ellipsoids=5, max_depth=4)
    pcs = []
    for k in range(6):
        s2 = copy.copy(sc)
        s2.camera_pos = (5.0 - 0.7 * k, 1.5 + 0.2 * k, 6.0)
        pcs.append(s2.push_constants())
    solo = _renderer(sc, {T.OPT_FRAMES_IN_FLIGHT: 1})
    want = [_frame(solo, pc)[1].copy() for pc in pcs]
    solo.shutdown()
    assert not np.array_equal(want[0], want[1])
    r = _renderer(sc, {T.OPT_FRAMES_IN_FLIGHT: in_flight})
    for pc in pcs:          # six frames queued back to back, no wait in between
        r.render(pc)
    assert r.lastSlot() == (len(pcs) - 1 + 1) % in_flight  # the first frame goes to slot 1
    r.present()
    np.testing.assert_array_equal(r.readbackF32().view(np.uint32), want[5].view(np.uint32))
    for k in (1, 0, 2):     # and one by one, in another order
        r.render(pcs[k])
        np.testing.assert_array_equal(r.readbackF32().view(np.uint32), want[k].view(np.uint32))
    cnt = r.renderCounted(pcs[0])  # the instrumented frame uses slot 0 whatever is in flight
    assert int(cnt[0]) > 0
    r.render(pcs[3])
    np.testing.assert_array_equal(r.readbackF32().view(np.uint32), want[3].view(np.uint32))
    r.shutdown()


@pytest.mark.parametrize("in_flight", [2, 3, 8])
def test_scene_updates_between_frames_in_flight(in_flight):
    """The reference's loop calls updateScene before every render (reference src/main_vulkan.cpp:352-374).  Here an
    update does not wait for the frames in flight: each frame slot refreshes its own copy of the scene in front of
    its next frame.  Every frame of a sequence queued back to back (update, render, update, render, ... with the
    ellipsoids re-oriented, a sphere moved and the object counts changing) must be the frame its own scene gives
    when drawn alone."""
    torch = pytest.importorskip("torch")
    import copy

    base = scenes.config2(320, 180, n_spheres=16, n_ellipsoids=5, max_depth=4)
    versions, pcs = [], []
    for k in range(7):
        s2 = copy.copy(base)
        s2.ellipsoids = base.ellipsoids.copy()
        s2.ellipsoids["rotation"] = scenes.axis_angle_quat((2.0, 1.0, 0.0), (k + 1) / 7.0)
        s2.spheres = base.spheres.copy()
        s2.spheres["center"][3] += np.float32(0.15 * k)
        if k % 3 == 2:  # fewer objects: the scene image shrinks and its offsets move
            s2.spheres = s2.spheres[:-3].copy()
            s2.ellipsoids = s2.ellipsoids[:-1].copy()
        s2.camera_pos = (5.0 - 0.3 * k, 1.5, 6.0)
        versions.append(s2)
        pcs.append(s2.push_constants())
    want = []
    for s2, pc in zip(versions, pcs):
        solo = _renderer(s2, {T.OPT_FRAMES_IN_FLIGHT: 1})
        want.append(_frame(solo, pc)[0].copy())
        solo.shutdown()
    assert not np.array_equal(want[0], want[1])

    r = _renderer(versions[0], {T.OPT_FRAMES_IN_FLIGHT: in_flight, T.OPT_FLOAT_OUTPUT: 0})
    outs = [torch.zeros((base.height, base.width, 4), dtype=torch.uint8, device="cuda:0") for _ in range(in_flight)]
    streams = [torch.cuda.Stream(device="cuda:0") for _ in range(in_flight)]
    torch.cuda.synchronize()
    slot = 1 % in_flight  # the first frame goes to slot 1
    for s2, pc in zip(versions, pcs):  # the whole sequence queued back to back, no wait in between
        r.updateScene(s2.spheres, s2.ellipsoids, s2.materials, s2.lights, s2.volumes)
        r.render(pc)
        assert r.lastSlot() == slot
        slot = (slot + 1) % in_flight
    r.present()
    # the last `in_flight` frames are still in their slots' buffers: look at the most recent one through readback ...
    np.testing.assert_array_equal(r.readbackRGBA8(), want[-1])
    # ... and draw the whole sequence again, this time reading every frame back (readback waits, so nothing overlaps):
    # the slots' scene copies must follow the updates whichever slot a frame lands in
    for k in (3, 0, 6, 2, 5):
        s2 = versions[k]
        r.updateScene(s2.spheres, s2.ellipsoids, s2.materials, s2.lights, s2.volumes)
        r.render(pcs[k])
        np.testing.assert_array_equal(r.readbackRGBA8(), want[k])
    # caller-owned frames per slot: queue `in_flight` updates + frames back to back and look at all of them
    r.present()
    for j in range(in_flight):
        r.setSlot(j, streams[j].cuda_stream, outs[j].data_ptr(), None)
    first = r.lastSlot()
    order = []
    for j in range(min(in_flight, len(versions))):
        s2 = versions[j]
        r.updateScene(s2.spheres, s2.ellipsoids, s2.materials, s2.lights, s2.volumes)
        r.render(pcs[j])
        order.append(r.lastSlot())
    r.present()
    torch.cuda.synchronize()
    assert len(set(order)) == len(order), (first, order)
    for j, sl in enumerate(order):
        np.testing.assert_array_equal(outs[sl].cpu().numpy(), want[j])
    r.shutdown()


def test_a_larger_resolution_later_keeps_the_uploaded_voxels(oracle):
    """The reference uploads the voxel bytes once and indexes them with whatever resolution the current
    scene states (reference src/VulkanRenderer.cpp:961-987); a later update that states a larger
    resolution must still see those bytes (and zeros beyond them), with or without the voxels passed again."""
    from path_tracer_amd import HipRenderer

    small, big = (20, 16, 12), (24, 20, 16)
    vox = scenes.synthetic_walnut(small)
    sc_small = scenes.config4(160, 90, voxels=vox, resolution=small)
    sc_big = scenes.config4(160, 90, voxels=vox, resolution=big)  # same bytes, re-indexed; the tail reads as 0
    padded = np.zeros(big[0] * big[1] * big[2], np.uint8)
    padded[:vox.size] = vox
    want = oracle.render(scenes.config4(160, 90, voxels=padded, resolution=big), nthreads=8)
    for resend in (True, False):
        r = HipRenderer()
        assert r.initialize(160, 90)
        r.setScene(sc_small, honour_max_depth=True)
        r.render(sc_small.push_constants()); r.present()
        r.updateScene(sc_big.spheres, sc_big.ellipsoids, sc_big.materials, sc_big.lights, sc_big.volumes,
                      vox if resend else None)
        cnt = r.renderCounted(sc_big.push_constants())
        r.render(sc_big.push_constants()); r.present()
        got = r.readbackF32()
        r.shutdown()
        np.testing.assert_array_equal(cnt[:10], want[2][:10])
        np.testing.assert_array_equal(got.view(np.uint32), want[1].view(np.uint32))
        assert got[..., :3].max() > 0 and (got[..., :3] != got[0, 0, :3]).any()  # the volume is visible


def test_wave_timeline_draws_the_same_frame_and_orders_its_stamps():
    """pt_wave_timeline: one record per wave of the launch, start <= tickets drained <= exit on the chip-wide
    clock, at least one pass each, and the frame it draws is the plain frame."""
    sc = scenes.config2(480, 270, max_depth=6)
    pc = sc.push_constants()
    r = _renderer(sc)
    plain = _frame(r, pc)[1]
    t = r.waveTimeline(pc)
    blocks, threads, _ = r.lastLaunch()
    assert t.shape == (blocks * threads // 64, 4)
    r.present()
    np.testing.assert_array_equal(plain.view(np.uint32), r.readbackF32().view(np.uint32))
    start, dry, end = (t[:, i].astype(np.int64) for i in range(3))
    passes, passes_at_dry = (t[:, 3] & np.uint64(0xFFFFFFFF)).astype(np.int64), (t[:, 3] >> np.uint64(32)).astype(np.int64)
    assert (passes >= 1).all() and (end >= start).all() and (passes_at_dry <= passes).all()
    found = dry != 0  # a wave that never needed another tile has no such stamp
    assert found.any() and (dry[found] >= start[found]).all() and (end[found] >= dry[found]).all()
    assert (end.max() - start.min()) < 100_000_000  # under a second of the 100 MHz clock
    r.shutdown()


def test_launches_sized_by_the_streams_side_by_side_draw_the_same_frame():
    """With frames in flight a frame with few tiles per wave takes a part of the compute units only (as many
    parts as streams really run side by side, which the library measures); the frames stay bit-identical to
    frames drawn one at a time, for the pooled kernel (>= 256 spheres) and the other."""
    for sc in (scenes.config2(640, 360, max_depth=5), scenes.config3(640, 360, n_spheres=400, max_depth=4)):
        pc = sc.push_constants()
        solo = _renderer(sc, {T.OPT_FRAMES_IN_FLIGHT: 1})
        want = _frame(solo, pc)[1].copy()
        full_blocks = solo.lastLaunch()[0]
        assert solo.streamConcurrency() == 1
        solo.shutdown()
        r = _renderer(sc, {T.OPT_FRAMES_IN_FLIGHT: 8})
        n = r.streamConcurrency()
        assert 1 <= n <= 8
        for _ in range(9):
            r.render(pc)
        blocks = r.lastLaunch()[0]
        assert blocks <= full_blocks
        if n >= 4:
            assert blocks < full_blocks  # the part of the chip one frame takes
        r.present()
        np.testing.assert_array_equal(r.readbackF32().view(np.uint32), want.view(np.uint32))
        for k in range(8):  # every slot holds the same frame
            r.render(pc)
            np.testing.assert_array_equal(r.readbackF32().view(np.uint32), want.view(np.uint32))
        r.shutdown()


def test_present_lag_shows_the_frame_before_the_one_just_queued():
    """PT_OPT_PRESENT_LAG = 1: present() after a render() waits for the earlier frames only and the readbacks show the
    frame before the one just queued, so the reference's loop (updateScene, render, present per frame) keeps a frame
    drawing while the host prepares the next.  The frames are the ones a present() that drains everything shows, one
    call later; a present() without a render() in between shows the last frame."""
    import math

    sc = scenes.config2(480, 270, n_spheres=20, n_ellipsoids=6)
    pcs = []
    for k in range(6):
        pc = sc.push_constants().copy()
        pc["cameraPos"] = (np.float32(2.0 + 3.0 * math.cos(0.2 * k)), 1.5, np.float32(6.0 + 3.0 * math.sin(0.2 * k)))
        pcs.append(pc)
    r = _renderer(sc)
    want = [_frame(r, pc)[1].copy() for pc in pcs]  # drained after every frame
    assert any((want[0] != w).any() for w in want[1:])
    r.setOption(T.OPT_PRESENT_LAG, 1)
    for k, pc in enumerate(pcs):
        r.render(pc)
        r.present()
        shown = r.readbackF32()
        # the frame queued before this one: for k = 0 that is the last frame of the loop above
        np.testing.assert_array_equal(shown.view(np.uint32), want[k - 1].view(np.uint32))
    r.present()  # nothing queued since: the last frame
    np.testing.assert_array_equal(r.readbackF32().view(np.uint32), want[-1].view(np.uint32))
    # one frame at a time has nothing to lag behind: the option changes nothing there
    r.setOption(T.OPT_FRAMES_IN_FLIGHT, 1)
    r.setOption(T.OPT_PRESENT_LAG, 1)
    for k, pc in enumerate(pcs[:3]):
        np.testing.assert_array_equal(_frame(r, pc)[1].view(np.uint32), want[k].view(np.uint32))
    r.shutdown()


@pytest.mark.parametrize("options", [{}, {T.OPT_VOLUME_FILTER: 1}, {T.OPT_VOLUME_BRICKS: 0}])
def test_frames_without_solids_are_the_same_on_either_kernel(options):
    """A frame with a volume and no spheres or ellipsoids is drawn by a kernel of its own (camera ray, march,
    background); tune bit 25 keeps it on the general kernel.  Same frame, bit for bit, also as one rank's bands
    and as the mean of sub-pixel samples."""
    sc = _small_volume_scene()
    pc = sc.push_constants()
    frames = []
    for tune in (0, 1 << 25):
        r = _renderer(sc, {**options, 100: tune})
        f = [_frame(r, pc)[1].copy()]
        assert (r.lastLaunch()[1] == 256 and r.lastLaunch()[2] == 1024) == (tune == 0)  # the volume kernel's launch
        r.renderSamples(pc, 0, 3)
        r.present()
        f.append(r.readbackF32().copy())
        r.setRowBands(16, 1, 3)
        f.append(_frame(r, pc)[1].copy())
        frames.append(f)
        r.shutdown()
    for a, b in zip(*frames):
        assert a.shape == b.shape
        np.testing.assert_array_equal(a.view(np.uint32), b.view(np.uint32))
