"""Differential test on random scenes: the HIP renderer (through the C-ABI) against the oracle.

Every scene draws its object counts, materials (plain, specular, mirror, transparent, emissive, rainbow and toon
shader, unknown shader index), lights, camera, bounce limit, background and an optional small volume from a seeded
generator; frames are small so that the oracle takes well under a second.  Work counters (rays by kind, tests,
roots, voxel samples, shaded hits) must be identical, the frame within the stated tolerance.
"""
import numpy as np
import pytest

from path_tracer_amd import scenes
from path_tracer_amd import types as T
from path_tracer_amd.scenes import Material

pytestmark = pytest.mark.gpu
NA = T.N_ALGORITHM_COUNTERS


def random_scene(seed: int) -> scenes.Scene:
    rng = np.random.default_rng(seed)
    u = lambda lo, hi: float(rng.uniform(lo, hi))
    col = lambda: (u(0.1, 1.0), u(0.1, 1.0), u(0.1, 1.0))
    makers = [
        lambda: Material.Diffuse(col(), u(0.3, 0.9), u(0.0, 0.3)),
        lambda: Material.Specular(col(), u(0.1, 0.9), float(rng.choice([1.0, 8.0, 32.0, 100.5, 256.0]))),
        lambda: Material.Mirror(col(), u(0.02, 0.95)),
        lambda: Material(color=col(), transparency=u(0.02, 0.9)),
        lambda: Material.Emissive(col(), u(0.2, 3.0)),
        lambda: Material.CustomShader(col(), T.SHADER_RAINBOW),
        lambda: Material.CustomShader(col(), T.SHADER_TOON),
        lambda: Material.CustomShader(col(), "shaders/surface_shaders/none.glsl"),  # unknown path: index 0
    ]
    mats = [makers[int(rng.integers(len(makers)))]() for _ in range(int(rng.integers(1, 9)))]
    if rng.random() < 0.5:   # rainbow on an emitter: the emissive-sphere sampling evaluates the shader too
        m = Material.CustomShader(col(), T.SHADER_RAINBOW)
        m.emissive_strength = u(0.5, 2.0)
        mats.append(m)
    w, h = int(rng.integers(40, 130)), int(rng.integers(30, 90))
    sc = scenes.Scene(width=w, height=h, name=f"random{seed}")
    sph, ell = [], []
    if rng.random() < 0.7:
        sph.append(scenes._sphere((0.0, -100.5, -1.0), 100.0, int(rng.integers(len(mats)))))
    for _ in range(int(rng.integers(0, 40))):
        c = (u(-3, 7), u(-0.4, 3), u(-6, 3))
        sph.append(scenes._sphere(c, u(0.05, 0.9), int(rng.integers(len(mats)))))
    for _ in range(int(rng.integers(0, 14))):
        c = (u(-3, 7), u(-0.4, 3), u(-6, 3))
        rad = (u(0.08, 0.8), u(0.08, 0.8), u(0.08, 0.8))
        q = rng.normal(size=4)
        q /= np.linalg.norm(q)
        if rng.random() < 0.2:
            q *= u(0.7, 1.4)  # the shader applies whatever quaternion it is given
        ell.append(scenes._ellipsoid(c, rad, int(rng.integers(len(mats))), q.astype(np.float32)))
    lights = [scenes._light((u(-6, 8), u(1, 9), u(-6, 8)), u(0.3, 2.5)) for _ in range(int(rng.integers(0, 5)))]
    vol_mat = None
    if rng.random() < 0.35:
        vol_mat = len(mats)
        mats.append(Material.Volumetric(col(), u(0.5, 8.0)))
        res = (int(rng.integers(5, 40)), int(rng.integers(5, 40)), int(rng.integers(5, 40)))
        sc.volumes = scenes._stack([scenes._volume((u(0, 2), u(0, 1.5), u(-2, 0)), (0, 0, 0), (u(0.5, 2), u(0.5, 2), u(0.5, 2)),
                                                   res, vol_mat)], T.GPU_VOLUME)
        sc.voxels = (scenes.synthetic_walnut(res) // int(rng.integers(1, 9))).astype(np.uint8)
    sc.materials = scenes._materials(mats)
    if sph:
        sc.spheres = scenes._stack(sph, T.GPU_SPHERE)
    if ell:
        sc.ellipsoids = scenes._stack(ell, T.GPU_ELLIPSOID)
    if lights:
        sc.lights = scenes._stack(lights, T.GPU_LIGHT)
    ang = u(0, 2 * np.pi)
    sc.camera_pos = (2.0 + u(2, 6) * np.cos(ang), u(0.2, 4.0), u(2, 6) * np.sin(ang))
    sc.time_ms = u(0, 5000.0)
    sc.max_depth = int(rng.integers(1, 11))
    sc.bg_top, sc.bg_bottom = col(), col()
    return sc


@pytest.mark.parametrize("seed", range(64))
def test_random_scene_matches_the_oracle(oracle, seed):
    from path_tracer_amd import HipRenderer

    sc = random_scene(1000 + seed)
    r = HipRenderer()
    assert r.initialize(sc.width, sc.height)
    r.setScene(sc, honour_max_depth=True)
    pc = sc.push_constants()
    r.render(pc)
    r.present()
    g8, gf = r.readbackRGBA8(), r.readbackF32()
    cnt = r.renderCounted(pc)
    r.shutdown()
    r8, rf, rc = oracle.render(sc, nthreads=8)
    np.testing.assert_array_equal(cnt[:NA], rc[:NA])
    d8 = np.abs(g8.astype(np.int16) - r8.astype(np.int16)).max(axis=-1)
    both_nan = np.isnan(gf) & np.isnan(rf)
    okf = ((np.abs(gf - rf) <= 1e-4 * np.maximum(np.abs(rf), 1e-2)) | both_nan).all(axis=-1)
    assert float((d8 <= 1).mean()) >= 0.999 and float(okf.mean()) >= 0.999, (sc.name, float((d8 <= 1).mean()), float(okf.mean()))


def random_hall_of_mirrors(seed: int) -> scenes.Scene:
    """Several hundred spheres, mostly mirrors and glass, deep bounce limits: the pooled end-of-frame passes, the
    exact routine for rays whose direction has drifted far from unit length, and the filter under many grazing hits."""
    rng = np.random.default_rng(seed)
    u = lambda lo, hi: float(rng.uniform(lo, hi))
    col = lambda: (u(0.4, 1.0), u(0.4, 1.0), u(0.4, 1.0))
    mats = [Material.Mirror(col(), u(0.6, 0.98)) for _ in range(5)] + [Material(color=col(), transparency=u(0.3, 0.9)) for _ in range(2)]
    mats += [Material.Diffuse(col(), 0.7, 0.1), Material.Emissive(col(), 1.5)]
    n = int(rng.integers(260, 700))
    sph = [scenes._sphere((0.0, -100.5, -1.0), 100.0, int(rng.integers(len(mats))))]
    for _ in range(n):
        sph.append(scenes._sphere((u(-4, 8), u(-0.3, 4), u(-8, 3)), u(0.1, 0.7), int(rng.integers(len(mats)))))
    sc = scenes.Scene(width=int(rng.integers(48, 100)), height=int(rng.integers(36, 70)), name=f"mirrors{seed}")
    sc.materials = scenes._materials(mats)
    sc.spheres = scenes._stack(sph, T.GPU_SPHERE)
    sc.lights = scenes._stack([scenes._light((u(-4, 8), u(3, 9), u(-4, 8)), u(0.5, 2.0)) for _ in range(int(rng.integers(1, 3)))], T.GPU_LIGHT)
    sc.camera_pos = (u(3, 7), u(0.5, 3), u(4, 8))
    sc.max_depth = int(rng.integers(6, 11))
    return sc


@pytest.mark.parametrize("seed", range(8))
def test_random_hall_of_mirrors_matches_the_oracle(oracle, seed):
    from path_tracer_amd import HipRenderer

    sc = random_hall_of_mirrors(2000 + seed)
    r = HipRenderer()
    assert r.initialize(sc.width, sc.height)
    r.setScene(sc, honour_max_depth=True)
    pc = sc.push_constants()
    r.render(pc)
    r.present()
    g8, gf = r.readbackRGBA8(), r.readbackF32()
    cnt = r.renderCounted(pc)
    r.shutdown()
    r8, rf, rc = oracle.render(sc, nthreads=16)
    np.testing.assert_array_equal(cnt[:NA], rc[:NA])
    c = dict(zip(T.COUNTER_NAMES, [int(x) for x in cnt]))
    d8 = np.abs(g8.astype(np.int16) - r8.astype(np.int16)).max(axis=-1)
    both_nan = np.isnan(gf) & np.isnan(rf)
    okf = ((np.abs(gf - rf) <= 1e-4 * np.maximum(np.abs(rf), 1e-2)) | both_nan).all(axis=-1)
    print(f"[{sc.name}] {len(sc.spheres)} spheres, depth {sc.max_depth}, exact rays {c['exact_rays']}, pooled passes {c['pooled_passes']}")
    assert float((d8 <= 1).mean()) >= 0.999 and float(okf.mean()) >= 0.999


@pytest.mark.parametrize("seed", range(12))
def test_random_scene_volume_modes_match_the_oracle(oracle, seed):
    """Random scenes that have a volume, drawn with the linear or the bricked voxel copy and with the nearest or
    the trilinear filter (the oracle states the same filter); two samples per pixel on the odd seeds."""
    from path_tracer_amd import HipRenderer

    base = 3000 + seed
    sc = random_scene(base)
    while len(sc.volumes) == 0:
        base += 100
        sc = random_scene(base)
    bricks, trilinear, samples = seed & 1, (seed >> 1) & 1, 1 + (seed % 3 == 1)
    r = HipRenderer()
    assert r.initialize(sc.width, sc.height)
    r.setOption(T.OPT_VOLUME_BRICKS, bricks)
    r.setOption(T.OPT_VOLUME_FILTER, trilinear)
    r.setScene(sc, honour_max_depth=True)
    pc = sc.push_constants()
    if samples > 1:
        r.renderSamples(pc, 0, samples)
    else:
        r.render(pc)
    r.present()
    g8, gf = r.readbackRGBA8(), r.readbackF32()
    r.shutdown()
    r8, rf, _ = oracle.render(sc, nthreads=8, volume_filter=trilinear, num_samples=samples)
    d8 = np.abs(g8.astype(np.int16) - r8.astype(np.int16)).max(axis=-1)
    both_nan = np.isnan(gf) & np.isnan(rf)
    okf = ((np.abs(gf - rf) <= 1e-4 * np.maximum(np.abs(rf), 1e-2)) | both_nan).all(axis=-1)
    assert float((d8 <= 1).mean()) >= 0.999 and float(okf.mean()) >= 0.999, (sc.name, bricks, trilinear, samples)


def random_volume_frame(seed: int) -> scenes.Scene:
    """A volume and no solids (the frames the volume kernel draws): extents that are powers of two (the march's trips
    without box tests) or not, boxes from a fraction of a unit to 16 units (the 500-step limit), cameras far off, in
    the plane of a face, on an edge line and inside the box, absorption up to where the sum overshoots."""
    rng = np.random.default_rng(seed)
    u = lambda lo, hi: float(rng.uniform(lo, hi))
    w, h = int(rng.integers(40, 200)), int(rng.integers(30, 120))
    sc = scenes.Scene(width=w, height=h, name=f"volume{seed}")
    pow2 = rng.random() < 0.7
    ext = tuple(float(rng.choice([0.5, 1.0, 2.0, 4.0, 16.0])) if pow2 else u(0.4, 5.0) for _ in range(3))
    # the camera looks at (2, 1.5, 0): keep the box around that point
    pos = (2.0 - ext[0] * u(0.2, 0.8), 1.5 - ext[1] * u(0.2, 0.8), 0.0 - ext[2] * u(0.2, 0.8))
    res = (int(rng.integers(3, 70)), int(rng.integers(3, 70)), int(rng.integers(3, 70)))
    sc.materials = scenes._materials([Material.Volumetric((u(0.1, 1), u(0.1, 1), u(0.1, 1)), u(0.2, 8.0))])
    sc.volumes = scenes._stack([scenes._volume(pos, (0, 0, 0), ext, res, 0)], T.GPU_VOLUME)
    sc.voxels = (scenes.synthetic_walnut(res) // int(rng.integers(1, 12))).astype(np.uint8)
    kind = int(rng.integers(5))
    if kind == 0:    # inside the box
        sc.camera_pos = tuple(pos[i] + ext[i] * u(0.1, 0.9) for i in range(3))
    elif kind == 1:  # in the plane of the top face
        sc.camera_pos = (2.0 + u(-6, 6), pos[1] + ext[1], u(3, 9))
    elif kind == 2:  # on the line of an edge
        sc.camera_pos = (pos[0] + ext[0], pos[1] + ext[1], u(3, 9))
    else:
        ang = u(0, 2 * np.pi)
        sc.camera_pos = (2.0 + u(2, 12) * np.cos(ang), u(-3, 6), u(2, 12) * np.sin(ang))
    sc.bg_top, sc.bg_bottom = (u(0, 1), u(0, 1), u(0, 1)), (u(0, 1), u(0, 1), u(0, 1))
    sc.max_depth = int(rng.integers(1, 11))
    return sc


@pytest.mark.parametrize("seed", range(40))
def test_random_volume_frame_matches_the_oracle(oracle, seed):
    """Bit for bit: no pow or cos on this path.  Bricked or linear voxels, nearest or trilinear filter by seed."""
    from path_tracer_amd import HipRenderer

    sc = random_volume_frame(5000 + seed)
    bricks, trilinear = seed & 1, int(seed % 5 == 3)
    r = HipRenderer()
    assert r.initialize(sc.width, sc.height)
    r.setOption(T.OPT_VOLUME_BRICKS, bricks)
    r.setOption(T.OPT_VOLUME_FILTER, trilinear)
    r.setScene(sc, honour_max_depth=True)
    pc = sc.push_constants()
    r.render(pc)
    r.present()
    gf = r.readbackF32()
    assert r.lastLaunch()[2] == 1024  # the volume kernel drew it
    cnt = r.renderCounted(pc)
    r.shutdown()
    r8, rf, rc = oracle.render(sc, nthreads=8, volume_filter=trilinear)
    np.testing.assert_array_equal(cnt[:NA], rc[:NA])
    same = (gf.view(np.uint32) == rf.view(np.uint32)) | (np.isnan(gf) & np.isnan(rf))
    assert same.all(), (sc.name, bricks, trilinear, float(same.all(axis=-1).mean()))
