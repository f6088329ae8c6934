"""The dense sphere loop's conservative filter must not change anything.

The loop may test a cheaper expression than hitSphere's discriminant as long as it lets
through every sphere whose discriminant is non-negative; the candidates are then evaluated
with hitSphere's own operations (reference shaders/raytrace.comp:142-159).  Frames and
work counters with the filter forced on, forced off and against the oracle must agree,
also where the filter's margin is under stress: scenes far from the origin, grazing rays,
tiny and huge radii side by side.
"""
import numpy as np
import pytest

from path_tracer_amd import scenes
from path_tracer_amd import types as T

pytestmark = pytest.mark.gpu
NA = T.N_ALGORITHM_COUNTERS
TUNE = 100
NO_FILTER, FORCE_FILTER = 4, 8


def _frame(sc, tune):
    from path_tracer_amd import HipRenderer

    r = HipRenderer()
    assert r.initialize(sc.width, sc.height)
    r.setOption(TUNE, tune)  # read when the scene is uploaded
    r.setScene(sc, honour_max_depth=True)
    pc = sc.push_constants()
    r.render(pc)
    r.present()
    f = r.readbackF32().copy()
    cnt = r.renderCounted(pc)
    r.shutdown()
    return f, cnt


def _shifted(sc, shift):
    """The same scene moved away from the origin (camera target stays where the shader puts it,
    so the picture changes; what matters is the size of the coordinates)."""
    sc.spheres["center"] += np.asarray(shift, np.float32)
    sc.camera_pos = tuple(np.asarray(sc.camera_pos, np.float32) + np.asarray(shift, np.float32))
    return sc


def _cases():
    yield "c2_small", scenes.config2(320, 180, n_spheres=40, n_ellipsoids=4, max_depth=5)
    yield "c3_small", scenes.config3(256, 144, n_spheres=600, max_depth=4)
    yield "far_from_origin", _shifted(scenes.config3(192, 108, n_spheres=300, max_depth=3), (900.0, -40.0, 700.0))
    sc = scenes.config3(192, 108, n_spheres=300, max_depth=3)
    sc.spheres["radius"][5:60] *= 1.0e-3   # specks: the margin is far larger than r^2
    sc.spheres["radius"][60:70] *= 40.0    # and a few very large ones
    yield "specks_and_giants", sc
    sc = scenes.config2(256, 144, n_spheres=24, n_ellipsoids=0, max_depth=3)
    # a row of equal spheres seen edge-on: many grazing rays
    for i in range(12):
        sc.spheres["center"][i] = (-3.0 + 0.8 * i, 1.5, 0.0)
        sc.spheres["radius"][i] = 0.4
    yield "grazing_row", sc


@pytest.mark.parametrize("name,sc", list(_cases()), ids=[n for n, _ in _cases()])
def test_filter_on_off_and_oracle_agree(oracle, name, sc):
    f_on, c_on = _frame(sc, FORCE_FILTER)
    f_off, c_off = _frame(sc, NO_FILTER)
    np.testing.assert_array_equal(f_on.view(np.uint32), f_off.view(np.uint32))
    np.testing.assert_array_equal(c_on[:NA], c_off[:NA])
    _, _, rc = oracle.render(sc, nthreads=16)
    np.testing.assert_array_equal(c_on[:NA], rc[:NA])  # same rays, same roots as the shader's loop


def _ellipsoid_cases():
    """Scenes that stress the packed ellipsoid loop's candidate filter (a cheaper evaluation of
    hitEllipsoid's quadratic plus a margin for the rounding of both): strong anisotropy, quaternions
    that are not unit, ellipsoids far from the origin, many grazing rays."""
    rng = np.random.default_rng(5)
    sc = scenes.config2(256, 144, n_spheres=6, n_ellipsoids=24, max_depth=4)
    yield "c2_many_ellipsoids", sc
    sc = scenes.config2(224, 126, n_spheres=4, n_ellipsoids=16, max_depth=3)
    sc.ellipsoids["radii"][:, 0] *= 0.12   # needles and discs: anisotropy up to ~30
    sc.ellipsoids["radii"][8:, 1] *= 3.0
    yield "needles_and_discs", sc
    sc = scenes.config2(224, 126, n_spheres=4, n_ellipsoids=16, max_depth=3)
    sc.ellipsoids["rotation"] *= rng.uniform(0.4, 2.5, size=(16, 1)).astype(np.float32)  # not rotations: the shader's map as is
    yield "non_unit_quaternions", sc
    sc = scenes.config2(192, 108, n_spheres=4, n_ellipsoids=12, max_depth=3)
    shift = np.asarray((600.0, -30.0, 450.0), np.float32)
    sc.spheres["center"] += shift
    sc.ellipsoids["center"] += shift
    sc.camera_pos = tuple(np.asarray(sc.camera_pos, np.float32) + shift)
    yield "ellipsoids_far_from_origin", sc
    sc = scenes.config2(256, 144, n_spheres=2, n_ellipsoids=14, max_depth=3)
    for i in range(12):                      # a row of equal flat ellipsoids seen edge-on
        sc.ellipsoids["center"][i] = (-3.0 + 0.8 * i, 1.5, 0.0)
        sc.ellipsoids["radii"][i] = (0.45, 0.05, 0.3)
        sc.ellipsoids["rotation"][i] = (0.0, 0.0, 0.0, 1.0)
    yield "grazing_discs", sc


@pytest.mark.parametrize("name,sc", list(_ellipsoid_cases()), ids=[n for n, _ in _ellipsoid_cases()])
def test_ellipsoid_filter_keeps_every_root(oracle, name, sc):
    f_gpu, cnt = _frame(sc, 0)
    r8, rf, rc = oracle.render(sc, nthreads=16)
    np.testing.assert_array_equal(cnt[:NA], rc[:NA])  # same rays, same roots as hitEllipsoid in the shader's loop
    err = np.abs(f_gpu - rf)
    ok = (err <= 1e-4 * np.maximum(np.abs(rf), 1e-2)) | (np.isnan(f_gpu) & np.isnan(rf))
    frac = float(ok.all(axis=-1).mean())
    print(f"[{name}] float<=1e-4 {frac:.6f}")
    assert frac >= 0.999
