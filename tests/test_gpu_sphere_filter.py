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
    r.setScene(sc)
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
