"""The CPU oracle against vectors produced by the reference's own classes.

tests/golden/ref_kat.json was written by oracle/_ref/ref_kat (source:
oracle/ref_kat.cpp), which links the reference's legacy double-precision
Sphere::hit (reference src/Sphere.cpp:9-46), Quaternion::rotateVector
(reference src/Quaternion.cpp:178-186) and Camera::getRay
(reference src/Camera.cpp:38-41).  They are not the render path itself (the
GLSL cannot run here), but they are the same formulas in double, so the
float32 oracle must agree to float rounding.
"""
import json
import os

import numpy as np
import pytest

GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "ref_kat.json")


@pytest.fixture(scope="module")
def kat():
    with open(GOLDEN) as f:
        return json.load(f)


def test_sphere_hit_matches_reference_sphere_hit(oracle, kat):
    n_hit = 0
    for v in kat["sphere_hit"]:
        hit, t, p, n = oracle.hit_sphere(v["origin"], v["dir"], v["center"], v["radius"], v["tmin"], v["tmax"])
        assert hit == bool(v["hit"]), v
        if hit:
            n_hit += 1
            # t = -hb -/+ sqrt(disc): a float32 rounding error of ~2^-24 * (|oc|^2 + hb^2)
            # in disc is amplified by 1 / (2 sqrt(disc)) near tangency.
            oc = np.array(v["origin"]) - np.array(v["center"])
            d = np.array(v["dir"])
            hb = oc @ d
            disc = hb * hb - (d @ d) * (oc @ oc - v["radius"] ** 2)
            tol = 1e-5 + 4e-7 * (oc @ oc + hb * hb) / max(np.sqrt(disc), 1e-3)
            assert abs(t - v["t"]) <= tol, (t, v["t"], tol)
            np.testing.assert_allclose(p, v["point"], rtol=0, atol=tol + 2e-6)
            np.testing.assert_allclose(n, v["normal"], rtol=0, atol=(tol + 2e-6) / v["radius"])
    assert n_hit > 64


def test_quaternion_rotation_matches_reference_rotate_vector(oracle, kat):
    for v in kat["quat_rotate"]:
        r = oracle.rotate(v["v"], v["q_xyzw"])
        ri = oracle.rotate(v["v"], v["q_xyzw"], inverse=True)
        np.testing.assert_allclose(r, v["rotated"], rtol=0, atol=5e-6)
        np.testing.assert_allclose(ri, v["inv_rotated"], rtol=0, atol=5e-6)


def test_camera_ray_matches_reference_camera_at_half_extent(oracle, kat):
    for v in kat["camera_ray"]:
        o, d = oracle.camera_ray(v["eye"], v["width"], v["height"], v["px"], v["py"])
        np.testing.assert_array_equal(o, np.array(v["eye"], np.float32))
        np.testing.assert_allclose(d, v["dir"], rtol=0, atol=1e-6)
