"""The test that lets a lane run over emissive entries without evaluating them, checked in float32 on the CPU.

calculateLighting skips an emissive object unless alignment = dot(N, (c - p) / |c - p|) > 0 (reference
shaders/raytrace.comp:417-421 for spheres, 452-455 for ellipsoids).  The HIP kernel's lanes first pass over the entries
for which

    dot(N, c - p) < -(2^-20 * sum |N_i (c - p)_i| + 1e-30)

and evaluate the shader's own expression only for the others.  The shortcut may never skip an entry whose alignment,
computed in float32 exactly as the shader (and the oracle) compute it, is positive.  Both sides are evaluated here with
the kernel's operation order (dot = fma(z, z', fma(y, y', x x')), vec / scalar = one IEEE division per component) on
random hit points and on adversarial ones: normals almost perpendicular to the direction, objects very near and very
far, coordinates of very different magnitudes.  float32 fused multiply-add is emulated in float64 (the product of two
float32 is exact there; the sum is rounded to 53 bits before 24, far below the margin the test leaves).
"""
import numpy as np

f32 = np.float32


def fma(a, b, c):
    return (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(f32)


def dot3(a, b):
    return fma(a[2], b[2], fma(a[1], b[1], (a[0] * b[0]).astype(f32)))


def shader_alignment(n, c, p):
    t = [(c[i] - p[i]).astype(f32) for i in range(3)]
    dist = np.sqrt(dot3(t, t).astype(np.float64)).astype(f32)  # float32 sqrt is correctly rounded
    with np.errstate(divide="ignore", invalid="ignore"):
        d = [(t[i] / dist).astype(f32) for i in range(3)]
    return dot3(n, d)


def turned_away(n, c, p):
    t = [(c[i] - p[i]).astype(f32) for i in range(3)]
    a = dot3(n, t)
    s = fma(np.abs(n[2]), np.abs(t[2]), fma(np.abs(n[1]), np.abs(t[1]), (np.abs(n[0]) * np.abs(t[0])).astype(f32)))
    return fma(np.full_like(s, f32(2.0 ** -20)), s, a) < f32(-1.0e-30)


def _unit(rng, n):
    v = rng.normal(size=(3, n))
    return v / np.linalg.norm(v, axis=0)


def _check(n, c, p):
    skip = turned_away(n, c, p)
    al = shader_alignment(n, c, p)
    bad = skip & (al > 0)
    assert not bad.any(), (int(bad.sum()), [x[bad][:3] for x in n], [x[bad][:3] for x in c], [x[bad][:3] for x in p])
    return skip, al


def test_skipped_entries_never_face_the_hit_point():
    rng = np.random.default_rng(7)
    m = 2_000_000
    skipped = facing = 0
    for scale_p, scale_c in ((1.0, 1.0), (1e-3, 1.0), (1e3, 1e3), (1e6, 1.0), (1.0, 1e-6), (3e8, 3e8)):
        n = [x.astype(f32) for x in _unit(rng, m) * rng.choice([1.0, 1.0, 0.999, 1.001, 1e-3, 40.0], m)]
        p = [(rng.normal(size=m) * scale_p).astype(f32) for _ in range(3)]
        c = [(rng.normal(size=m) * scale_c).astype(f32) + p[i] * f32(rng.random() < 0.5) for i in range(3)]
        skip, al = _check(n, c, p)
        skipped += int(skip.sum())
        facing += int((al > 0).sum())
    # the shortcut is worth something: it rules out nearly every entry that faces away
    assert skipped > 0.45 * 6 * m and facing > 0.45 * 6 * m


def test_normals_almost_perpendicular_to_the_direction():
    rng = np.random.default_rng(8)
    m = 1_000_000
    for tilt in (0.0, 1e-8, 1e-7, 3e-7, 1e-6, 1e-5, 1e-3):
        u = _unit(rng, m)
        w = _unit(rng, m)
        w -= (w * u).sum(axis=0) * u  # perpendicular to u
        w /= np.linalg.norm(w, axis=0)
        nrm = w + tilt * rng.choice([-1.0, 1.0], m) * u
        dist = 10.0 ** rng.uniform(-4, 5, m)
        p = [(rng.normal(size=m) * 10.0 ** rng.uniform(-2, 4)).astype(f32) for _ in range(3)]
        c = [(p[i].astype(np.float64) + u[i] * dist).astype(f32) for i in range(3)]
        n = [nrm[i].astype(f32) for i in range(3)]
        _check(n, c, p)


def test_degenerate_inputs_are_left_to_the_shader():
    z = np.zeros(4, dtype=f32)
    one = np.ones(4, dtype=f32)
    nan = np.full(4, np.nan, dtype=f32)
    inf = np.full(4, np.inf, dtype=f32)
    # the object at the hit point, a zero normal, NaN and infinite coordinates: never skipped
    for n, c, p in (([one, z, z], [z, z, z], [z, z, z]), ([z, z, z], [one, one, one], [z, z, z]),
                    ([nan, z, z], [one, z, z], [z, z, z]), ([one, z, z], [nan, z, z], [z, z, z]),
                    ([-one, z, z], [inf, z, z], [z, z, z]), ([one, z, z], [z, z, z], [nan, nan, nan])):
        with np.errstate(invalid="ignore"):
            assert not turned_away(n, c, p).any()
