"""The inequality behind the dense sphere loop's filter, checked in float32 arithmetic on the CPU.

The HIP kernel's packed loop may skip a sphere only if hitSphere's discriminant
(reference shaders/raytrace.comp:142-159, evaluated in float32 with the fused multiply-adds
the kernel and the oracle use) is negative.  The loop tests

    (o.d - c.d)^2 + 2 o.c - k  >=  |o|^2 (1 - 2^-17),
    k = |c|^2 - r^2 - 2^-17 (|c|^2 + r^2) - 1e-30   (rounded down to float32)

for rays with |d|^2 <= 1.25 (any shorter direction included) and coordinates up to 1e9.  Here both sides are evaluated
with the same operation order as the kernel on random and on adversarial inputs (grazing rays,
scenes far from the origin, tiny spheres): wherever the discriminant is >= 0 the filter must pass.
float32 fused multiply-add is emulated in float64 (the product of two float32 is exact there;
the sum is rounded to 53 bits before 24, which the filter's margin of 40+ ulps does not notice).
"""
import numpy as np

f32 = np.float32


def fma(a, b, c):
    return (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(f32)


def discriminant(o, d, c, r):
    oc = [(o[i] - c[i]).astype(f32) for i in range(3)]
    hb = fma(oc[2], d[2], fma(oc[1], d[1], (oc[0] * d[0]).astype(f32)))
    dd = fma(oc[2], oc[2], fma(oc[1], oc[1], (oc[0] * oc[0]).astype(f32)))
    cc = fma(-r, r, dd)
    return fma(hb, hb, -cc)


def filter_constant(c, r):
    c64 = [x.astype(np.float64) for x in c]
    c2 = c64[0] ** 2 + c64[1] ** 2 + c64[2] ** 2
    r2 = r.astype(np.float64) ** 2
    k = (c2 - r2 - 2.0 ** -17 * (c2 + r2) - 1.0e-30).astype(f32)
    return np.nextafter(k, f32(-np.inf))


def filter_passes(o, d, c, k):
    od = fma(o[2], d[2], fma(o[1], d[1], (o[0] * d[0]).astype(f32)))
    oo = fma(o[2], o[2], fma(o[1], o[1], (o[0] * o[0]).astype(f32)))
    thr = fma(f32(-2.0 ** -17) * np.ones_like(oo), oo, oo)
    hb = fma(c[0], -d[0], fma(c[1], -d[1], fma(c[2], -d[2], od)))
    m2o = [(f32(-2.0) * x).astype(f32) for x in o]
    oc2 = fma(c[0], m2o[0], fma(c[1], m2o[1], (c[2] * m2o[2]).astype(f32)))
    t = (fma(hb, hb, -oc2) - k).astype(f32)
    return t >= thr


def _rays_towards(rng, n, c, r, scale, graze):
    """Origins around the scene, directions aimed at a point at `graze` radii from the centre
    (1.0 = tangent), with |d|^2 anywhere in the accepted range."""
    o = [(rng.normal(size=n) * scale).astype(f32) + c[i] * f32(rng.random() < 0.5) for i in range(3)]
    u = rng.normal(size=(3, n))
    u /= np.linalg.norm(u, axis=0)
    to_c = np.stack([c[i].astype(np.float64) - o[i].astype(np.float64) for i in range(3)])
    dist = np.linalg.norm(to_c, axis=0) + 1e-30
    w = to_c / dist
    perp = u - (u * w).sum(axis=0) * w
    perp /= np.linalg.norm(perp, axis=0) + 1e-30
    target = to_c + perp * (r.astype(np.float64) * graze)
    dirn = target / (np.linalg.norm(target, axis=0) + 1e-30)
    length = np.sqrt(rng.uniform(0.0, 1.25, size=n) ** rng.choice([1.0, 4.0], size=n))  # also far shorter than 1
    d = [(dirn[i] * length).astype(f32) for i in range(3)]
    return o, d


def test_filter_never_rejects_a_sphere_the_shader_would_evaluate():
    rng = np.random.default_rng(20240611)
    n = 400_000
    checked = roots = passed = 0
    for scale, rlo, rhi in [(1.0, 0.05, 1.0), (30.0, 0.05, 0.5), (1.0e3, 1.0e-3, 10.0), (1.0e6, 1.0, 1.0e4),
                            (3.0e8, 1.0, 1.0e8), (1.0e-6, 1.0e-8, 1.0e-6), (1.0e-18, 1.0e-19, 1.0e-18)]:
        c = [(rng.normal(size=n) * scale).astype(f32) for _ in range(3)]
        r = rng.uniform(rlo, rhi, size=n).astype(f32)
        k = filter_constant(c, r)
        for graze in (0.0, 0.9, 0.999, 0.99999, 1.0, 1.00001, 1.001):
            o, d = _rays_towards(rng, n, c, r, scale, graze)
            dd = fma(d[2], d[2], fma(d[1], d[1], (d[0] * d[0]).astype(f32)))
            ok_ray = dd - f32(1.0) <= f32(0.25)  # shorter directions take the loop too
            disc = discriminant(o, d, c, r)
            need = ok_ray & (disc >= 0)
            got = filter_passes(o, d, c, k)
            assert not (need & ~got).any(), (scale, graze, int((need & ~got).sum()))
            checked += int(ok_ray.sum()); roots += int(need.sum()); passed += int((got & ok_ray).sum())
    assert roots > 0.1 * checked            # the adversarial part of the sample is real
    print(f"[sphere filter] {checked} ray-sphere pairs, {roots} with a root, {passed} let through")


def test_filter_is_selective_at_ordinary_scene_sizes():
    """With coordinates of a few tens and radii of 0.05-0.25 the filter lets through little more
    than the discriminant does (otherwise the renderer would pick the discriminant loop)."""
    rng = np.random.default_rng(7)
    n = 500_000
    c = [rng.uniform(-20, 14, size=n).astype(f32) for _ in range(3)]
    r = rng.uniform(0.05, 0.25, size=n).astype(f32)
    o = [np.full(n, v, f32) for v in (5.0, 1.5, 6.0)]
    u = rng.normal(size=(3, n)); u /= np.linalg.norm(u, axis=0)
    d = [u[i].astype(f32) for i in range(3)]
    disc = discriminant(o, d, c, r)
    got = filter_passes(o, d, c, filter_constant(c, r))
    assert not ((disc >= 0) & ~got).any()
    assert got.sum() <= 1.5 * (disc >= 0).sum() + 20
