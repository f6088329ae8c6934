"""The inequality behind the packed ellipsoid loop's filter, checked in float32 arithmetic on the CPU.

The HIP kernel's wave-uniform ellipsoid loop only decides which (ray, ellipsoid) pairs go on to the
evaluation of hitEllipsoid proper (reference shaders/raytrace.comp:171-210); it may let through more
pairs than hitEllipsoid's discriminant would, never fewer.  It evaluates the same quadratic through one
3x3 matrix per ellipsoid, M = diag(1/radii) R(conj q) (computed in double precision on the host and
rounded once), and adds a margin that covers the rounding of both evaluations:

    y = M (o - c), z = M d, D' = (y.z)^2 - (z.z)(y.y - 1)
    candidate  iff  D' + |d|^2 (k1 |o - c|^2 + k2) >= -1e-30,
    k1 = (124 g + 98) u mu^4, k2 = (31 g + 29) u mu^2, u = 2^-24, mu >= |M|_2,
    g = max(1/radii) (1 + 2 |q||qw| + 2 |q|^2) / mu   (the size of the shader's intermediates against |M|)

Before the matrix test the loop runs a cheaper one on the bounding sphere: with sigma the smallest singular
value of M, Rb = 1/sigma, x = o - c:

    V = (x.d)^2 - |d|^2 (|x|^2 c1 - c2) >= 0,  c1 = 1 - Rb^4 k1 - 24 u (rounded down),
                                               c2 = Rb^2 (1 + Rb^2 k2) (1 + 8 u) + 1e-30 (rounded up)

which must also hold wherever the shader's discriminant is >= 0.

Here both sides are evaluated with the kernel's operation order on random and grazing rays against
random ellipsoids (anisotropy up to 8, unit and non-unit quaternions, scenes far from the origin):
wherever hitEllipsoid's float32 discriminant is >= 0 the filter must pass.  float32 fused multiply-add
is emulated in float64 (exact product, the sum rounded to 53 bits and then to 24).
"""
import numpy as np

f32 = np.float32
U = 2.0 ** -24


def fma(a, b, c):
    return (np.asarray(a, np.float64) * np.asarray(b, np.float64) + np.asarray(c, np.float64)).astype(f32)


def mul(a, b):
    return (a * b).astype(f32)


def dot3(a, b):
    return fma(a[2], b[2], fma(a[1], b[1], mul(a[0], b[0])))


def cross3(a, b):
    return [fma(a[1], b[2], -mul(a[2], b[1])), fma(a[2], b[0], -mul(a[0], b[2])), fma(a[0], b[1], -mul(a[1], b[0]))]


def rotate_q(v, q, q2, qw):
    t = cross3(q2, v)
    w = [fma(qw, t[i], v[i]) for i in range(3)]
    c = cross3(q, t)
    return [(w[i] + c[i]).astype(f32) for i in range(3)]


def shader_discriminant(o, d, ctr, radii, quat):
    """hitEllipsoid's discriminant, operation for operation (inverse rotation = conjugate quaternion)."""
    nq = [(-quat[i]).astype(f32) for i in range(3)]
    nq2 = [mul(f32(-2.0), quat[i]) for i in range(3)]
    x = [(o[i] - ctr[i]).astype(f32) for i in range(3)]
    oc = rotate_q(x, nq, nq2, quat[3])
    dr = rotate_q(d, nq, nq2, quat[3])
    inv = [(f32(1.0) / radii[i]).astype(f32) for i in range(3)]
    ys = [mul(oc[i], inv[i]) for i in range(3)]
    zs = [mul(dr[i], inv[i]) for i in range(3)]
    a = dot3(zs, zs)
    hb = dot3(ys, zs)
    c = (dot3(ys, ys) - f32(1.0)).astype(f32)
    return fma(hb, hb, -mul(a, c))


def host_constants(radii, quat):
    """M (float32), k1, k2 as the host computes them, in float64 from the float32 records."""
    q = [-quat[i].astype(np.float64) for i in range(3)]
    qw = quat[3].astype(np.float64)
    inv = [(f32(1.0) / radii[i]).astype(f32).astype(np.float64) for i in range(3)]
    n = len(qw)
    K = np.zeros((n, 3, 3))
    K[:, 0, 1], K[:, 0, 2] = -q[2], q[1]
    K[:, 1, 0], K[:, 1, 2] = q[2], -q[0]
    K[:, 2, 0], K[:, 2, 1] = -q[1], q[0]
    R = np.eye(3)[None] + 2.0 * qw[:, None, None] * K + 2.0 * (K @ K)
    M = np.stack(inv, axis=1)[:, :, None] * R
    qn = np.sqrt(q[0] ** 2 + q[1] ** 2 + q[2] ** 2)
    G = 1.0 + 2.0 * qn * np.abs(qw) + 2.0 * qn * qn          # bounds every intermediate of the shader's rotation
    sigma = np.max(np.abs(np.stack(inv, axis=1)), axis=1)
    MtM = np.abs(np.transpose(M, (0, 2, 1)) @ M)
    mu = np.sqrt(MtM.sum(axis=2).max(axis=1)) * (1.0 + 1e-12)  # >= the spectral norm of M
    g = sigma * G / mu
    k1 = np.nextafter(((124.0 * g + 98.0) * U * mu ** 4).astype(f32), f32(np.inf))
    k2 = np.nextafter(((31.0 * g + 29.0) * U * mu ** 2).astype(f32), f32(np.inf))
    return M.astype(f32), k1, k2


def sphere_constants(radii, quat):
    """c1 and -c2 of the bounding-sphere stage, as the host computes them."""
    M32, k1, k2 = host_constants(radii, quat)
    # the host takes sigma from the double-precision M, before rounding it
    q = [-quat[i].astype(np.float64) for i in range(3)]
    qw = quat[3].astype(np.float64)
    inv = [(f32(1.0) / radii[i]).astype(f32).astype(np.float64) for i in range(3)]
    n = len(qw)
    K = np.zeros((n, 3, 3))
    K[:, 0, 1], K[:, 0, 2] = -q[2], q[1]
    K[:, 1, 0], K[:, 1, 2] = q[2], -q[0]
    K[:, 2, 0], K[:, 2, 1] = -q[1], q[0]
    M = np.stack(inv, axis=1)[:, :, None] * (np.eye(3)[None] + 2.0 * qw[:, None, None] * K + 2.0 * (K @ K))
    lmin = np.linalg.eigvalsh(np.transpose(M, (0, 2, 1)) @ M)[:, 0] * (1.0 - 1e-9)
    rb2 = 1.0 / lmin
    b1 = rb2 * rb2 * k1.astype(np.float64)
    b2 = rb2 * (1.0 + rb2 * k2.astype(np.float64))
    c1 = np.nextafter((1.0 - b1 - 24.0 * U).astype(f32), f32(-np.inf))
    c2 = np.nextafter((b2 * (1.0 + 8.0 * U) + 1.0e-30).astype(f32), f32(np.inf))
    return c1, (-c2).astype(f32)


def sphere_value(o, d, ctr, c1, nc2):
    x = [(o[i] - ctr[i]).astype(f32) for i in range(3)]
    xd = dot3(x, d)
    xx = dot3(x, x)
    inner = fma(xx, c1, nc2)
    dd = dot3(d, d)
    return fma(xd, xd, -mul(dd, inner))


def filter_value(o, d, ctr, M, k1, k2):
    x = [(o[i] - ctr[i]).astype(f32) for i in range(3)]
    y = [fma(M[:, i, 2], x[2], fma(M[:, i, 1], x[1], mul(M[:, i, 0], x[0]))) for i in range(3)]
    z = [fma(M[:, i, 2], d[2], fma(M[:, i, 1], d[1], mul(M[:, i, 0], d[0]))) for i in range(3)]
    a = dot3(z, z)
    hb = dot3(y, z)
    c = fma(y[2], y[2], fma(y[1], y[1], fma(y[0], y[0], f32(-1.0))))
    disc = fma(hb, hb, -mul(a, c))
    dd = dot3(d, d)
    xx = dot3(x, x)
    return fma(mul(k1, dd), xx, fma(k2, dd, disc))


def _scene(rng, n, scale, ratio, qscale):
    ctr = [(rng.normal(size=n) * scale).astype(f32) for _ in range(3)]
    base = rng.uniform(0.1, 1.0, size=n)
    radii = [(base * rng.uniform(1.0, ratio, size=n)).astype(f32) for _ in range(3)]
    q = rng.normal(size=(4, n))
    q /= np.linalg.norm(q, axis=0)
    q *= rng.uniform(1.0 / qscale, qscale, size=n)
    quat = [q[i].astype(f32) for i in range(4)]
    return ctr, radii, quat


def _grazing_rays(rng, n, ctr, radii, quat, scale, graze):
    """Rays aimed at a point of the ellipsoid's surface scaled by `graze` about its centre, roughly
    tangentially (direction perpendicular to the surface normal there), from a distance."""
    p = rng.normal(size=(3, n)); p /= np.linalg.norm(p, axis=0)        # point on the unit sphere
    tdir = np.cross(p.T, rng.normal(size=(n, 3))).T; tdir /= np.linalg.norm(tdir, axis=0) + 1e-30
    r = np.stack([x.astype(np.float64) for x in radii])
    q = np.stack([x.astype(np.float64) for x in quat])
    qv, qw = q[:3], q[3]
    def rot(v):  # the shader's rotateByQuaternion with (q, qw), in double
        t = 2.0 * np.cross(qv.T, v.T).T
        return v + qw * t + np.cross(qv.T, t.T).T
    surf = rot(p * r * graze)                                           # local -> world
    tang = rot(tdir * r)
    tang /= np.linalg.norm(tang, axis=0) + 1e-30
    c = np.stack([x.astype(np.float64) for x in ctr])
    dist = rng.uniform(0.5, 30.0, size=n) * scale
    o = c + surf - tang * dist
    length = np.sqrt(rng.uniform(0.3, 3.0, size=n))
    return [o[i].astype(f32) for i in range(3)], [(tang[i] * length).astype(f32) for i in range(3)]


def test_filter_never_rejects_an_ellipsoid_the_shader_would_evaluate():
    rng = np.random.default_rng(20240612)
    n = 200_000
    checked = roots = passed = 0
    for scale, ratio, qscale in [(1.0, 1.0, 1.0), (5.0, 2.0, 1.0), (5.0, 4.0, 1.0), (30.0, 8.0, 1.0), (5.0, 3.0, 1.5),
                                 (300.0, 4.0, 1.0), (1.0e-3, 4.0, 1.0), (5.0, 4.0, 3.0)]:
        ctr, radii, quat = _scene(rng, n, scale, ratio, qscale)
        if scale < 1.0:
            radii = [(x * f32(scale)).astype(f32) for x in radii]
        M, k1, k2 = host_constants(radii, quat)
        c1, nc2 = sphere_constants(radii, quat)
        for graze in (0.0, 0.9, 0.999, 0.99999, 1.0, 1.00001, 1.001, 1.05):
            o, d = _grazing_rays(rng, n, ctr, radii, quat, max(scale, 1.0) if scale >= 1.0 else scale, graze)
            disc = shader_discriminant(o, d, ctr, radii, quat)
            val = filter_value(o, d, ctr, M, k1, k2)
            need = disc >= 0
            got = val >= f32(-1.0e-30)
            bad = need & ~got
            assert not bad.any(), (scale, ratio, qscale, graze, int(bad.sum()))
            bad1 = need & ~(sphere_value(o, d, ctr, c1, nc2) >= 0)
            assert not bad1.any(), ("bounding-sphere stage", scale, ratio, qscale, graze, int(bad1.sum()))
            checked += n; roots += int(need.sum()); passed += int(got.sum())
    assert roots > 0.2 * checked
    print(f"[ellipsoid filter] {checked} pairs, {roots} with a root, {passed} let through")


def test_filter_is_selective_for_ordinary_ellipsoids():
    """Radii of 0.15-0.6 per axis (anisotropy up to 4), unit quaternions, a scene a few units wide: the filter
    lets through little more than the discriminant does."""
    rng = np.random.default_rng(8)
    n = 400_000
    ctr = [rng.uniform(lo, hi, size=n).astype(f32) for lo, hi in ((-3, 7), (-0.2, 3), (-6, 3))]
    radii = [rng.uniform(0.15, 0.6, size=n).astype(f32) for _ in range(3)]
    q = rng.normal(size=(4, n)); q /= np.linalg.norm(q, axis=0)
    quat = [q[i].astype(f32) for i in range(4)]
    o = [np.full(n, v, f32) for v in (5.0, 1.5, 6.0)]
    tgt = np.stack([c.astype(np.float64) for c in ctr]) + rng.normal(size=(3, n)) * 0.8
    dv = tgt - np.stack([x.astype(np.float64) for x in o]); dv /= np.linalg.norm(dv, axis=0)
    d = [dv[i].astype(f32) for i in range(3)]
    M, k1, k2 = host_constants(radii, quat)
    disc = shader_discriminant(o, d, ctr, radii, quat)
    got = filter_value(o, d, ctr, M, k1, k2) >= f32(-1.0e-30)
    assert not ((disc >= 0) & ~got).any()
    c1, nc2 = sphere_constants(radii, quat)
    near = sphere_value(o, d, ctr, c1, nc2) >= 0
    assert not ((disc >= 0) & ~near).any()
    print(f"[ellipsoid filter] bounding-sphere stage lets through {int(near.sum())} of {n}")
    extra = int(got.sum()) - int((disc >= 0).sum())
    print(f"[ellipsoid filter] roots {int((disc >= 0).sum())}, extra candidates {extra}")
    assert extra <= 0.25 * (disc >= 0).sum()
