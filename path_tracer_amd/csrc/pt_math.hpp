// pt_math.hpp — float3 / packed-pair arithmetic of the device code, with the operation
// order the numerics contract fixes (see pt_device.hpp).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ptdev {

struct V3 { float x, y, z; };

__device__ __forceinline__ V3 mk3(float x, float y, float z) { V3 r; r.x = x; r.y = y; r.z = z; return r; }
__device__ __forceinline__ V3 ld3(const float *p) { return mk3(p[0], p[1], p[2]); }
__device__ __forceinline__ V3 xyz(float4 v) { return mk3(v.x, v.y, v.z); }
__device__ __forceinline__ V3 add3(V3 a, V3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ V3 sub3(V3 a, V3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ V3 mul3(V3 a, V3 b) { return mk3(a.x * b.x, a.y * b.y, a.z * b.z); }
__device__ __forceinline__ V3 scale3(V3 a, float s) { return mk3(a.x * s, a.y * s, a.z * s); }
__device__ __forceinline__ V3 div3s(V3 a, float s) { return mk3(a.x / s, a.y / s, a.z / s); }
__device__ __forceinline__ V3 neg3(V3 a) { return mk3(-a.x, -a.y, -a.z); }
__device__ __forceinline__ float dot3(V3 a, V3 b) {
  return __builtin_fmaf(a.z, b.z, __builtin_fmaf(a.y, b.y, a.x * b.x));
}
__device__ __forceinline__ V3 cross3(V3 a, V3 b) {
  return mk3(__builtin_fmaf(a.y, b.z, -(a.z * b.y)), __builtin_fmaf(a.z, b.x, -(a.x * b.z)),
             __builtin_fmaf(a.x, b.y, -(a.y * b.x)));
}
__device__ __forceinline__ V3 madd3(V3 a, float s, V3 b) {
  return mk3(__builtin_fmaf(s, b.x, a.x), __builtin_fmaf(s, b.y, a.y), __builtin_fmaf(s, b.z, a.z));
}
__device__ __forceinline__ float length3(V3 a) { return __builtin_sqrtf(dot3(a, a)); }
__device__ __forceinline__ V3 normalize3(V3 a) {
  float inv = 1.0f / length3(a);
  return scale3(a, inv);
}
// GLSL max/min/clamp: max(x,y) = (x<y)?y:x ; min(x,y) = (y<x)?y:x (NaN in x propagates)
__device__ __forceinline__ float glmax(float x, float y) { return (x < y) ? y : x; }
__device__ __forceinline__ float glmin(float x, float y) { return (y < x) ? y : x; }
__device__ __forceinline__ float glclamp(float x, float lo, float hi) { return glmin(glmax(x, lo), hi); }
__device__ __forceinline__ float glfract(float x) { return x - __builtin_floorf(x); }

// Two-wide float math: one v_pk_* instruction works on two spheres at once.
typedef float f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f2 mk2(float a, float b) { f2 r; r.x = a; r.y = b; return r; }
__device__ __forceinline__ f2 fma2(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }

// Three-vectors of pairs: the same operations as V3, on two objects at once.
struct P3 { f2 x, y, z; };
__device__ __forceinline__ P3 mkp3(f2 x, f2 y, f2 z) { P3 r; r.x = x; r.y = y; r.z = z; return r; }
__device__ __forceinline__ P3 splat3(V3 v) { return mkp3(mk2(v.x, v.x), mk2(v.y, v.y), mk2(v.z, v.z)); }
__device__ __forceinline__ P3 add3p(P3 a, P3 b) { return mkp3(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ P3 sub3p(P3 a, P3 b) { return mkp3(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ P3 mul3p(P3 a, P3 b) { return mkp3(a.x * b.x, a.y * b.y, a.z * b.z); }
__device__ __forceinline__ f2 dot3p(P3 a, P3 b) { return fma2(a.z, b.z, fma2(a.y, b.y, a.x * b.x)); }
__device__ __forceinline__ P3 cross3p(P3 a, P3 b) {
  return mkp3(fma2(a.y, b.z, -(a.z * b.y)), fma2(a.z, b.x, -(a.x * b.z)), fma2(a.x, b.y, -(a.y * b.x)));
}
__device__ __forceinline__ P3 madd3p(P3 a, f2 s, P3 b) { return mkp3(fma2(s, b.x, a.x), fma2(s, b.y, a.y), fma2(s, b.z, a.z)); }
__device__ __forceinline__ P3 rotate_qp(P3 v, P3 q, P3 q2, f2 qw) {
  P3 t = cross3p(q2, v);
  return add3p(madd3p(v, qw, t), cross3p(q, t));
}

// rotateByQuaternion with q.xyz and 2*q.xyz supplied (raytrace.comp:99-108):
// t = 2*cross(q, v) == cross(2q, v) exactly; v' = (v + qw*t) + cross(q, t)
__device__ __forceinline__ V3 rotate_q(V3 v, V3 q, V3 q2, float qw) {
  V3 t = cross3(q2, v);
  return add3(madd3(v, qw, t), cross3(q, t));
}

// pow(x, y) as GLSL's pow in calculateLighting and the toon shader (raytrace.comp:357, toon.glsl:33),
// rounded like a correctly rounded powf (which the CPU restatement's libm is in all but a few of 10^4
// cases).  The device library's generic powf costs about 170 instructions and ran once per pass; the
// highlight only ever raises max(0, N.H) to the material's shininess, so the common cases are taken in
// double precision, where x^y = 2^(y log2 x) needs no extended-precision tricks:
//   log2(m 2^e) = e + s P(s^2), s = (m - 1)/(m + 1), m in [sqrt(1/2), sqrt(2)), P of degree 7 (error 2^-39);
//   2^t = 2^n 2^f, n = rint(t), 2^f a degree-10 polynomial on [-1/2, 1/2] (error 2^-42);
// one rounding to float at the end.  Finite x > 0 with finite y take it, x = 0 with y > 0 gives 0,
// anything else (zero or negative exponents of zero, infinities, NaNs, negative x) the generic routine.
__device__ __attribute__((noinline)) float highlight_pow(float x, float y) {
  const bool finiteY = __builtin_fabsf(y) < __builtin_inff();
  const bool fast = x > 0.0f && x < __builtin_inff() && finiteY;
  const bool zero = x == 0.0f && y > 0.0f;
  float res = 0.0f;
  // Whole exponents up to 4096 (a shininess is 16, 32, 128 ...): square and multiply in double precision.  Each of the
  // at most 24 products adds a relative error of 2^-53, which leaves the rounded float the correctly rounded power
  // except within 3e-15 of a rounding boundary (the polynomial route below: 2e-12).  Taken when every lane that
  // has a power to compute can take it.
  const float yRound = __builtin_rintf(y);
  const bool whole = fast && y == yRound && y >= 1.0f && y <= 4096.0f;
  if (__builtin_amdgcn_ballot_w64(fast && !whole) == 0ull) {
    if (whole) {
      double b = (double)x, r = 1.0;
      unsigned int n = (unsigned int)y;
      while (n > 1u) {
        if (n & 1u) r *= b;
        b *= b;
        n >>= 1;
      }
      res = (float)(r * b);
    } else if (!zero && !fast) {
      res = powf(x, y);
    }
    return res;
  }
  if (fast) {
    const double xd = (double)x;
    double m = __builtin_amdgcn_frexp_mant(xd); // [1/2, 1)
    int e = __builtin_amdgcn_frexp_exp(xd);
    if (m < 0x1.6a09e667f3bcdp-1) { m *= 2.0; e -= 1; }
    const double a = m - 1.0, b = m + 1.0;
    double r = __builtin_amdgcn_rcp(b);
    r = __builtin_fma(__builtin_fma(-b, r, 1.0), r, r);
    r = __builtin_fma(__builtin_fma(-b, r, 1.0), r, r);
    const double s = a * r, s2 = s * s;
    const double lc[8] = {0x1.71547652b82fep+1, 0x1.ec709dc3a03fdp-1, 0x1.2776c50ef9bfep-1, 0x1.a61762a7aded9p-2, 0x1.484b13d7c02a9p-2, 0x1.0c9a84994022dp-2, 0x1.c68f568d31760p-3, 0x1.89f3b1694cffep-3};
    double p = lc[7];
#pragma unroll
    for (int k = 6; k >= 0; --k) p = __builtin_fma(p, s2, lc[k]);
    const double l2 = __builtin_fma(s, p, (double)e);
    double t = (double)y * l2;
    t = __builtin_fmin(__builtin_fmax(t, -1100.0), 1100.0); // beyond: 0 or infinity either way
    const double n = __builtin_rint(t), f = t - n;
    const double ec[11] = {0x1.0000000000000p+0, 0x1.62e42fefa39efp-1, 0x1.ebfbdff82c58ep-3, 0x1.c6b08d704a0bfp-5, 0x1.3b2ab6fba4e77p-7, 0x1.5d87fe78a6730p-10, 0x1.430912f86c786p-13, 0x1.ffcbfc588b0c5p-17, 0x1.62c0223a5c822p-20, 0x1.b5253d395e7c1p-24, 0x1.e4cf5158b8ec7p-28};
    double q = ec[10];
#pragma unroll
    for (int k = 9; k >= 0; --k) q = __builtin_fma(q, f, ec[k]);
    res = (float)__builtin_ldexp(q, (int)n);
  } else if (!zero) {
    res = powf(x, y);
  }
  return res;
}

__device__ __forceinline__ uint32_t unorm8(float c) {
  if (c != c) return 0u;
  return (uint32_t)(int)__builtin_rintf(c * 255.0f);
}

} // namespace ptdev
