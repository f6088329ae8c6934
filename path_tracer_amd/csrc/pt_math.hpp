// pt_math.hpp — float3 / packed-pair arithmetic of the device code, with the operation
// order the numerics contract fixes (see pt_device.hpp), and the compiled-in surface shaders.
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

// ---- surface shaders (compiled-in; index = position in the sorted path list + 1) ----

// rainbow.glsl:1-8
__device__ __forceinline__ float hue2rgb(float p, float q, float t) {
  if (t < 0.0f) t += 1.0f;
  if (t > 1.0f) t -= 1.0f;
  if (t < 1.0f / 6.0f) return p + ((q - p) * 6.0f) * t;
  if (t < 0.5f) return q;
  if (t < 2.0f / 3.0f) return p + ((q - p) * (2.0f / 3.0f - t)) * 6.0f;
  return p;
}
// rainbow.glsl:10-22
__device__ __forceinline__ V3 hsl2rgb(float h, float s, float l) {
  float H = glfract(h);
  float S = glclamp(s, 0.0f, 1.0f);
  float L = glclamp(l, 0.0f, 1.0f);
  if (S == 0.0f) return mk3(L, L, L);
  float q = (L < 0.5f) ? (L * (1.0f + S)) : ((L + S) - L * S);
  float p = 2.0f * L - q;
  return mk3(hue2rgb(p, q, H + 1.0f / 3.0f), hue2rgb(p, q, H), hue2rgb(p, q, H - 1.0f / 3.0f));
}
// rainbow.glsl:24-50 — returns result.emissive (result.color is 0)
__device__ __forceinline__ V3 rainbow_emissive(V3 hitPoint, float emissiveStrength, float time) {
  V3 sweepDir = normalize3(mk3(1.0f, 0.5f, 0.25f));
  float phase = dot3(hitPoint, sweepDir) * 2.0f + time * 0.002f;
  float cosVal = cosf(phase);
  float hue = glfract(0.5f + 0.5f * cosVal);
  return scale3(hsl2rgb(hue, 1.0f, 0.5f), emissiveStrength);
}

__device__ __forceinline__ uint32_t unorm8(float c) {
  if (c != c) return 0u;
  return (uint32_t)(int)__builtin_rintf(c * 255.0f);
}

} // namespace ptdev
