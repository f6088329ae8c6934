// loop_bench.hip — isolates the sphere-discriminant loop of the renderer:
// 4 spheres per trip from an LDS image, two per packed instruction.
// Variants: 0 = LDS reads + math (the renderer's fast path)
//           1 = math only (sphere data held in registers)
//           2 = LDS reads only
//           3 = LDS + math, software-pipelined (next trip's reads issued before this trip's math)
//           4 = scalar (non-packed) math from LDS
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef float f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f2 mk2(float a, float b) { f2 r; r.x = a; r.y = b; return r; }
__device__ __forceinline__ f2 fma2(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }

#define PAIR_DISC(A, B, DISC) { \
  const f2 ocx = OX - mk2(A.x, A.y), ocy = OY - mk2(A.z, A.w), ocz = OZ - mk2(B.x, B.y); \
  const f2 rr = mk2(B.z, B.w); \
  const f2 hb = fma2(ocz, DZ, fma2(ocy, DY, ocx * DX)); \
  const f2 dd = fma2(ocz, ocz, fma2(ocy, ocy, ocx * ocx)); \
  const f2 cc = fma2(-rr, rr, dd); \
  DISC = fma2(hb, hb, -cc); }

template <int V>
__global__ void __launch_bounds__(1024) loop_kernel(const float4 *pairs, int nGroups, int passes, int *out) {
  extern __shared__ float4 s_obj[];
  for (int i = threadIdx.x; i < 4 * nGroups; i += blockDim.x) s_obj[i] = pairs[i];
  __syncthreads();
  const float ox = 5.0f + 0.001f * threadIdx.x, oy = 1.5f, oz = 6.0f + 0.01f * blockIdx.x;
  const float dx = -0.3f, dy = 0.01f * (threadIdx.x & 63), dz = -0.9f;
  const f2 OX = mk2(ox, ox), OY = mk2(oy, oy), OZ = mk2(oz, oz), DX = mk2(dx, dx), DY = mk2(dy, dy), DZ = mk2(dz, dz);
  int hits = 0;
  for (int p = 0; p < passes; ++p) {
    if (V == 3) {
      float4 a0 = s_obj[0], b0 = s_obj[1], a1 = s_obj[2], b1 = s_obj[3];
      for (int g = 0; g < nGroups; ++g) {
        const int gn = (g + 1 < nGroups) ? g + 1 : g;
        const float4 na0 = s_obj[4 * gn], nb0 = s_obj[4 * gn + 1], na1 = s_obj[4 * gn + 2], nb1 = s_obj[4 * gn + 3];
        f2 d0, d1;
        PAIR_DISC(a0, b0, d0) PAIR_DISC(a1, b1, d1)
        const float m = __builtin_fmaxf(__builtin_fmaxf(d0.x, d0.y), __builtin_fmaxf(d1.x, d1.y));
        if (__ballot(m >= 0.0f) != 0ull) hits++;
        a0 = na0; b0 = nb0; a1 = na1; b1 = nb1;
      }
    } else if (V == 5) { // ping-pong prefetch: no register copies (nGroups even)
      float4 a0 = s_obj[0], b0 = s_obj[1], a1 = s_obj[2], b1 = s_obj[3];
      for (int g = 0; g < nGroups; g += 2) {
        const float4 c0 = s_obj[4 * g + 4], e0 = s_obj[4 * g + 5], c1 = s_obj[4 * g + 6], e1 = s_obj[4 * g + 7];
        {
          f2 d0, d1;
          PAIR_DISC(a0, b0, d0) PAIR_DISC(a1, b1, d1)
          const float m = __builtin_fmaxf(__builtin_fmaxf(d0.x, d0.y), __builtin_fmaxf(d1.x, d1.y));
          if (__ballot(m >= 0.0f) != 0ull) hits++;
        }
        const int gn = (g + 2 < nGroups) ? g + 2 : 0;
        a0 = s_obj[4 * gn]; b0 = s_obj[4 * gn + 1]; a1 = s_obj[4 * gn + 2]; b1 = s_obj[4 * gn + 3];
        {
          f2 d0, d1;
          PAIR_DISC(c0, e0, d0) PAIR_DISC(c1, e1, d1)
          const float m = __builtin_fmaxf(__builtin_fmaxf(d0.x, d0.y), __builtin_fmaxf(d1.x, d1.y));
          if (__ballot(m >= 0.0f) != 0ull) hits++;
        }
      }
    } else if (V == 6) { // 8 spheres per trip, one sign test per trip
      for (int g = 0; g < nGroups; g += 2) {
        const float4 a0 = s_obj[4 * g], b0 = s_obj[4 * g + 1], a1 = s_obj[4 * g + 2], b1 = s_obj[4 * g + 3];
        const float4 c0 = s_obj[4 * g + 4], e0 = s_obj[4 * g + 5], c1 = s_obj[4 * g + 6], e1 = s_obj[4 * g + 7];
        f2 d0, d1, d2, d3;
        PAIR_DISC(a0, b0, d0) PAIR_DISC(a1, b1, d1) PAIR_DISC(c0, e0, d2) PAIR_DISC(c1, e1, d3)
        const float m = __builtin_fmaxf(__builtin_fmaxf(__builtin_fmaxf(d0.x, d0.y), __builtin_fmaxf(d1.x, d1.y)),
                                        __builtin_fmaxf(__builtin_fmaxf(d2.x, d2.y), __builtin_fmaxf(d3.x, d3.y)));
        if (__ballot(m >= 0.0f) != 0ull) hits++;
      }
    } else if (V == 7) { // two rays per lane: sphere data read once for both
      const f2 PX = mk2(ox + 0.5f, ox + 0.5f), PY = mk2(oy + 0.25f, oy + 0.25f), PZ = OZ;
      const f2 EX = mk2(dx, dx), EY = mk2(dy + 0.1f, dy + 0.1f), EZ = DZ;
      for (int g = 0; g < nGroups; ++g) {
        const float4 a0 = s_obj[4 * g], b0 = s_obj[4 * g + 1], a1 = s_obj[4 * g + 2], b1 = s_obj[4 * g + 3];
        f2 d0, d1, d2, d3;
        PAIR_DISC(a0, b0, d0) PAIR_DISC(a1, b1, d1)
        {
          const f2 OX = PX, OY = PY, OZ = PZ, DX = EX, DY = EY, DZ = EZ;
          PAIR_DISC(a0, b0, d2) PAIR_DISC(a1, b1, d3)
        }
        const float m = __builtin_fmaxf(__builtin_fmaxf(__builtin_fmaxf(d0.x, d0.y), __builtin_fmaxf(d1.x, d1.y)),
                                        __builtin_fmaxf(__builtin_fmaxf(d2.x, d2.y), __builtin_fmaxf(d3.x, d3.y)));
        if (__ballot(m >= 0.0f) != 0ull) hits++;
      }
    } else if (V == 4) {
      for (int g = 0; g < nGroups; ++g) {
        float m = -1.0f;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float4 s = s_obj[4 * g + k];
          const float ocx = ox - s.x, ocy = oy - s.y, ocz = oz - s.z;
          const float hb = __builtin_fmaf(ocz, dz, __builtin_fmaf(ocy, dy, ocx * dx));
          const float dd = __builtin_fmaf(ocz, ocz, __builtin_fmaf(ocy, ocy, ocx * ocx));
          const float cc = __builtin_fmaf(-s.w, s.w, dd);
          m = __builtin_fmaxf(m, __builtin_fmaf(hb, hb, -cc));
        }
        if (__ballot(m >= 0.0f) != 0ull) hits++;
      }
    } else {
      float4 a0 = s_obj[0], b0 = s_obj[1], a1 = s_obj[2], b1 = s_obj[3];
      for (int g = 0; g < nGroups; ++g) {
        if (V == 0 || V == 2) { a0 = s_obj[4 * g]; b0 = s_obj[4 * g + 1]; a1 = s_obj[4 * g + 2]; b1 = s_obj[4 * g + 3]; }
        if (V == 2) {
          asm volatile("" :: "v"(a0.x), "v"(b0.x), "v"(a1.x), "v"(b1.x));
          continue;
        }
        if (V == 1) asm volatile("" : "+v"(a0.x), "+v"(b0.x), "+v"(a1.x), "+v"(b1.x));
        f2 d0, d1;
        PAIR_DISC(a0, b0, d0) PAIR_DISC(a1, b1, d1)
        const float m = __builtin_fmaxf(__builtin_fmaxf(d0.x, d0.y), __builtin_fmaxf(d1.x, d1.y));
        if (__ballot(m >= 0.0f) != 0ull) hits++;
      }
    }
  }
  if (hits == 0x7fffffff) out[0] = hits;
}

template <int V>
int run(const char *name, const float4 *dPairs, int nGroups, int cus, int *dOut) {
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  size_t lds = (size_t)nGroups * 64;
  CHECK(hipFuncSetAttribute((const void *)loop_kernel<V>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  for (int threads : {256, 512, 1024}) {
    const int passes = 40;
    hipLaunchKernelGGL(loop_kernel<V>, dim3(cus), dim3(threads), lds, 0, dPairs, nGroups, 2, dOut);
    CHECK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
      CHECK(hipEventRecord(e0));
      hipLaunchKernelGGL(loop_kernel<V>, dim3(cus), dim3(threads), lds, 0, dPairs, nGroups, passes, dOut);
      CHECK(hipEventRecord(e1));
      CHECK(hipEventSynchronize(e1));
      float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
      if (ms < best) best = ms;
    }
    double trips = (double)passes * nGroups;            // per wave
    double wavesPerSimd = threads / 256.0;
    double cyc = best * 1e-3 * 2.4e9 / (trips * wavesPerSimd);
    double tests = (double)cus * threads * passes * nGroups * 4.0;
    printf("%-28s waves/SIMD %.0f  %8.3f ms  %6.1f cycles/trip/SIMD-share  %7.2f Gtests/s  alg %6.2f TFLOP/s\n", name, wavesPerSimd,
           best, cyc, tests / (best * 1e-3) / 1e9, tests * 17 / (best * 1e-3) / 1e12);
  }
  return 0;
}

int main() {
  hipDeviceProp_t p; CHECK(hipGetDeviceProperties(&p, 0));
  const int nGroups = 1024;
  std::vector<float4> h(nGroups * 4);
  for (size_t i = 0; i < h.size(); ++i) h[i] = make_float4(50.f + i % 7, 60.f + i % 5, -40.f - i % 3, 0.1f + 0.001f * (i % 11));
  float4 *d; int *o;
  CHECK(hipMalloc(&d, h.size() * 16)); CHECK(hipMalloc(&o, 4));
  CHECK(hipMemcpy(d, h.data(), h.size() * 16, hipMemcpyHostToDevice));
  if (run<0>("LDS + packed math", d, nGroups, p.multiProcessorCount, o)) return 1;
  if (run<1>("packed math only", d, nGroups, p.multiProcessorCount, o)) return 1;
  if (run<2>("LDS reads only", d, nGroups, p.multiProcessorCount, o)) return 1;
  if (run<3>("LDS + packed, prefetch", d, nGroups, p.multiProcessorCount, o)) return 1;
  if (run<4>("LDS + scalar math", d, nGroups, p.multiProcessorCount, o)) return 1;
  if (run<5>("LDS + packed, ping-pong", d, nGroups, p.multiProcessorCount, o)) return 1;
  if (run<6>("LDS + packed, 8 per trip", d, nGroups, p.multiProcessorCount, o)) return 1;
  if (run<7>("2 rays/lane (per 2 trips)", d, nGroups, p.multiProcessorCount, o)) return 1;
  return 0;
}
