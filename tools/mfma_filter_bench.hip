// mfma_filter_bench.hip — can the matrix pipe take part of the renderer's sphere filter?
//
// The filter of the dense sphere loop is a K = 4 contraction per (ray, sphere) pair:
//   h     = (cx, cy, cz, 1) . (-dx, -dy, -dz, o.d)
//   q + k = (cx, cy, cz, k) . (-2ox, -2oy, -2oz, 1)
//   candidate iff h*h - (q + k) >= |o|^2 (1 - 2^-17)
// v_mfma_f32_16x16x4_f32 computes 16 spheres x 16 rays of one such product per instruction, on a pipe of its own.
// This program times, on 64 rays per wave against 4096 spheres:
//   valu    the production loop: 8 spheres per trip from scalar loads, 32 packed instructions + a max-tree
//   mfma    16 spheres per trip: 8 MFMAs (4 blocks of 16 rays x 2 products), threshold in the accumulator input,
//           16 v_fma + a max-tree on the products
//   mixed   per trip M groups of 16 spheres through the matrix pipe and V groups of 8 through the vector ALU
//           (16 M + 8 V a multiple of 16: the MFMA image is read in groups of 16)
//   first of all: bare MFMA chains, bare vector chains and both in one loop (do the two pipes overlap for FP32?)
// and counts the candidates each variant finds (the MFMA's internal rounding may move a few).
// Build: hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -o /tmp/mfma_filter_bench tools/mfma_filter_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f2 mk2(float a, float b) { f2 r; r.x = a; r.y = b; return r; }
__device__ __forceinline__ f2 fma2(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }

// pair image as the renderer lays it out: per pair 2 x float4 (cx0,cx1,cy0,cy1) (cz0,cz1,k0,k1); 8 spheres = 8 float4
// mfma image: per 16 spheres 64 floats, [component 0..3][sphere 0..15], component 3 = k
typedef const f4 __attribute__((address_space(4))) *ConstImage;

#define PAIR_FILT(A, B, T)                                                       \
  {                                                                              \
    const f2 cx = mk2(A.x, A.y), cy = mk2(A.z, A.w), cz = mk2(B.x, B.y);         \
    const f2 hb = fma2(cx, NDX, fma2(cy, NDY, fma2(cz, NDZ, OD)));               \
    const f2 oc2 = fma2(cx, M2OX, fma2(cy, M2OY, cz * M2OZ));                    \
    T = fma2(hb, hb, -oc2) - mk2(B.z, B.w);                                      \
  }

template <int MF, int VA>
__global__ void __launch_bounds__(256) filter_kernel(const float4 *pairImage, const float *mfmaImage, int nSpheres, int passes,
                                                     unsigned long long *out) {
  const int lane = threadIdx.x & 63;
  // one ray per lane
  const unsigned int seed = (blockIdx.x * 256u + threadIdx.x) * 2654435761u;
  const float ox = 5.0f + 0.0001f * (float)(seed & 1023), oy = 1.5f, oz = 6.0f;
  float dx = -0.3f + 0.0005f * (float)((seed >> 10) & 1023), dy = -0.2f + 0.0004f * (float)((seed >> 20) & 1023), dz = -0.9f;
  const float inv = 1.0f / __builtin_sqrtf(dx * dx + dy * dy + dz * dz);
  dx *= inv; dy *= inv; dz *= inv;
  const float od1 = __builtin_fmaf(oz, dz, __builtin_fmaf(oy, dy, ox * dx));
  const float oo1 = __builtin_fmaf(oz, oz, __builtin_fmaf(oy, oy, ox * ox));
  const float thr = __builtin_fmaf(-0x1p-17f, oo1, oo1);
  const f2 OD = mk2(od1, od1), NDX = mk2(-dx, -dx), NDY = mk2(-dy, -dy), NDZ = mk2(-dz, -dz);
  const f2 M2OX = mk2(-2.0f * ox, -2.0f * ox), M2OY = mk2(-2.0f * oy, -2.0f * oy), M2OZ = mk2(-2.0f * oz, -2.0f * oz);
  // MFMA operands of the rays: block b holds rays 16b .. 16b+15; lane l supplies component l / 16 of ray 16b + l % 16
  float bh[4], bq[4], thrB[4];
  {
    const int comp = lane >> 4;
    const float hv[4] = {-dx, -dy, -dz, od1};
    const float qv[4] = {-2.0f * ox, -2.0f * oy, -2.0f * oz, 1.0f};
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const int src = 16 * b + (lane & 15);
      float h = 0.0f, q = 0.0f;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const float hs = __shfl(hv[c], src), qs = __shfl(qv[c], src);
        if (comp == c) { h = hs; q = qs; }
      }
      bh[b] = h; bq[b] = q;
      thrB[b] = __shfl(thr, src);
    }
  }
  const ConstImage g_img = (ConstImage)pairImage;
  unsigned long long cand = 0;
  const int perTrip = 16 * MF + 8 * VA;
  for (int p = 0; p < passes; ++p) {
    for (int s0 = 0; s0 + perTrip <= nSpheres; s0 += perTrip) {
      float m = -__builtin_inff(); // matrix-pipe spheres: max of T - thr over this lane's (ray, sphere) pairs
      float mv = -__builtin_inff();
      f4 dh[MF > 0 ? MF : 1][4], dq[MF > 0 ? MF : 1][4];
#pragma unroll
      for (int g = 0; g < MF; ++g) {
        const float v = mfmaImage[(s0 + 16 * g) * 4 + lane];
        const float ah = lane < 48 ? v : 1.0f; // (cx, cy, cz, 1)
#pragma unroll
        for (int b = 0; b < 4; ++b) {
          const f4 z = {0.0f, 0.0f, 0.0f, 0.0f};
          const f4 t = {thrB[b], thrB[b], thrB[b], thrB[b]};
          dh[g][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(ah, bh[b], z, 0, 0, 0);
          dq[g][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(v, bq[b], t, 0, 0, 0); // q + k + threshold
        }
      }
      f2 d[VA > 0 ? VA : 1][4];
#pragma unroll
      for (int t = 0; t < VA; ++t) {
        const int G = (s0 + 16 * MF) / 8 + t;
        const f4 a0 = g_img[8 * G], b0 = g_img[8 * G + 1], a1 = g_img[8 * G + 2], b1 = g_img[8 * G + 3];
        const f4 a2 = g_img[8 * G + 4], b2 = g_img[8 * G + 5], a3 = g_img[8 * G + 6], b3 = g_img[8 * G + 7];
        PAIR_FILT(a0, b0, d[t][0]) PAIR_FILT(a1, b1, d[t][1]) PAIR_FILT(a2, b2, d[t][2]) PAIR_FILT(a3, b3, d[t][3])
        float x = __builtin_fmaxf(__builtin_fmaxf(d[t][0].x, d[t][0].y), d[t][1].x);
        x = __builtin_fmaxf(__builtin_fmaxf(x, d[t][1].y), d[t][2].x);
        x = __builtin_fmaxf(__builtin_fmaxf(x, d[t][2].y), d[t][3].x);
        x = __builtin_fmaxf(x, d[t][3].y);
        mv = __builtin_fmaxf(mv, x);
      }
#pragma unroll
      for (int g = 0; g < MF; ++g) {
#pragma unroll
        for (int b = 0; b < 4; ++b) {
          const float t0 = __builtin_fmaf(dh[g][b].x, dh[g][b].x, -dq[g][b].x), t1 = __builtin_fmaf(dh[g][b].y, dh[g][b].y, -dq[g][b].y);
          const float t2 = __builtin_fmaf(dh[g][b].z, dh[g][b].z, -dq[g][b].z), t3 = __builtin_fmaf(dh[g][b].w, dh[g][b].w, -dq[g][b].w);
          m = __builtin_fmaxf(__builtin_fmaxf(m, t0), t1);
          m = __builtin_fmaxf(__builtin_fmaxf(m, t2), t3);
        }
      }
      const bool any = (MF > 0 && m >= 0.0f) || (VA > 0 && mv >= thr);
      if (__ballot(any) != 0ull) { // the slow path: count the candidates pair by pair
#pragma unroll
        for (int g = 0; g < MF; ++g)
#pragma unroll
          for (int b = 0; b < 4; ++b) {
            const float *h = (const float *)&dh[g][b], *q = (const float *)&dq[g][b];
#pragma unroll
            for (int r = 0; r < 4; ++r) cand += __builtin_fmaf(h[r], h[r], -q[r]) >= 0.0f ? 1u : 0u;
          }
#pragma unroll
        for (int t = 0; t < VA; ++t)
#pragma unroll
          for (int k = 0; k < 4; ++k) cand += (d[t][k].x >= thr ? 1u : 0u) + (d[t][k].y >= thr ? 1u : 0u);
      }
    }
  }
  for (int off = 32; off > 0; off >>= 1) cand += __shfl_down(cand, off, 64);
  if (lane == 0) atomicAdd(out, cand);
}

// Do the two pipes overlap at all for FP32?  NM independent MFMA chains and NV independent vector chains per trip
// (PK: packed v_pk_fma_f32, else v_fma_f32); time of both against each alone.
template <int NM, int NV, bool PK>
__global__ void __launch_bounds__(256) coissue_kernel(int trips, float *out) {
  f4 acc[8];
  f2 x[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { acc[i] = (f4){0.0f, 0.0f, 0.0f, 0.0f}; x[i] = mk2(0.001f * threadIdx.x, 0.002f * i); }
  const float a = 1.0f + 1e-6f * threadIdx.x, b = 1.0f - 1e-6f * threadIdx.x;
  const f2 c1 = mk2(0.999f, 1.001f), c2 = mk2(1e-3f, -1e-3f);
  for (int t = 0; t < trips; ++t) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (i < NM) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
        if (PK) { if (i < NV) x[i] = fma2(x[i], c1, c2); }
        else { if (i < NV) { x[i].x = __builtin_fmaf(x[i].x, c1.x, c2.x); } if (i < NV) { x[i].y = __builtin_fmaf(x[i].y, c1.y, c2.y); } }
      }
    }
  }
  float s = 0.0f;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += acc[i].x + acc[i].y + acc[i].z + acc[i].w + x[i].x + x[i].y;
  if (s == 123.456f) out[0] = s;
}

template <int NM, int NV, bool PK>
int run_coissue(const char *name, int cus, float *dOut) {
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  for (int blocksPerCu : {1, 2, 4}) {
    const int trips = 20000;
    hipLaunchKernelGGL((coissue_kernel<NM, NV, PK>), dim3(cus * blocksPerCu), dim3(256), 0, 0, 10, dOut);
    CHECK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
      CHECK(hipEventRecord(e0));
      hipLaunchKernelGGL((coissue_kernel<NM, NV, PK>), dim3(cus * blocksPerCu), dim3(256), 0, 0, trips, dOut);
      CHECK(hipEventRecord(e1));
      CHECK(hipEventSynchronize(e1));
      float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
      if (ms < best) best = ms;
    }
    // per SIMD and trip: 4 NM MFMAs and 4 NV (packed: x1, plain: x2) vector instructions per wave
    printf("%-34s waves/SIMD %d  %8.3f ms  %7.1f ns per trip and wave\n", name, blocksPerCu, best, best * 1e6 / trips / blocksPerCu);
  }
  return 0;
}

static double frand(unsigned long long &s) { s = s * 6364136223846793005ull + 1442695040888963407ull; return (double)(s >> 11) / 9007199254740992.0; }

template <int MF, int VA>
int run(const char *name, const float4 *dPairs, const float *dMf, int nSpheres, int cus, unsigned long long *dOut) {
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  const int perTrip = 16 * MF + 8 * VA;
  const int used = nSpheres / perTrip * perTrip;
  for (int blocksPerCu : {1, 2, 4}) { // 256 threads = one wave per SIMD per workgroup
    const int passes = 20;
    CHECK(hipMemset(dOut, 0, 8));
    hipLaunchKernelGGL((filter_kernel<MF, VA>), dim3(cus * blocksPerCu), dim3(256), 0, 0, dPairs, dMf, nSpheres, 1, dOut);
    CHECK(hipDeviceSynchronize());
    unsigned long long c1 = 0;
    CHECK(hipMemcpy(&c1, dOut, 8, hipMemcpyDeviceToHost));
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
      CHECK(hipEventRecord(e0));
      hipLaunchKernelGGL((filter_kernel<MF, VA>), dim3(cus * blocksPerCu), dim3(256), 0, 0, dPairs, dMf, nSpheres, passes, dOut);
      CHECK(hipEventRecord(e1));
      CHECK(hipEventSynchronize(e1));
      float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
      if (ms < best) best = ms;
    }
    const double tests = (double)cus * blocksPerCu * 256 * passes * used;
    const double cycPerSphere = best * 1e-3 * 2.4e9 / ((double)passes * used * blocksPerCu); // per wave's share of its SIMD
    printf("%-26s waves/SIMD %d  %8.3f ms  %6.2f cycles per sphere and wave  %8.1f Gtests/s  alg %6.1f TFLOP/s  candidates per ray-pass %.3f\n",
           name, blocksPerCu, best, cycPerSphere, tests / (best * 1e-3) / 1e9, tests * 17 / (best * 1e-3) / 1e12,
           (double)c1 / ((double)cus * blocksPerCu * 256));
  }
  return 0;
}

int main() {
  hipDeviceProp_t p; CHECK(hipGetDeviceProperties(&p, 0));
  const int n = 4096 + 16 * 8; // a little more than the largest trip needs to end on its boundary
  unsigned long long s = 0xC0FFEE03ull;
  std::vector<float> cx(n), cy(n), cz(n), kk(n);
  for (int i = 0; i < n; ++i) {
    cx[i] = (float)(-10.0 + 24.0 * frand(s)); cy[i] = (float)(-0.3 + 5.3 * frand(s)); cz[i] = (float)(-20.0 + 23.0 * frand(s));
    const double r = 0.05 + 0.2 * frand(s);
    const double c2 = (double)cx[i] * cx[i] + (double)cy[i] * cy[i] + (double)cz[i] * cz[i];
    kk[i] = (float)(c2 - r * r - std::ldexp(c2 + r * r, -17) - 1e-30);
    kk[i] = std::nextafterf(kk[i], -INFINITY);
  }
  std::vector<float4> pairs(n); // 8 spheres = 8 float4: 4 pairs x (A, B)
  for (int i = 0; i < n; i += 2) {
    pairs[i] = make_float4(cx[i], cx[i + 1], cy[i], cy[i + 1]);
    pairs[i + 1] = make_float4(cz[i], cz[i + 1], kk[i], kk[i + 1]);
  }
  std::vector<float> mf((size_t)n * 4);
  for (int g = 0; g < n / 16; ++g)
    for (int j = 0; j < 16; ++j) {
      mf[(size_t)g * 64 + j] = cx[16 * g + j]; mf[(size_t)g * 64 + 16 + j] = cy[16 * g + j];
      mf[(size_t)g * 64 + 32 + j] = cz[16 * g + j]; mf[(size_t)g * 64 + 48 + j] = kk[16 * g + j];
    }
  float4 *dPairs; float *dMf; unsigned long long *dOut;
  CHECK(hipMalloc(&dPairs, pairs.size() * 16)); CHECK(hipMalloc(&dMf, mf.size() * 4)); CHECK(hipMalloc(&dOut, 8));
  CHECK(hipMemcpy(dPairs, pairs.data(), pairs.size() * 16, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(dMf, mf.data(), mf.size() * 4, hipMemcpyHostToDevice));
  const int cus = p.multiProcessorCount;
  {
    float *dF; CHECK(hipMalloc(&dF, 4));
    if (run_coissue<8, 0, true>("32 mfma", cus, dF)) return 1;
    if (run_coissue<0, 8, true>("32 v_pk_fma", cus, dF)) return 1;
    if (run_coissue<8, 8, true>("32 mfma + 32 v_pk_fma", cus, dF)) return 1;
    if (run_coissue<0, 8, false>("64 v_fma", cus, dF)) return 1;
    if (run_coissue<8, 8, false>("32 mfma + 64 v_fma", cus, dF)) return 1;
    if (run_coissue<8, 4, false>("32 mfma + 32 v_fma", cus, dF)) return 1;
    if (run_coissue<0, 4, false>("32 v_fma", cus, dF)) return 1;
  }
  if (run<0, 1>("valu 8/trip (production)", dPairs, dMf, 4096, cus, dOut)) return 1;
  if (run<0, 2>("valu 16/trip", dPairs, dMf, 4096, cus, dOut)) return 1;
  if (run<1, 0>("mfma 16/trip", dPairs, dMf, 4096, cus, dOut)) return 1;
  if (run<2, 0>("mfma 32/trip", dPairs, dMf, 4096, cus, dOut)) return 1;
  if (run<2, 2>("mixed 32 mfma + 16 valu", dPairs, dMf, 4096, cus, dOut)) return 1;
  if (run<1, 2>("mixed 16 mfma + 16 valu", dPairs, dMf, 4096, cus, dOut)) return 1;
  return 0;
}
