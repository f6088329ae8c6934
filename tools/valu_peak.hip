// valu_peak.hip — what vector-FP32 rate can a gfx950 CU actually issue?
//
// Times long runs of independent v_fma_f32 and v_pk_fma_f32 at 1, 2, 4 and 8
// waves per SIMD and prints achieved TFLOP/s (2 flop per FMA lane-op, 4 per
// packed one).  The renderer's roofline is quoted against the 157.3 TFLOP/s
// datasheet figure; this shows which instruction mix can reach it.
//
//   hipcc -O3 --offload-arch=gfx950 -o valu_peak tools/valu_peak.hip && ./valu_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

typedef float float2v __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ void __launch_bounds__(256) fma_loop(float *out, int iters, float a, float b) {
  float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
  float2v p0 = {x0, x1}, p1 = {x2, x3}, p2 = {x4, x5}, p3 = {x6, x7}, p4 = {x1, x0}, p5 = {x3, x2}, p6 = {x5, x4}, p7 = {x7, x6};
  float2v pa = {a, a}, pb = {b, b};
  for (int i = 0; i < iters; ++i) {
    if (MODE == 0) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                     "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                     : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b));
      }
    } else if (MODE == 1) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        asm volatile("v_pk_fma_f32 %0, %0, %8, %9\n v_pk_fma_f32 %1, %1, %8, %9\n v_pk_fma_f32 %2, %2, %8, %9\n v_pk_fma_f32 %3, %3, %8, %9\n"
                     "v_pk_fma_f32 %4, %4, %8, %9\n v_pk_fma_f32 %5, %5, %8, %9\n v_pk_fma_f32 %6, %6, %8, %9\n v_pk_fma_f32 %7, %7, %8, %9\n"
                     : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(pa), "v"(pb));
      }
    } else { // mixed stream like the sphere test: sub, mul, fma
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        asm volatile("v_sub_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                     "v_sub_f32 %4, %4, %8\n v_mul_f32 %5, %5, %8\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                     : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b));
      }
    }
  }
  float s = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y + p4.x + p4.y + p5.x +
            p5.y + p6.x + p6.y + p7.x + p7.y;
  if (s == 12345.678f) out[0] = s;
}

template <int MODE>
int run(const char *name, double flop_per_inst, int cus) {
  float *out;
  CHECK(hipMalloc(&out, 4));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  const int iters = 20000;
  for (int wavesPerSimd : {1, 2, 4, 8}) {
    int blocks = cus * wavesPerSimd; // 256 threads = 4 waves = 1 per SIMD
    hipLaunchKernelGGL(fma_loop<MODE>, dim3(blocks), dim3(256), 0, 0, out, 100, 0.999f, 0.001f);
    CHECK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int rep = 0; rep < 5; ++rep) {
      CHECK(hipEventRecord(e0));
      hipLaunchKernelGGL(fma_loop<MODE>, dim3(blocks), dim3(256), 0, 0, out, iters, 0.999f, 0.001f);
      CHECK(hipEventRecord(e1));
      CHECK(hipEventSynchronize(e1));
      float ms;
      CHECK(hipEventElapsedTime(&ms, e0, e1));
      if (ms < best) best = ms;
    }
    double insts = (double)blocks * 256.0 * iters * 32.0; // lane-instructions
    double tf = insts * flop_per_inst / (best * 1e-3) / 1e12;
    double cyc = best * 1e-3 * 2.4e9 / ((double)iters * 32.0 * wavesPerSimd);
    printf("%-14s waves/SIMD %d  %8.3f ms  %7.2f TFLOP/s  (%.2f cycles per wave-instruction per SIMD at 2.4 GHz)\n", name,
           wavesPerSimd, best, tf, cyc);
  }
  CHECK(hipFree(out));
  return 0;
}

int main() {
  hipDeviceProp_t p;
  CHECK(hipGetDeviceProperties(&p, 0));
  printf("device: %s, %d CUs, clock %d kHz\n", p.gcnArchName, p.multiProcessorCount, p.clockRate);
  if (run<0>("v_fma_f32", 2.0, p.multiProcessorCount)) return 1;
  if (run<1>("v_pk_fma_f32", 4.0, p.multiProcessorCount)) return 1;
  if (run<2>("sub/mul/fma", 1.5, p.multiProcessorCount)) return 1;
  return 0;
}
