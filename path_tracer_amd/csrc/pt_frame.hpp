// pt_frame.hpp — what one frame launch is told (FrameParams), the tile / pixel / camera
// mapping shared by the kernels, and the small kernels around the render kernel: the tile
// drawing-order pass, the volume re-tiling and the de-interleave of gathered row bands.
#pragma once

#include "pt_math.hpp"

namespace ptdev {

constexpr float kMaxDistance = 1e10f; // raytrace.comp:94
constexpr float kEpsilon = 0.001f;    // raytrace.comp:95
constexpr int kTileDim = 8;           // pixels per tile edge: one 8x8 tile = one refill of a whole wave
constexpr int kTilePixels = kTileDim * kTileDim;
constexpr int kFanRounds = 16;        // fan-out: rays handed out per lighting lane and pass, at most
constexpr int kFanBytes = 13 * 64 * 4 + kFanRounds * 12; // fan-out records and round tables of one wave
constexpr int kTicketGroups = 32;     // tile-ticket counters (power of two), one per group of workgroups
constexpr int kTicketStride = 32;     // unsigned ints between two counters: one 128-byte line each

struct FrameParams {
  // camera basis, computed once per frame on the host with the same float
  // operations createCameraRay uses per pixel (raytrace.comp:121-131)
  float cam[3], fwd[3], right[3], up[3];
  float aspect, fov;
  float time;
  float bgTop[3], bgBottom[3];
  int width, height;     // full frame
  int localRows;         // rows this renderer draws (compact band-major frame)
  int bandRows, bandIndex, bandCount;
  int numSpheres, numEllipsoids, numLights, numVolumes;
  int numEmSpheres, numEmEllipsoids;
  int maxBounces;
  int tilesX, numTiles;
  int sceneHuge;         // 1: some sphere has a non-finite or > 1e9 coordinate (no fast path)
  float ellCenterMax;    // largest |centre coordinate| over the ellipsoids
  float ellScale;        // bound on (1/radius) x (growth under rotateByQuaternion) over the ellipsoids; +inf: no deferral
  // volume 0 (raytrace.comp:538-550); boxMin/boxMax/extent precomputed
  float volPos[3], v0[3], v1[3], boxMin[3], boxMax[3], extent[3];
  int res[3];
  int volMaterial;
  // Scene image: one contiguous block of float4, copied into LDS by every
  // workgroup when it fits (offsets in float4 units):
  //   [0)          sphere pairs, 2 float4 per pair (cx0,cx1,cy0,cy1) (cz0,cz1,w0,w1),
  //                padded with copies of the last sphere to a multiple of 8 spheres;
  //                w = radius, or with sphereFilter the filter constant k (see below)
  //   [offRadius)  one float per sphere: radius (read when a candidate's roots are evaluated)
  //   [offEll)     4 per ellipsoid: (c,qw) (1/radii,maxRadius) (-q.xyz,0) (-2q.xyz,0)
  //   [offEllPair) 8 per pair of ellipsoids, every field as (first, second) of the pair:
  //                (cx,cy) (cz,c1) (-c2,m00) (m01,m02) (m10,m11) (m12,m20) (m21,m22) (k1,k2): centre, the constants
  //                of the bounding-sphere test, the matrix M = diag(1/radii) R(conjugate q) and the margin
  //                constants of the matrix test (the two stages of the candidate loop); an odd count is padded
  //                with a copy of the last ellipsoid
  //   [offMat)     5 per material, as GPUMaterial
  //   [offLight)   1 per light: (pos, intensity)
  //   [offEmS)     3 per emissive sphere, ascending sphere index:
  //                (c,r) (material colour, emissiveStrength) (shaderIndex, sphereIndex as int bits)
  //   [offEmE)     3 per emissive ellipsoid: (c,maxRadius) (material colour, emissiveStrength) (ellipsoidIndex bits)
  //   [offMatIdx)  uint16 material index per sphere, then per ellipsoid
  const float4 *image;
  int imageFloat4s;
  int offRadius, offEll, offEllPair, offMat, offLight, offEmS, offEmE, offMatIdx;
  int matIdxEllBase;         // element offset of the ellipsoid part of the uint16 table
  const float4 *spheres;     // (cx, cy, cz, r), one per sphere, global memory
  const uint8_t *voxels;   // linear (x fastest, as uploaded) when brickX == 0, else bricked
  int brickX, brickY;      // bricks per row and per slice of the bricked copy
  uint32_t *outRgba8;
  float4 *outF32;           // may be null
  unsigned int *tileCounter;    // kTicketGroups counters, kTicketStride apart
  unsigned long long *counters; // PT_CNT_COUNT slots, instrumented kernel only
  unsigned long long *waveLog;  // or null; phase-instrumented kernel: 4 words per wave (start, tickets drained, exit stamps; passes)
  // Optional drawing order (null: tiles are drawn in index order).  tileOrder holds
  // three lists of numTiles entries each, one per cost class; classCount[c] is the
  // length of list c.  Tickets run through class 2, then 1, then 0.
  int *tileOrder;
  unsigned int *classCount;
  // Extensions without a counterpart in the reference (all zero: the reference's frame)
  int volumeFilter; // 0 nearest neighbour (raytrace.comp:485-517), 1 trilinear
  int sampleIndex;  // sub-pixel sample drawn by this launch; 0 = the reference's pixel-corner sample
  int accumCount;   // with accum: number of samples in the sum after this launch
  float4 *accum;    // running sum of the clamped samples per pixel, or null; in a launch of several samples: the planes
  // Several samples in one launch (the general kernel's SAMPLES variant): the tickets run over numTiles x sampleCount
  // (tile, sample) pairs, tile-major in drawing order, sample sampleIndex + s of a pixel goes to
  // accum[s * planeStride + pixel] (one plane of clamped colours per sample) and pt_sum_samples_kernel adds the planes
  // to the running sum in sample order afterwards.
  int sampleCount;  // samples drawn by this launch, >= 1
  int planeStride;  // pixels per plane
  int fanOut;       // 1: at the end of the frame lighting rays of one pixel are spread over idle lanes (not with coop)
  int coop;         // 1: at the end of the frame the waves of a workgroup pool their rays (see the kernel)
  float unitTol;    // dot(d,d) - 1 up to which a ray takes the dense sphere loop (longer: too many spheres are candidates)
  float unitTolLow; // 1 - dot(d,d) up to which it does (1: any shorter direction; those have fewer candidates, not more)
  // 1: the dense sphere loop runs a cheaper conservative test in place of hitSphere's discriminant:
  //   (o.d - c.d)^2 - (k - 2 o.c) >= |o|^2 (1 - 2^-17),  k = |c|^2 - r^2 - 2^-17 (|c|^2 + r^2) - 1e-30
  // which holds whenever the float discriminant is >= 0 (rounding of both sides is below the 2^-17
  // terms for |d|^2 <= 1.75 (1 + unitTol cap; shorter directions only make the rounding terms smaller) and coordinates <= 1e9); the candidates it lets through are then
  // evaluated with hitSphere's own operations, so the result does not change.
  int sphereFilter;
  float rcpExtent[3]; // 1 / extent[i], rounded once (from a double on the host): the march's fast voxel index
  int exactExtent;    // 1: every extent[i] is a power of two, so x * rcpExtent[i] is exactly the quotient x / extent[i]
  int sparseLanes;  // at most this many live rays in a wave: trace them one by one (exact routine)
  int refillMin;    // idle lanes that make the refill of a wave worth its fixed cost (1: any idle lane is refilled at once)
  int tune;         // tuning switches of pt_set_option(100, ...), 0 in production; the kernel reads bit 0: no wave priorities
};

// Pixel w (0..63) of the tile in tile column tx, tile row ty: x, local row ly (row of
// the compact band-major frame) and full-frame row gy.  False when the pixel lies
// outside the frame.
__device__ __forceinline__ bool tile_pixel_xy(const FrameParams &P, int tx, int ty, int w, int &x, int &ly, int &gy) {
  x = tx * kTileDim + (w & 7);
  ly = ty * kTileDim + (w >> 3);
  gy = ly;
  if (P.bandCount > 1) {
    const int band = ly / P.bandRows;
    gy = (band * P.bandCount + P.bandIndex) * P.bandRows + (ly - band * P.bandRows);
  }
  return x < P.width && ly < P.localRows && gy < P.height;
}
__device__ __forceinline__ bool tile_pixel(const FrameParams &P, int tile, int w, int &x, int &ly, int &gy) {
  return tile_pixel_xy(P, tile % P.tilesX, tile / P.tilesX, w, x, ly, gy);
}

// createCameraRay for pixel (x, gy) (raytrace.comp:121-140 with main()'s uv, :692-700)
__device__ __forceinline__ V3 camera_dir(const FrameParams &P, int x, int gy, int sample) {
  float fx = (float)x, fy = (float)gy;
  if (sample != 0) {
    // sub-pixel offset in [0,1)^2 from a counter-based hash of (pixel, sample)
    uint32_t h = ((uint32_t)gy * (uint32_t)P.width + (uint32_t)x) * 0x9E3779B1u + (uint32_t)sample * 0x85EBCA77u;
    h ^= h >> 16; h *= 0x7FEB352Du; h ^= h >> 15; h *= 0x846CA68Bu; h ^= h >> 16;
    fx = fx + (float)(h & 0xFFFFu) / 65536.0f;
    fy = fy + (float)(h >> 16) / 65536.0f;
  }
  float uvx = fx / (float)P.width;
  float uvy = fy / (float)P.height;
  uvy = 1.0f - uvy;
  const float sx = ((uvx - 0.5f) * P.aspect) * P.fov;
  const float sy = (uvy - 0.5f) * P.fov;
  return normalize3(madd3(madd3(ld3(P.fwd), sx, ld3(P.right)), sy, ld3(P.up)));
}
__device__ __forceinline__ V3 camera_dir(const FrameParams &P, int x, int gy) { return camera_dir(P, x, gy, P.sampleIndex); }

// Drawing-order pass: what the centre pixel of every tile sees first files the
// tile under a cost class: 2 = a surface that bounces the path on (long ray
// chains), 1 = any other surface or the volume box, 0 = background.  The render
// kernel draws class 2 first and class 0 last, so the pixels with the longest
// chains of rays are started early and the frame does not end on them.  Only the
// ORDER of the tiles depends on this pass; what is drawn for each pixel does
// not, so nothing here needs the render kernel's exact arithmetic.
// A workgroup of LANES x TILES threads files TILES tiles: LANES lanes share a tile's ray and take every
// LANES-th object each, and the workgroup draws its places in the three lists with one atomic per class.
// The host launches <4, 256> for a frame drawn alone (the pass runs in front of every frame; one lane walking
// all objects of a tile and three same-address atomics per 64 tiles made it 3 % of a small frame: 15.5 -> 8.4 us
// at 1080p with 64 objects) and <1, 64>, single waves, with frames in flight: there the pass of the next frame has
// to find room beside the previous frame's resident waves, which leave one by one, and a workgroup of sixteen
// waves waits for a whole compute unit (measured: the frame rate with two in flight drops by 3 %).
// clearWords / clearCount: the frame's ticket and work counters, cleared here so that no clearing launch has to
// precede the pass; clearNextClass: the three class counters the NEXT ordered frame of these buffers will use (a
// frame's own were cleared by the pass before it; the two sets alternate).  Null: the host cleared everything.
template <bool STAGED, int LANES, int TILES>
__global__ void __launch_bounds__(LANES * TILES) pt_tile_order_kernel(const FrameParams P, unsigned int *clearWords, int clearCount,
                                                                      unsigned int *clearNextClass) {
  if (blockIdx.x == 0 && clearWords) {
    for (int i = threadIdx.x; i < clearCount; i += blockDim.x) clearWords[i] = 0u;
    if (threadIdx.x < 3) clearNextClass[threadIdx.x] = 0u;
  }
  // STAGED: the spheres and ellipsoid records are copied to LDS first (the loops below
  // would otherwise wait for one dependent global load per object)
  extern __shared__ float4 s_objs[];
  __shared__ int s_cls[TILES];
  __shared__ unsigned int s_waveCount[TILES / 64][3], s_base[3];
  if (STAGED) {
    for (int i = threadIdx.x; i < P.numSpheres; i += blockDim.x) s_objs[i] = P.spheres[i];
    for (int i = threadIdx.x; i < 4 * P.numEllipsoids; i += blockDim.x) s_objs[P.numSpheres + i] = P.image[P.offEll + i];
    __syncthreads();
  }
  const int lane = threadIdx.x & 63, sub = lane & (LANES - 1);
  const int place = (int)threadIdx.x / LANES; // tile within the workgroup's TILES
  {
    const int tileRaw = (int)blockIdx.x * TILES + place;
    const bool inRange = tileRaw < P.numTiles;
    const int tile = inRange ? tileRaw : P.numTiles - 1;
    int x, ly, gy;
    tile_pixel(P, tile, 4 * 8 + 4, x, ly, gy);
    if (x >= P.width) x = P.width - 1;
    if (gy >= P.height) gy = P.height - 1;
    const V3 o = ld3(P.cam), d = camera_dir(P, x, gy);
    // approximate square roots and reciprocals are good enough to rank tiles
    float best = kMaxDistance;
    int bestIdx = -1;
    for (int i = sub; i < P.numSpheres; i += LANES) {
      const float4 sp = STAGED ? s_objs[i] : P.spheres[i];
      const V3 oc = sub3(o, xyz(sp));
      const float hb = dot3(oc, d);
      const float disc = hb * hb - (dot3(oc, oc) - sp.w * sp.w);
      if (disc >= 0.0f) {
        const float sq = __builtin_amdgcn_sqrtf(disc);
        float t = -hb - sq;
        if (t < kEpsilon) t = -hb + sq;
        if (t >= kEpsilon && t <= best) { best = t; bestIdx = i; }
      }
    }
    for (int j = sub; j < P.numEllipsoids; j += LANES) {
      const float4 *er = STAGED ? s_objs + P.numSpheres + 4 * j : P.image + P.offEll + 4 * j;
      const float4 e0 = er[0], e1 = er[1], e2 = er[2], e3 = er[3];
      const V3 oc = mul3(rotate_q(sub3(o, xyz(e0)), xyz(e2), xyz(e3), e0.w), xyz(e1));
      const V3 dir = mul3(rotate_q(d, xyz(e2), xyz(e3), e0.w), xyz(e1));
      const float a = dot3(dir, dir), hb = dot3(oc, dir), c = dot3(oc, oc) - 1.0f;
      const float disc = hb * hb - a * c;
      if (disc >= 0.0f) {
        const float sq = __builtin_amdgcn_sqrtf(disc), ia = __builtin_amdgcn_rcpf(a);
        float t = (-hb - sq) * ia;
        if (t < kEpsilon) t = (-hb + sq) * ia;
        if (t >= kEpsilon && t <= best) { best = t; bestIdx = P.numSpheres + j; }
      }
    }
    // the nearest of the lanes' finds
    for (int off = LANES / 2; off > 0; off >>= 1) {
      const float ot = __shfl_xor(best, off, 64);
      const int oi = __shfl_xor(bestIdx, off, 64);
      if (ot < best) { best = ot; bestIdx = oi; }
    }
    if (sub == 0) {
      int cls = 0;
      if (bestIdx >= 0) {
        const unsigned short *mi = (const unsigned short *)(P.image + P.offMatIdx);
        const int m = bestIdx < P.numSpheres ? mi[bestIdx] : mi[P.matIdxEllBase + (bestIdx - P.numSpheres)];
        const float reflectivity = P.image[P.offMat + 5 * m + 1].w, transparency = P.image[P.offMat + 5 * m + 2].x;
        cls = (P.maxBounces > 1 && (reflectivity > 0.01f || transparency > 0.01f)) ? 2 : 1;
      }
      if (cls == 0 && P.numVolumes > 0) {
        // slab test against the volume box (raytrace.comp:519-533), any overlap in front of the eye
        float t0 = 0.0f, t1 = kMaxDistance;
        const float oo[3] = {o.x, o.y, o.z}, dd[3] = {d.x, d.y, d.z};
        for (int a = 0; a < 3; ++a) {
          const float inv = 1.0f / dd[a];
          const float ta = (P.boxMin[a] - oo[a]) * inv, tb = (P.boxMax[a] - oo[a]) * inv;
          t0 = __builtin_fmaxf(t0, __builtin_fminf(ta, tb));
          t1 = __builtin_fminf(t1, __builtin_fmaxf(ta, tb));
        }
        if (t0 <= t1) cls = 1;
      }
      s_cls[place] = inRange ? cls : -1;
    }
  }
  __syncthreads();
  // The first TILES threads file the tiles: per class, the tiles of a wave take consecutive places behind the
  // tiles of the waves before it, behind the value of the class counter the workgroup drew.
  const int filer = (int)(threadIdx.x >> 6);
  const bool files = filer < TILES / 64;
  const int cls = files ? s_cls[threadIdx.x] : -1;
  const unsigned long long m0 = __ballot(cls == 0), m1 = __ballot(cls == 1), m2 = __ballot(cls == 2);
  if (files && lane < 3) s_waveCount[filer][lane] = (unsigned int)__popcll(lane == 0 ? m0 : (lane == 1 ? m1 : m2));
  __syncthreads();
  if (threadIdx.x < 3) {
    unsigned int total = 0;
    for (int w = 0; w < TILES / 64; ++w) total += s_waveCount[w][threadIdx.x];
    s_base[threadIdx.x] = total != 0u ? atomicAdd(&P.classCount[threadIdx.x], total) : 0u;
  }
  __syncthreads();
  if (cls >= 0) {
    unsigned int at = s_base[cls];
    for (int w = 0; w < filer; ++w) at += s_waveCount[w][cls];
    const unsigned long long lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    const unsigned long long m = cls == 0 ? m0 : (cls == 1 ? m1 : m2);
    P.tileOrder[cls * P.numTiles + (int)(at + (unsigned int)__popcll(m & lt))] = (int)blockIdx.x * TILES + (int)threadIdx.x;
  }
}

// Re-tile a linear x-fastest volume into 8x8x8 bricks (512 contiguous bytes
// each; bricks x fastest).  Voxels outside the volume are written as 0.
__global__ void __launch_bounds__(256) pt_brick_volume_kernel(const uint8_t *__restrict__ linear, uint8_t *__restrict__ bricks,
                                                              int rx, int ry, int rz, int nbx, int nby, int nbz) {
  const long long total = (long long)nbx * nby * nbz * 128; // 4 voxels per thread
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const long long brick = i >> 7;
    const int in = (int)(i & 127) << 2; // first of 4 voxels along x
    const int bx = (int)(brick % nbx), by = (int)((brick / nbx) % nby), bz = (int)(brick / ((long long)nbx * nby));
    const int ix = (bx << 3) | (in & 7), iy = (by << 3) | ((in >> 3) & 7), iz = (bz << 3) | (in >> 6);
    uint32_t w = 0;
    if (iy < ry && iz < rz) {
      const long long base = ix + (long long)iy * rx + (long long)iz * rx * ry;
      for (int k = 0; k < 4; ++k)
        if (ix + k < rx) w |= (uint32_t)linear[base + k] << (8 * k);
    }
    ((uint32_t *)bricks)[i] = w;
  }
}

// Occupies its stream for `ticks` of the 100 MHz chip-wide clock (the host uses it to find out how many of the
// renderer's streams the device really runs side by side).  The clock advances by itself, so every wave leaves.
__global__ void __launch_bounds__(64) pt_spin_kernel(unsigned long long ticks) {
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(16);
}

template <typename T>
__global__ void pt_assemble_bands_kernel(const T *__restrict__ src, T *__restrict__ dst, int width, int height,
                                         int bandRows, int bandCount, int strideRows) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x;
  if (x >= width) return;
  for (int gy = blockIdx.y; gy < height; gy += gridDim.y) { // grid.y is capped at 65535
    const int band = gy / bandRows;
    const int owner = band % bandCount;
    const int localBand = band / bandCount;
    const int ly = localBand * bandRows + (gy - band * bandRows);
    dst[(size_t)gy * width + x] = src[((size_t)owner * strideRows + ly) * width + x];
  }
}

} // namespace ptdev
