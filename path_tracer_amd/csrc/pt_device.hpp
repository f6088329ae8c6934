// pt_device.hpp — gfx950 device code of the per-pixel ray loop.
//
// One lane owns one pixel at a time and runs it as a small state machine:
// every trip round the main loop each lane holds exactly one ray (a path ray,
// a light shadow ray or an emissive-object occlusion ray), the whole wave
// tests its 64 rays against the object list in one wave-uniform loop over
// LDS-staged objects, and then each lane consumes its result and emits its
// next ray.  A lane whose pixel is finished takes the next pixel of the
// wave's tile pool (ballot + prefix count), so the intersection loop always
// runs with every lane live until the frame runs out of pixels.
//
// What it computes follows the reference shader function by function
// (reference shaders/raytrace.comp, shaders/surface_shaders/*.glsl); the
// floating-point contract is:
//   dot(a,b)      = fma(a.z,b.z, fma(a.y,b.y, a.x*b.x))
//   cross(a,b).x  = fma(a.y,b.z, -(a.z*b.y))
//   a + s*b       = fma(s, b, a)   for points / directions (madd3)
//   normalize(v)  = v * (1 / sqrt(dot(v,v)));  vec / scalar = IEEE division
//   sphere  : c = fma(-r, r, dot(oc,oc)), disc = fma(hb, hb, -c)
//   ellipsoid: disc = fma(hb, hb, -(a*c));  volume: acc = fma(1-acc, alpha, acc)
//   everything else: one IEEE operation per source operator, left to right.
// The translation unit must be compiled with -ffp-contract=off so that these
// explicit fmaf calls are the only fused operations.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "pt_frame.hpp"
#include "pt_shaders.hpp"
#include "pt_volume.hpp"

namespace ptdev {

// Where the dense sphere loop reads the sphere pairs from (compile-time choice per call site).

struct ScalarTag { static constexpr bool value = true; };
struct LdsTag { static constexpr bool value = false; };

// Lane states: which kind of ray the lane has in flight.
enum : int {
  ST_IDLE = 0,       // wants a pixel
  ST_PATH = 1,       // primary / bounce ray, closest hit        (raytrace.comp:610)
  ST_LIGHT = 2,      // getShadowFactor for light k, Phong        (raytrace.comp:345)
  ST_SHADER_LIGHT = 3, // getShadowFactor for light k, in a surface shader's light loop (toon.glsl:19)
  ST_EM_SPHERE = 4,  // emissive-sphere occlusion ray             (raytrace.comp:431)
  ST_EM_ELLIPSOID = 5, // emissive-ellipsoid occlusion ray        (raytrace.comp:469)
  ST_JOB = 6,        // a shadow / occlusion ray cast on behalf of another lane (fan-out, see the kernel)
  ST_WAIT = 7        // all of this lane's lighting rays are with other lanes; it casts none itself this pass
};

// calculateLighting skips an emissive object unless alignment = dot(N, (c - p) / |c - p|) > 0 (raytrace.comp:417-421,
// 452-455).  True where that alignment, computed in float32 exactly as the shader computes it, is certain to be <= 0,
// from a = dot(N, c - p) alone: with t = c - p (the same subtraction), S = sum |N_i t_i| and u = 2^-24, the three
// quotients t_i / |t| carry a relative error of at most u each and the two dot products at most 3.1 u S each (scaled by
// |t| > 0), so alignment |t| <= a + 7.2 u S; a < -(2^-20 S + 1e-30) leaves it negative (1e-30: products that
// underflow).  A NaN anywhere makes the comparison false: such entries take the shader's own arithmetic.
__device__ __forceinline__ bool turned_away(const V3 c, const V3 p, const V3 n) {
  const V3 t = sub3(c, p);
  const float a = dot3(n, t);
  const float S = __builtin_fmaf(__builtin_fabsf(n.z), __builtin_fabsf(t.z),
                                 __builtin_fmaf(__builtin_fabsf(n.y), __builtin_fabsf(t.y), __builtin_fabsf(n.x) * __builtin_fabsf(t.x)));
  return __builtin_fmaf(0x1p-20f, S, a) < -1.0e-30f;
}

// MODE 0: the production kernel.  MODE 1: also counts rays, tests, roots and
// wave trips (what the oracle counts, for parity checks and the bench's ray
// count).  MODE 2: production code plus per-phase cycle sums per wave (kept
// apart from MODE 1, whose extra registers would distort the timing).
// POOLED: the variant for scenes whose frames end in pooled passes (P.coop: many spheres); the other variant is the one
// that can fan lighting rays out (P.fanOut).  The host launches the one that matches; each carries only its own code.
// VOLUME false: the variant for scenes without a volume, compiled without the march (the registers it would hold
// through the shading sweep go to the rest of the kernel).
// SAMPLES: the variant for launches that draw several samples of every pixel (P.sampleCount > 1): the tickets are
// (tile, sample) pairs and every sample's colours go to a plane of their own (see FrameParams::sampleCount).
template <int MODE, bool LDS_OBJECTS, bool POOLED, bool VOLUME = true, bool SAMPLES = false>
__global__ void __launch_bounds__(1024) pt_render_kernel(const FrameParams P) {
  const bool poolingOn = POOLED && P.coop != 0;
  const bool fanOutOn = !POOLED && P.fanOut != 0;
  constexpr bool COUNT = MODE == 1;
  constexpr bool PHASES = MODE == 2;
  extern __shared__ float4 s_obj[];
  const int Ns = P.numSpheres, Ne = P.numEllipsoids;
  const int nGroups = (Ns + 7) >> 3; // trips of the sphere loop: 8 spheres = 8 float4
  const bool sphFilter = P.sphereFilter != 0;
  if (LDS_OBJECTS) {
    for (int i = threadIdx.x; i < P.imageFloat4s; i += blockDim.x) s_obj[i] = P.image[i];
    __syncthreads();
  }
  // the scene image is read-only for the whole launch: through the constant address space, so that
  // loads at wave-uniform addresses become scalar loads
  typedef float v4f __attribute__((ext_vector_type(4)));
  typedef const v4f __attribute__((address_space(4))) *ConstImage;
  const ConstImage g_img = (ConstImage)P.image;
  auto ld_image = [&](int i) -> float4 {
    const v4f v = g_img[i];
    return make_float4(v.x, v.y, v.z, v.w);
  };
#define PT_IMG(i) (LDS_OBJECTS ? s_obj[(i)] : ld_image(i))
#define PT_ELL(j, k) PT_IMG(P.offEll + 4 * (j) + (k))
#define PT_ELLP(p, k) ld_image(P.offEllPair + 8 * (p) + (k)) /* wave-uniform: scalar loads, like the sphere pairs */
#define PT_MAT(m, k) PT_IMG(P.offMat + 5 * (m) + (k))
#define PT_LIGHT(i) PT_IMG(P.offLight + (i))
#define PT_EMS(e, k) PT_IMG(P.offEmS + 3 * (e) + (k))
#define PT_EME(e, k) PT_IMG(P.offEmE + 3 * (e) + (k))
#define PT_MATIDX(i)                                                                        \
  ((int)(LDS_OBJECTS ? ((const unsigned short *)(s_obj + P.offMatIdx))[(i)]                 \
                     : ((const unsigned short __attribute__((address_space(4))) *)(g_img + P.offMatIdx))[(i)]))

  // Frame constants that are not needed in the intersection loops are read
  // from an LDS copy where they are used, instead of living in registers for
  // the whole kernel.
  FrameParams *const paramsLds =
      (FrameParams *)(s_obj + (LDS_OBJECTS ? P.imageFloat4s : 0) + 64 * (int)(blockDim.x >> 6));
  {
    const int __attribute__((address_space(4))) *ka =
        (const int __attribute__((address_space(4))) *)__builtin_amdgcn_kernarg_segment_ptr();
    for (int i = threadIdx.x; i < (int)(sizeof(FrameParams) / 4); i += blockDim.x) ((int *)paramsLds)[i] = ka[i];
    __syncthreads();
  }
  const FrameParams &L = *paramsLds;
  // Pooled tracing at the end of the frame (P.coop): control words, the pooled
  // rays (six arrays of blockDim.x floats) and one result slot per pooled ray.
  const int nThreads = (int)blockDim.x, nWaves = (int)(blockDim.x >> 6), waveInBlock = (int)(threadIdx.x >> 6);
  // density of a voxel byte: the 256 quotients b / 255 (the IEEE division the shader does per sample), tabulated
  float *const s_div255 = (float *)((char *)paramsLds + ((sizeof(FrameParams) + 15) & ~(size_t)15));
  if (VOLUME && P.numVolumes > 0)
    for (int i = threadIdx.x; i < 256; i += blockDim.x) s_div255[i] = (float)i / 255.0f;
  char *const coopBase = (char *)(s_div255 + 256);
  int *const c_ctrl = (int *)coopBase; // [0] switch, [1..2] pooled-ray count, [3..4] any ray left, [5..6] exact-ray count (by parity)
  // the control words are shared between waves: relaxed workgroup-scope atomics (LDS reads and writes that are not cached in registers)
#define PT_CTRL_LOAD(i) __hip_atomic_load(&c_ctrl[(i)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)
#define PT_CTRL_STORE(i, v) __hip_atomic_store(&c_ctrl[(i)], (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)
  float *const c_rays = (float *)(coopBase + 64);
  // Result slots of the pooled rays, one set per pass parity: a wave reads its results after the pass's second
  // barrier, and nothing orders that read against a faster wave's slot initialisation for the next pass (which
  // comes before that pass's first barrier), so consecutive passes must not share slots.  The other pooled
  // arrays are only touched between a pass's two barriers, or written after its second and read after the
  // next pass's first: the barriers order those.
  unsigned long long *const c_slot2 = (unsigned long long *)(coopBase + 64 + 6 * sizeof(float) * (size_t)nThreads);
  if (poolingOn && threadIdx.x < 8) PT_CTRL_STORE(threadIdx.x, 0);
  __syncthreads(); // the table and the control words are in place
  bool coop = false; // wave-uniform: this wave has joined the pooled mode
  int coopPass = 0;
  const int lane = threadIdx.x & 63;
  // per-wave scratch behind the object image: 128-entry candidate ring + 64 result slots (1 KiB)
  const int objFloat4s = LDS_OBJECTS ? P.imageFloat4s : 0;
  float4 *const w_area = s_obj + objFloat4s + 64 * (int)(threadIdx.x >> 6);
  unsigned int *const w_queue = (unsigned int *)w_area;
  unsigned long long *const w_slot = (unsigned long long *)(w_area + 32);

  // wave-uniform tile pool
  const int homeGroup = (int)(blockIdx.x & (kTicketGroups - 1));
  int groupsTried = 0; // ticket groups this wave has found drained
  int poolNext = 0, poolEnd = 0; // pixels [poolNext, poolEnd) of the tile at (poolTx, poolTy) are still to hand out
  int poolTx = 0, poolTy = 0;
  int poolS = 0; // SAMPLES: the sample (counted from L.sampleIndex) the pool's pixels are drawn for
  bool exhausted = false;
  unsigned int nClass2 = 0, nClass1 = 0; // drawing order: tickets [0,n2) class 2, [n2,n2+n1) class 1, rest class 0
  if (P.tileOrder) {
    nClass2 = (unsigned int)__builtin_amdgcn_readfirstlane((int)P.classCount[2]);
    nClass1 = (unsigned int)__builtin_amdgcn_readfirstlane((int)P.classCount[1]);
  }

  // per-lane pixel / path state
  int state = ST_IDLE;
  int outIdx = 0;
  int bounce = 0;
  V3 curD = mk3(0, 0, 1); // direction of the current path ray (its origin is rO while the path ray is in flight;
                          // the next one's is written to rO when it is cast)
  float thr = 1.0f;                            // throughput (always a multiple of (1,1,1))
  V3 fin = mk3(0, 0, 0);                       // finalColor
  // ray in flight
  V3 rO = mk3(0, 0, 0), rD = mk3(0, 0, 1);
  float rTmax = -1.0f; // tMin is EPSILON for every ray the shader casts
  // hit being shaded
  V3 hitP = mk3(0, 0, 0), hitN = mk3(0, 0, 1), viewDir = mk3(0, 0, 1);
  int hitMat = 0;
  V3 lit = mk3(0, 0, 0);     // calculateLighting's finalColor / toon finalColor
  V3 pending = mk3(0, 0, 0); // contribution added if the occlusion ray is clear
  int k = 0;                 // light index or emissive-list cursor
  // Fan-out at the end of the frame (P.fanOut): a lane in calculateLighting hands the shadow and
  // occlusion rays that follow its current one to idle lanes of the wave, one record each in LDS.
  int nRec = 0;              // records of this lane waiting for their results (one per round it took part in)
  int jobPos = 0;            // ST_JOB: the record this lane traces
  // A lane whose lighting rays are all with helpers casts its next path ray in the same pass (the bounce does not
  // depend on how the lighting ends): what calculateLighting's sum is still owed to finalColor is kept here and
  // added, with the operands the shader would use, once the helpers have reported (before the new hit is looked at).
  bool owedLit = false;      // finalColor += (lit * owedThr) * owedFactor is still to come
  float owedThr = 0.0f, owedFactor = 0.0f;
  float *const f_rec = (float *)(coopBase + (size_t)waveInBlock * kFanBytes); // 13 fields x 64 records, field-major
  unsigned long long *const f_roundMask = (unsigned long long *)(f_rec + 13 * 64); // owners that took part in round r
  int *const f_roundBase = (int *)(f_roundMask + kFanRounds);                      // first record of round r

  unsigned int cRays = 0, cPath = 0, cLight = 0, cEmis = 0, cShaded = 0, cVoxel = 0;
  unsigned int cRoots = 0, cERoots = 0; // lane-level root evaluations (discriminant >= 0)
  unsigned int cGroups = 0, cSlow = 0;  // wave-level: 8-sphere trips, trips with at least one root
  unsigned int cExact = 0;              // rays this lane sent to the exact routine
  unsigned int cTests = 0, cETests = 0; // (ray, sphere) and (ray, ellipsoid) pairs this lane put through a test, counted where the test runs
  unsigned long long tPooled = 0, nPooled = 0, nPooledRays = 0;
  unsigned long long cycRefill = 0, cycSpheres = 0, cycExact = 0, cycEll = 0, cycShade = 0, tPhase = 0; // wave-level s_memtime sums
  unsigned long long cycTicket = 0, nTickets = 0, tExhausted = 0, tStart = 0;
  unsigned long long rtStart = 0, rtDry = 0; // s_memrealtime stamps (one clock for the whole chip, unlike s_memtime)
  unsigned int nPasses = 0, passesAtDry = 0;
  // MODE 2 always stamps a wave's start, its finding the tickets drained and its exit (three stamps per wave: the
  // timeline of the production code); the per-phase and per-ticket stamps, which cost a scalar-memory round trip
  // each and slow short passes down noticeably, only with tune bit 26.
  const bool phaseStamps = PHASES && (P.tune & (1 << 26)) != 0;
#define PT_PHASE(ACC) if (PHASES && phaseStamps) { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); ACC += now_ - tPhase; tPhase = now_; }
  if (PHASES) tStart = tPhase = __builtin_amdgcn_s_memtime();
  if (PHASES && P.waveLog) rtStart = __builtin_amdgcn_s_memrealtime();

  for (;;) {
    if (PHASES) nPasses++;
    // ------------------------------------------------------------------
    // 1. refill idle lanes from the wave's tile pool
    // ------------------------------------------------------------------
    // Idle lanes take the next pixels of the pool in lane order; when the pool
    // runs out part-way, the rest continue in the next tile (fetched first).
    const unsigned long long idleMask = ballot64(state == ST_IDLE);
    // The refill has a fixed cost per pass, whatever the number of lanes it fills; where a pass over the objects is
    // cheap next to it (few objects), it waits until P.refillMin lanes are idle (or nothing else is left to do).
    if (idleMask != 0ull && !(exhausted && poolNext == poolEnd) &&
        ((int)__popcll(idleMask) >= P.refillMin || idleMask == ~0ull)) {
      const int nIdle = __popcll(idleMask);
      const int avail = poolEnd - poolNext;
      int nextBase = -1; // first pixel of the following tile, if one is needed and left
      int nextTx = 0, nextTy = 0, nextS = 0;
      if (nIdle > avail && !exhausted) {
        // tickets: one per (tile, sample) pair of the launch, the samples of a tile next to each other
        const unsigned int nSamples = SAMPLES ? (unsigned int)__builtin_amdgcn_readfirstlane(L.sampleCount) : 1u;
        const unsigned int numTickets = (unsigned int)P.numTiles * nSamples;
        // Tickets come from kTicketGroups counters instead of one: group g owns
        // tickets g, g + G, g + 2G, ...  A single counter would see one same-address
        // atomic per tile from every wave of the chip, and those queue for
        // microseconds each.  A wave draws from its own group until that is
        // drained, then from the next group that still has tickets (peeked with a
        // plain load first, so drained groups cost no atomics).
        unsigned int t = 0xFFFFFFFFu;
        unsigned long long tTicket = 0;
        if (PHASES && phaseStamps) { tTicket = __builtin_amdgcn_s_memtime(); nTickets++; }
        while (groupsTried < kTicketGroups) {
          const int g = (homeGroup + groupsTried) & (kTicketGroups - 1);
          const unsigned int owned = (numTickets + (unsigned int)(kTicketGroups - 1 - g)) / kTicketGroups;
          unsigned int *ctr = P.tileCounter + g * kTicketStride;
          bool drained = owned == 0u;
          if (!drained && groupsTried > 0)
            drained = (unsigned int)__builtin_amdgcn_readfirstlane((int)*(volatile unsigned int *)ctr) >= owned;
          if (!drained) {
            unsigned int a = 0;
            if (lane == 0) a = atomicAdd(ctr, 1u);
            a = (unsigned int)__builtin_amdgcn_readfirstlane((int)a);
            if (a < owned) { t = (unsigned int)g + (unsigned int)kTicketGroups * a; break; }
          }
          ++groupsTried;
        }
        if (t >= numTickets) {
          exhausted = true;
          if (PHASES) tExhausted = __builtin_amdgcn_s_memtime();
          if (PHASES && P.waveLog) { rtDry = __builtin_amdgcn_s_memrealtime(); passesAtDry = nPasses; }
        } else {
          if (SAMPLES) {
            const unsigned int q = t / nSamples;
            nextS = (int)(t - q * nSamples);
            t = q;
          }
          if (P.tileOrder) {
            const int at = t < nClass2 ? 2 * P.numTiles + (int)t
                           : t < nClass2 + nClass1 ? P.numTiles + (int)(t - nClass2) : (int)(t - nClass2 - nClass1);
            t = (unsigned int)__builtin_amdgcn_readfirstlane(P.tileOrder[at]);
          }
          nextBase = (int)t * kTilePixels;
          nextTx = (int)(t % (unsigned int)P.tilesX); // once per tile, wave-uniform
          nextTy = (int)(t / (unsigned int)P.tilesX);
        }
        if (PHASES && phaseStamps) cycTicket += __builtin_amdgcn_s_memtime() - tTicket;
      }
      const int rank = lanes_below(idleMask);
      int p = -1; // pixel within its tile
      bool fromNext = false;
      if (state == ST_IDLE) {
        if (rank < avail) p = (poolNext & (kTilePixels - 1)) + rank;
        else if (nextBase >= 0) { p = rank - avail; fromNext = true; } // nIdle <= 64 = one tile
      }
      if (p >= 0) {
        int x, ly, gy;
        if (tile_pixel_xy(L, fromNext ? nextTx : poolTx, fromNext ? nextTy : poolTy, p, x, ly, gy)) {
          // main() + createCameraRay (raytrace.comp:686-700, 121-140)
          const int sample = SAMPLES ? (fromNext ? nextS : poolS) : 0;
          curD = camera_dir(L, x, gy, L.sampleIndex + sample);
          outIdx = ly * L.width + x;
          if (SAMPLES) outIdx += sample * L.planeStride;
          fin = mk3(0, 0, 0);
          thr = 1.0f;
          bounce = 0;
          rO = ld3(L.cam); rD = curD; rTmax = kMaxDistance;
          state = ST_PATH;
          if (COUNT) cPath++;
        }
      }
      if (nIdle > avail) {
        if (nextBase >= 0) {
          poolNext = nextBase + (nIdle - avail);
          poolEnd = nextBase + kTilePixels;
          poolTx = nextTx;
          poolTy = nextTy;
          if (SAMPLES) poolS = nextS;
        }
        else poolNext = poolEnd;
      } else {
        poolNext += nIdle;
      }
    }
    if (poolingOn && !coop) {
      // The first wave of the workgroup that finds every ticket group drained asks
      // all of them to switch; each joins at the top of its next pass.
      if (exhausted && lane == 0) PT_CTRL_STORE(0, 1);
      if (__builtin_amdgcn_readfirstlane(PT_CTRL_LOAD(0)) != 0) {
        coop = true;
        if (PHASES) tPooled = __builtin_amdgcn_s_memtime();
      }
    }
    if (!coop && ballot64(state != ST_IDLE) == 0ull) break;
    // Waves that carry bounced paths, or that are finishing their last pixels,
    // hold the longest chains of dependent rays: let them issue first.
    if (!(P.tune & 1)) {
      if (exhausted || ballot64(state != ST_IDLE && bounce >= 1) != 0ull) __builtin_amdgcn_s_setprio(3);
      else __builtin_amdgcn_s_setprio(0);
    }
    PT_PHASE(cycRefill)

    // ------------------------------------------------------------------
    // 2. traceRay for every live lane (raytrace.comp:212-241); bestIdx < Ns is
    //    a sphere, else an ellipsoid.
    //
    //    Spheres.  The wave-uniform loop only computes discriminants (two
    //    spheres per packed instruction) and tests their sign.  A lane whose
    //    discriminant is non-negative appends (lane, sphere) to the wave's LDS
    //    queue; whenever 64 entries are waiting the whole wave evaluates them,
    //    one entry per lane: exact roots, then a 64-bit LDS atomic-min per
    //    owning lane on (t, ~index).  For finite rays hitSphere's result does
    //    not depend on the order of the spheres (smallest accepted t wins, an
    //    exact tie goes to the later sphere), so the deferred evaluation gives
    //    the same winner as the shader's sequential loop.  Rays with
    //    non-finite or huge coordinates, where NaNs make the shader's loop
    //    order-dependent, take the exact routine below instead.
    // ------------------------------------------------------------------
    const bool live = state != ST_IDLE && state != ST_WAIT;
    float bestT = rTmax;
    int bestIdx = -1;
    bool qLane = false; // this lane's ellipsoid roots may be deferred (set below)
    float dLen2q = 0.0f; // |d|^2 of this lane's ray, for the ellipsoid filter's margin
    {
      const float tMin = kEpsilon; // every ray of the shader uses tMin = EPSILON
      const float kBig = 1.0e9f;   // |coordinates| up to here cannot overflow the discriminant
      const bool finiteRay = __builtin_fabsf(rO.x) <= kBig && __builtin_fabsf(rO.y) <= kBig &&
                             __builtin_fabsf(rO.z) <= kBig && __builtin_fabsf(rD.x) <= kBig &&
                             __builtin_fabsf(rD.y) <= kBig && __builtin_fabsf(rD.z) <= kBig;
      // hitSphere assumes a unit direction.  After a few mirror bounces the
      // shader's own rounding feedback can leave |d| far from 1; such a ray has
      // a non-negative discriminant for most spheres, so it is cheaper to hand
      // it to the exact routine than to let it flood the candidate queue.  (Only
      // a question of speed: the dense loop computes hitSphere's discriminant
      // whatever |d| is.)
      const float dLen2 = dot3(rD, rD);
      dLen2q = dLen2;
      const bool unitDir = dLen2 - 1.0f <= P.unitTol && 1.0f - dLen2 <= P.unitTolLow;
      // With only a few rays left in the wave (the end of the frame), the dense
      // loop would still cost a full pass; those rays go through the exact
      // routine instead, whose cost is per ray.
      const bool sparse = !coop && (int)__popcll(ballot64(live)) <= P.sparseLanes;
      const bool fastLane = live && finiteRay && unitDir && P.sceneHuge == 0 && !sparse;
      const bool special = live && !fastLane;
      // lanes whose ellipsoid roots can be deferred too: finite ray with a usable direction
      // ... and magnitudes for which no intermediate of hitEllipsoid can overflow (ellScale bounds
      // 1/radii times the growth under the rotation over all ellipsoids; it is infinite when
      // an ellipsoid has a zero, non-finite or huge parameter).  Other rays keep the shader's
      // sequential loop below, where an infinity or a NaN does what it does in the shader.
      const float oMax = __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(rO.x), __builtin_fabsf(rO.y)), __builtin_fabsf(rO.z));
      const float dMax = __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(rD.x), __builtin_fabsf(rD.y)), __builtin_fabsf(rD.z));
      qLane = live && finiteRay && P.sceneHuge == 0 && dLen2 >= 1.0e-30f &&
              (oMax + P.ellCenterMax) * P.ellScale <= 1.0e8f && dMax * P.ellScale <= 1.0e8f;
      if (COUNT && live) cRays++;
      if (COUNT && special) cExact++;
      // slot[lane] = (bits(t) << 32) | ~(index + 1) of the best accepted object so
      // far; t >= tMin > 0, so unsigned order on the bits is float order, and on
      // equal t the larger index has the smaller key.  Start at (tMax, none):
      // acceptance is t <= tMax.
      unsigned long long slotInit = 0ull; // tMax < tMin: nothing can be accepted
      if (rTmax >= tMin) slotInit = ((unsigned long long)__float_as_uint(rTmax) << 32) | 0xFFFFFFFFull;
      w_slot[lane] = slotInit;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");

      // The dense sphere pass for the 64 rays (ox..dz, one per lane) over sphere
      // groups [gBegin, gEnd), results into slot[lane].  A lane without a ray
      // passes one that can never produce a candidate (origin 4e9, direction 0).
      // scalarTag: read the sphere pairs with scalar loads from the global copy of the image
      // (the bulk of the frame: no LDS instruction in the loop, operands in SGPRs; the latency is
      // hidden by the other waves) or from the LDS copy (pooled end of the frame, where few
      // waves run and the short LDS latency counts).
      auto denseSpheres = [&](auto scalarTag, const bool has, const float ox, const float oy, const float oz, const float dx,
                              const float dy, const float dz, unsigned long long *slot, const int gBegin, const int gEnd) {
        int qHead = 0, qTail = 0; // ring of 128 entries, wave-uniform cursors
        // Evaluate n <= 64 queued (lane, sphere) candidates, one per lane.
        auto drain = [&](int n) {
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
          const unsigned int e = w_queue[(qHead + lane) & 127];
          const bool on = lane < n;
          const int owner = on ? (int)(e & 63u) : lane;
          const int idx = on ? (int)(e >> 6) : 0;
          const float eox = __shfl(ox, owner), eoy = __shfl(oy, owner), eoz = __shfl(oz, owner);
          const float edx = __shfl(dx, owner), edy = __shfl(dy, owner), edz = __shfl(dz, owner);
          float cx, cy, cz, rr;
          if (LDS_OBJECTS) { // the pair image starts at float4 0 of the scene image
            const float *sf = (const float *)s_obj;
            const int f = 8 * (idx >> 1) + (idx & 1);
            cx = sf[f]; cy = sf[f + 2]; cz = sf[f + 4]; rr = sf[4 * P.offRadius + idx];
          } else {
            const float4 sp = P.spheres[idx];
            cx = sp.x; cy = sp.y; cz = sp.z; rr = sp.w;
          }
          // hitSphere (raytrace.comp:142-159), same operations as the packed loop
          const float ocx = eox - cx, ocy = eoy - cy, ocz = eoz - cz;
          const float hb = __builtin_fmaf(ocz, edz, __builtin_fmaf(ocy, edy, ocx * edx));
          const float dd = __builtin_fmaf(ocz, ocz, __builtin_fmaf(ocy, ocy, ocx * ocx));
          const float cc = __builtin_fmaf(-rr, rr, dd);
          const float disc = __builtin_fmaf(hb, hb, -cc);
          const float sq = __builtin_sqrtf(disc);
          float t = -hb - sq;
          if (t < tMin) t = -hb + sq;
          // The packed loop's filter may let a sphere through whose discriminant is negative.
          const bool root = on && disc >= 0.0f;
          if (COUNT && root) cRoots++;
          // accepted iff tMin <= t <= closest-so-far; the atomic min applies the upper bound
          if (root && !(t < tMin)) {
            const unsigned long long key =
                ((unsigned long long)__float_as_uint(t) << 32) | (unsigned long long)(~((unsigned int)idx + 1u));
            atomicMin(&slot[owner], key);
          }
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        };
        const f2 OX = mk2(ox, ox), OY = mk2(oy, oy), OZ = mk2(oz, oz);
        const f2 DX = mk2(dx, dx), DY = mk2(dy, dy), DZ = mk2(dz, dz);
#define PT_PAIR_DISC(A, B, DISC)                                                 \
  {                                                                              \
    const f2 ocx = OX - mk2(A.x, A.y), ocy = OY - mk2(A.z, A.w), ocz = OZ - mk2(B.x, B.y); \
    const f2 rr = mk2(B.z, B.w);                                                 \
    const f2 hb = fma2(ocz, DZ, fma2(ocy, DY, ocx * DX));                        \
    const f2 dd = fma2(ocz, ocz, fma2(ocy, ocy, ocx * ocx));                     \
    const f2 cc = fma2(-rr, rr, dd);                                             \
    DISC = fma2(hb, hb, -cc);                                                    \
  }
// The cheaper conservative test (see FrameParams::sphereFilter): 8 packed instructions per pair,
// each with one scalar operand (k is added last: an instruction reads one scalar register pair).
#define PT_PAIR_FILT(A, B, T)                                                    \
  {                                                                              \
    const f2 cx = mk2(A.x, A.y), cy = mk2(A.z, A.w), cz = mk2(B.x, B.y);         \
    const f2 hb = fma2(cx, NDX, fma2(cy, NDY, fma2(cz, NDZ, OD)));               \
    const f2 oc2 = fma2(cx, M2OX, fma2(cy, M2OY, cz * M2OZ));                    \
    T = fma2(hb, hb, -oc2) - mk2(B.z, B.w);                                      \
  }
#define PT_ENQUEUE(IDX, DISC)                                                    \
  {                                                                              \
    /* (IDX) < Ns is wave-uniform: kept out of the per-lane compare */           \
    const unsigned long long m_ = ((IDX) < Ns) ? ballot64(DISC >= thr) : 0ull;   \
    if (m_ != 0ull) {                                                            \
      const int pre_ = lanes_below(m_);                                          \
      if (DISC >= thr) w_queue[(qTail + pre_) & 127] = (unsigned int)lane | ((unsigned int)(IDX) << 6); \
      qTail += __popcll(m_);                                                     \
      if (qTail - qHead >= 64) { drain(64); qHead += 64; }                       \
    }                                                                            \
  }
        // Per ray: o.d, -d, -2o and the threshold |o|^2 (1 - 2^-17) for the filter; 0 for the discriminant itself.
        const float od1 = __builtin_fmaf(oz, dz, __builtin_fmaf(oy, dy, ox * dx));
        const float oo1 = __builtin_fmaf(oz, oz, __builtin_fmaf(oy, oy, ox * ox));
        const f2 OD = mk2(od1, od1), NDX = mk2(-dx, -dx), NDY = mk2(-dy, -dy), NDZ = mk2(-dz, -dz);
        const f2 M2OX = mk2(-2.0f * ox, -2.0f * ox), M2OY = mk2(-2.0f * oy, -2.0f * oy), M2OZ = mk2(-2.0f * oz, -2.0f * oz);
        auto loop = [&](auto filterTag) {
          constexpr bool FILTER = decltype(filterTag)::value;
          const float thr = !FILTER ? 0.0f : (has ? __builtin_fmaf(-0x1p-17f, oo1, oo1) : __builtin_inff());
#define PT_PAIR_AT(i) ((decltype(scalarTag)::value || !LDS_OBJECTS) ? ld_image(i) : s_obj[(i)])
#define PT_LOAD8(A0, B0, A1, B1, A2, B2, A3, B3, G)                                                                      \
  A0 = PT_PAIR_AT(8 * (G)); B0 = PT_PAIR_AT(8 * (G) + 1); A1 = PT_PAIR_AT(8 * (G) + 2); B1 = PT_PAIR_AT(8 * (G) + 3);     \
  A2 = PT_PAIR_AT(8 * (G) + 4); B2 = PT_PAIR_AT(8 * (G) + 5); A3 = PT_PAIR_AT(8 * (G) + 6); B3 = PT_PAIR_AT(8 * (G) + 7);
          // one trip: eight spheres against the wave's rays
          auto trip = [&](const int g, const float4 a0, const float4 b0, const float4 a1, const float4 b1, const float4 a2,
                          const float4 b2, const float4 a3, const float4 b3) {
            f2 d0, d1, d2, d3;
            if (FILTER) {
              PT_PAIR_FILT(a0, b0, d0)
              PT_PAIR_FILT(a1, b1, d1)
              PT_PAIR_FILT(a2, b2, d2)
              PT_PAIR_FILT(a3, b3, d3)
            } else {
              PT_PAIR_DISC(a0, b0, d0)
              PT_PAIR_DISC(a1, b1, d1)
              PT_PAIR_DISC(a2, b2, d2)
              PT_PAIR_DISC(a3, b3, d3)
            }
            // finite rays and a finite scene give finite values (no NaN to lose in max)
            // (three three-input maxima and one of two)
            float m = __builtin_fmaxf(__builtin_fmaxf(d0.x, d0.y), d1.x);
            m = __builtin_fmaxf(__builtin_fmaxf(m, d1.y), d2.x);
            m = __builtin_fmaxf(__builtin_fmaxf(m, d2.y), d3.x);
            m = __builtin_fmaxf(m, d3.y);
            if (COUNT) cGroups++;
            if (COUNT && has) cTests += (unsigned int)(Ns - 8 * g < 8 ? Ns - 8 * g : 8); // the padding copies are no spheres
            if (ballot64(m >= thr) != 0ull) {
              if (COUNT) cSlow++;
              PT_ENQUEUE(8 * g, d0.x)
              PT_ENQUEUE(8 * g + 1, d0.y)
              PT_ENQUEUE(8 * g + 2, d1.x)
              PT_ENQUEUE(8 * g + 3, d1.y)
              PT_ENQUEUE(8 * g + 4, d2.x)
              PT_ENQUEUE(8 * g + 5, d2.y)
              PT_ENQUEUE(8 * g + 6, d3.x)
              PT_ENQUEUE(8 * g + 7, d3.y)
            }
          };
          if (POOLED && decltype(scalarTag)::value) { // many spheres: worth the second set of registers
            // Two sets of scalar registers take turns: the next trip's spheres are requested as soon as this trip's
            // have arrived (scalar loads return in any order, so a wait is a wait for all of them: it has to come
            // before the next request goes out), and arrive while this trip computes.
            float4 pa0, pb0, pa1, pb1, pa2, pb2, pa3, pb3, qa0, qb0, qa1, qb1, qa2, qb2, qa3, qb3;
            if (gBegin < gEnd) { PT_LOAD8(pa0, pb0, pa1, pb1, pa2, pb2, pa3, pb3, gBegin) }
            for (int g = gBegin; g < gEnd; g += 2) {
              const int g1 = g + 1 < gEnd ? g + 1 : g;
              __builtin_amdgcn_s_waitcnt(0xC07F); // lgkmcnt(0): the p set is there
              PT_LOAD8(qa0, qb0, qa1, qb1, qa2, qb2, qa3, qb3, g1)
              __builtin_amdgcn_sched_barrier(0);
              trip(g, pa0, pb0, pa1, pb1, pa2, pb2, pa3, pb3);
              if (g + 1 >= gEnd) break;
              const int g2 = g + 2 < gEnd ? g + 2 : g + 1;
              __builtin_amdgcn_s_waitcnt(0xC07F); // the q set is there
              PT_LOAD8(pa0, pb0, pa1, pb1, pa2, pb2, pa3, pb3, g2)
              __builtin_amdgcn_sched_barrier(0);
              trip(g + 1, qa0, qb0, qa1, qb1, qa2, qb2, qa3, qb3);
            }
          } else {
            for (int g = gBegin; g < gEnd; ++g) {
              float4 a0, b0, a1, b1, a2, b2, a3, b3;
              PT_LOAD8(a0, b0, a1, b1, a2, b2, a3, b3, g)
              trip(g, a0, b0, a1, b1, a2, b2, a3, b3);
            }
          }
#undef PT_LOAD8
#undef PT_PAIR_AT
        };
        if (sphFilter) loop(ScalarTag{}); else loop(LdsTag{});
        if (qTail > qHead) drain(qTail - qHead);
      };

      // Exact routine, finite case: one ray against every sphere, one sphere per
      // lane per chunk of 64 (coalesced 16-byte loads from the plain array).  A
      // finite ray in a finite scene cannot produce a NaN, so the result is the
      // smallest root >= tMin (later sphere on a tie) if it is <= tMax: each lane
      // keeps the best of its own spheres and one reduction picks the winner.
      // Returns the slot key of the winner, or all ones for none.
      auto exactFinite = [&](const float eox, const float eoy, const float eoz, const float edx, const float edy,
                             const float edz, const float etmax, unsigned int &roots) -> unsigned long long {
        float myBest = __builtin_inff();
        int myIdx = -1;
        // four chunks per trip: the loads go out together and the four root
        // chains are independent, which hides their latency
        for (int base = 0; base < Ns; base += 256) {
          float4 sp4[4];
          bool in4[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const int j = base + 64 * u + lane;
            in4[u] = j < Ns;
            sp4[u] = P.spheres[in4[u] ? j : 0];
          }
          float hb4[4], disc4[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const float ocx = eox - sp4[u].x, ocy = eoy - sp4[u].y, ocz = eoz - sp4[u].z;
            hb4[u] = __builtin_fmaf(ocz, edz, __builtin_fmaf(ocy, edy, ocx * edx));
            const float dd = __builtin_fmaf(ocz, ocz, __builtin_fmaf(ocy, ocy, ocx * ocx));
            const float cc = __builtin_fmaf(-sp4[u].w, sp4[u].w, dd);
            disc4[u] = __builtin_fmaf(hb4[u], hb4[u], -cc);
          }
          float sq4[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) sq4[u] = __builtin_sqrtf(disc4[u]);
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const bool root = in4[u] && !(disc4[u] < 0.0f);
            const float t0 = -hb4[u] - sq4[u], t1 = -hb4[u] + sq4[u];
            const float v = (t0 < tMin) ? t1 : t0;
            if (root && !(v < tMin) && v <= myBest) { myBest = v; myIdx = base + 64 * u + lane; }
            if (COUNT) roots += root ? 1u : 0u;
            if (COUNT && in4[u]) cTests++;
          }
        }
        unsigned long long key = ((unsigned long long)__float_as_uint(myBest) << 32) |
                                 (unsigned long long)(~((unsigned int)myIdx + 1u));
        for (int off = 32; off > 0; off >>= 1) {
          const unsigned long long other = __shfl_xor(key, off, 64);
          key = other < key ? other : key;
        }
        if (COUNT) for (int off = 32; off > 0; off >>= 1) roots += __shfl_xor(roots, off, 64);
        const float mn = __uint_as_float((unsigned int)(key >> 32));
        const unsigned int widx = ~(unsigned int)key - 1u;
        return (widx < (unsigned int)Ns && !(mn > etmax)) ? key : ~0ull;
      };

      bool pooledX = false; // this lane's exact-routine ray is in the workgroup's list
      int myXPos = 0;
      if (coop) {
        // Pooled pass.  Every wave of the workgroup appends its fast-path rays to
        // one list; the list is cut into batches of 64 and each batch into as
        // many sphere ranges as there are waves to go round, so a pass costs in
        // proportion to the rays that are left instead of one full sphere loop
        // per wave that still holds a ray.
        const int par = coopPass & 1;
        unsigned long long *const c_slot = c_slot2 + (size_t)par * nThreads;
        ++coopPass;
        if (PHASES && waveInBlock == 0) nPooled++;
        if (ballot64(live) != 0ull && lane == 0) PT_CTRL_STORE(3 + par, 1);
        const unsigned long long fm = ballot64(fastLane);
        int base = 0;
        if (fm != 0ull) {
          if (lane == 0) base = atomicAdd(&c_ctrl[1 + par], (int)__popcll(fm));
          base = __builtin_amdgcn_readfirstlane(base);
        }
        const int myPos = base + lanes_below(fm);
        if (fastLane) {
          c_rays[myPos] = rO.x; c_rays[nThreads + myPos] = rO.y; c_rays[2 * nThreads + myPos] = rO.z;
          c_rays[3 * nThreads + myPos] = rD.x; c_rays[4 * nThreads + myPos] = rD.y; c_rays[5 * nThreads + myPos] = rD.z;
          c_slot[myPos] = slotInit;
        }
        // finite rays for the exact routine go to a second list (up to nThreads / 4 of them)
        const int xCap = nThreads >> 2;
        float *const c_xrays = (float *)(c_slot2 + 2 * nThreads);
        unsigned long long *const c_xslot = (unsigned long long *)(c_xrays + 7 * xCap);
        {
          const bool want = special && qLane;
          const unsigned long long xm = ballot64(want);
          int xb = 0;
          if (xm != 0ull) {
            if (lane == 0) xb = atomicAdd(&c_ctrl[5 + par], (int)__popcll(xm));
            xb = __builtin_amdgcn_readfirstlane(xb);
          }
          myXPos = xb + lanes_below(xm);
          pooledX = want && myXPos < xCap;
          if (pooledX) {
            c_xrays[myXPos] = rO.x; c_xrays[xCap + myXPos] = rO.y; c_xrays[2 * xCap + myXPos] = rO.z;
            c_xrays[3 * xCap + myXPos] = rD.x; c_xrays[4 * xCap + myXPos] = rD.y; c_xrays[5 * xCap + myXPos] = rD.z;
            c_xrays[6 * xCap + myXPos] = rTmax;
          }
        }
        __syncthreads();
        if (PT_CTRL_LOAD(3 + par) == 0) break; // no wave of the workgroup holds a ray: all leave together
        if (threadIdx.x == 0) { PT_CTRL_STORE(1 + (par ^ 1), 0); PT_CTRL_STORE(3 + (par ^ 1), 0); PT_CTRL_STORE(5 + (par ^ 1), 0); }
        const int pooled = __builtin_amdgcn_readfirstlane(PT_CTRL_LOAD(1 + par));
        const int batches = (pooled + 63) >> 6;
        if (PHASES && waveInBlock == 0) nPooledRays += (unsigned long long)pooled;
        if (batches > 0) {
          const int split = nWaves / batches > 0 ? nWaves / batches : 1;
          const int units = batches * split;
          for (int u = waveInBlock; u < units; u += nWaves) {
            const int b = u / split, part = u - b * split;
            const int at = b * 64 + lane;
            const bool has = at < pooled;
            const float ox = has ? c_rays[at] : 4.0e9f, oy = has ? c_rays[nThreads + at] : 4.0e9f;
            const float oz = has ? c_rays[2 * nThreads + at] : 4.0e9f, dx = has ? c_rays[3 * nThreads + at] : 0.0f;
            const float dy = has ? c_rays[4 * nThreads + at] : 0.0f, dz = has ? c_rays[5 * nThreads + at] : 0.0f;
            denseSpheres(LdsTag{}, has, ox, oy, oz, dx, dy, dz, c_slot + b * 64, (int)((long long)part * nGroups / split),
                         (int)((long long)(part + 1) * nGroups / split));
          }
        }
        {
          int xCount = __builtin_amdgcn_readfirstlane(PT_CTRL_LOAD(5 + par));
          if (xCount > xCap) xCount = xCap;
          for (int e = waveInBlock; e < xCount; e += nWaves) {
            unsigned int roots = 0;
            const unsigned long long key =
                exactFinite(c_xrays[e], c_xrays[xCap + e], c_xrays[2 * xCap + e], c_xrays[3 * xCap + e],
                            c_xrays[4 * xCap + e], c_xrays[5 * xCap + e], c_xrays[6 * xCap + e], roots);
            if (lane == 0) {
              c_xslot[e] = key;
              if (COUNT) cRoots += roots;
            }
          }
        }
        __syncthreads();
        if (fastLane) w_slot[lane] = c_slot[myPos];
        if (pooledX) {
          const unsigned long long key = c_xslot[myXPos];
          if (key != ~0ull) w_slot[lane] = key;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      } else if (ballot64(fastLane) != 0ull) {
        denseSpheres(ScalarTag{}, fastLane, fastLane ? rO.x : 4.0e9f, fastLane ? rO.y : 4.0e9f, fastLane ? rO.z : 4.0e9f,
                     fastLane ? rD.x : 0.0f, fastLane ? rD.y : 0.0f, fastLane ? rD.z : 0.0f, w_slot, 0, nGroups);
      }

      // Exact routine for the few rays the fast path must not (non-finite) or
      // should not (non-unit direction) take: the 64 lanes each take one sphere
      // of a chunk and the shader's sequential acceptance (including its NaN
      // behaviour: a NaN t is accepted and then voids the upper bound for the
      // next sphere) is reproduced chunk by chunk.
      PT_PHASE(cycSpheres)
      unsigned long long todo = ballot64(special && !pooledX);
      const unsigned long long finiteMask = (P.sceneHuge == 0) ? ballot64(finiteRay) : 0ull;
      while (todo != 0ull) {
        const int owner = (int)__builtin_ctzll(todo);
        todo &= todo - 1ull;
        const float eox = __shfl(rO.x, owner), eoy = __shfl(rO.y, owner), eoz = __shfl(rO.z, owner);
        const float edx = __shfl(rD.x, owner), edy = __shfl(rD.y, owner), edz = __shfl(rD.z, owner);
        const float etmax = __shfl(rTmax, owner);
        float C = etmax; // closestT
        int win = -1;
        unsigned int roots = 0;
        if ((finiteMask >> owner) & 1ull) {
          const unsigned long long key = exactFinite(eox, eoy, eoz, edx, edy, edz, etmax, roots);
          if (key != ~0ull) {
            C = __uint_as_float((unsigned int)(key >> 32));
            win = (int)(~(unsigned int)key - 1u);
          }
        } else {
          // The shader's sequential acceptance, NaN behaviour included (a NaN t is
          // accepted and then voids the upper bound for the next sphere),
          // reproduced chunk by chunk.
          for (int base = 0; base < Ns; base += 64) {
            const int j = base + lane;
            const bool inb = j < Ns;
            const float4 sp = P.spheres[inb ? j : 0];
            const float ocx = eox - sp.x, ocy = eoy - sp.y, ocz = eoz - sp.z;
            const float hb = __builtin_fmaf(ocz, edz, __builtin_fmaf(ocy, edy, ocx * edx));
            const float dd = __builtin_fmaf(ocz, ocz, __builtin_fmaf(ocy, ocy, ocx * ocx));
            const float cc = __builtin_fmaf(-sp.w, sp.w, dd);
            const float disc = __builtin_fmaf(hb, hb, -cc);
            const bool root = inb && !(disc < 0.0f);
            const float sq = __builtin_sqrtf(disc);
            const float t0 = -hb - sq, t1 = -hb + sq;
            // a NaN root passes both range tests of hitSphere whatever closestT is
            const bool nanacc = root && ((t0 != t0) || ((t0 < tMin) && (t1 != t1)));
            float v = t0;
            bool valid = root && !nanacc;
            if (t0 < tMin) { v = t1; valid = valid && !(t1 < tMin); }
            if (COUNT) roots += (unsigned int)__popcll(ballot64(root));
            if (COUNT && inb) cTests++;
            const unsigned long long nm = ballot64(nanacc);
            if (nm != 0ull) {
              const int kl = 63 - (int)__builtin_clzll(nm);
              C = __builtin_nanf("");
              win = base + kl;
              valid = valid && lane > kl;
            }
            // accepted iff !(v > closestT); with closestT = NaN every valid root is
            const bool cand = valid && !(v > C);
            if (ballot64(cand) != 0ull) {
              float mn = cand ? v : __builtin_inff();
              for (int off = 32; off > 0; off >>= 1) mn = __builtin_fminf(mn, __shfl_xor(mn, off, 64));
              const unsigned long long em = ballot64(cand && v == mn);
              C = mn;
              win = base + 63 - (int)__builtin_clzll(em);
            }
          }
        }
        if (lane == owner) {
          if (qLane) { // finite: C is a plain t >= tMin; the ellipsoid queue continues from it
            if (win >= 0)
              w_slot[lane] = ((unsigned long long)__float_as_uint(C) << 32) | (unsigned long long)(~((unsigned int)win + 1u));
          } else {
            bestT = C;
            bestIdx = win;
          }
          if (COUNT) cRoots += roots;
        }
      }
      PT_PHASE(cycExact)

      // hitEllipsoid (raytrace.comp:171-210), after the spheres as in traceRay.
      // Same scheme as the spheres: a wave-uniform loop selects candidates (never too few), the queue
      // evaluates them 64 at a time with hitEllipsoid's own operations.
      // (a = |dirScaled|^2 is part of the test, so non-unit directions are fine.)
      int qHead = 0, qTail = 0; // the candidate ring is empty again
      auto drainE = [&](int n) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const unsigned int e = w_queue[(qHead + lane) & 127];
        const bool on = lane < n;
        const int owner = on ? (int)(e & 63u) : lane;
        const int j = on ? (int)(e >> 6) - Ns : 0;
        const V3 eo = mk3(__shfl(rO.x, owner), __shfl(rO.y, owner), __shfl(rO.z, owner));
        const V3 ed = mk3(__shfl(rD.x, owner), __shfl(rD.y, owner), __shfl(rD.z, owner));
        const float4 e0 = PT_ELL(j, 0), e1 = PT_ELL(j, 1), e2 = PT_ELL(j, 2), e3 = PT_ELL(j, 3);
        const V3 nq = xyz(e2), nq2 = xyz(e3);
        const V3 oc = rotate_q(sub3(eo, xyz(e0)), nq, nq2, e0.w);
        const V3 dir = rotate_q(ed, nq, nq2, e0.w);
        const V3 invR = xyz(e1);
        const V3 ocS = mul3(oc, invR), dS = mul3(dir, invR);
        const float a = dot3(dS, dS);
        const float hb = dot3(ocS, dS);
        const float c = dot3(ocS, ocS) - 1.0f;
        const float disc = __builtin_fmaf(hb, hb, -(a * c));
        const float sq = __builtin_sqrtf(disc);
        float t = (-hb - sq) / a;
        if (t < tMin) t = (-hb + sq) / a;
        // The packed loop's filter may let an ellipsoid through whose discriminant is negative.
        const bool root = on && disc >= 0.0f;
        if (COUNT && root) cERoots++;
        if (root && !(t < tMin)) {
          const unsigned long long key = ((unsigned long long)__float_as_uint(t) << 32) |
                                         (unsigned long long)(~((unsigned int)(Ns + j) + 1u));
          atomicMin(&w_slot[owner], key);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      };
      const unsigned long long qMask = ballot64(qLane);
      if (Ne > 0 && qMask != 0ull) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        // Two ellipsoids per trip, one packed instruction for both.  The loop only selects candidates for
        // drainE, which evaluates hitEllipsoid proper; it may select too many, never too few.  It evaluates
        // hitEllipsoid's quadratic (raytrace.comp:173-187) through one matrix per ellipsoid,
        // M = diag(1/radii) R(conjugate q) (from the host, in double precision, rounded once):
        //   y = M (o - c), z = M d, D' = (y.z)^2 - (z.z)(y.y - 1)
        // and adds what the two evaluations can differ by in float32, |d|^2 (k1 |o - c|^2 + k2), with
        // k1 = (124 g + 98) u mu^4, k2 = (31 g + 29) u mu^2, u = 2^-24, mu >= |M|_2 and g the size of the
        // shader's intermediates against mu (see pt_update_scene): 38 packed instructions instead of 57.
        const P3 RO = splat3(rO), RD = splat3(rD);
        const f2 DD = mk2(dLen2q, dLen2q);
        for (int p = 0; 2 * p < Ne; ++p) {
          // Stage 1, the bounding sphere: V = (x.d)^2 - |d|^2 (|x|^2 c1 - c2) >= 0 is necessary for a root
          // (c1, c2 from the host, see pt_update_scene); 12 packed instructions.
          const float4 g0 = PT_ELLP(p, 0), g1 = PT_ELLP(p, 1), g2 = PT_ELLP(p, 2);
          const P3 x = sub3p(RO, mkp3(mk2(g0.x, g0.y), mk2(g0.z, g0.w), mk2(g1.x, g1.y)));
          const f2 xd = dot3p(x, RD);
          const f2 xx = dot3p(x, x);
          const f2 inner = fma2(xx, mk2(g1.z, g1.w), mk2(g2.x, g2.y));
          const f2 V = fma2(xd, xd, -(DD * inner));
          const unsigned long long nearMask = ballot64(__builtin_fmaxf(V.x, V.y) >= 0.0f) & qMask;
          if (COUNT && qLane) cETests += 2 * p + 1 < Ne ? 2u : 1u; // an odd count's last pair holds one ellipsoid twice
          if (nearMask == 0ull) continue;
          // Stage 2, the quadratic through the matrix, for the pairs some ray of the wave comes near.
          const float4 g3 = PT_ELLP(p, 3), g4 = PT_ELLP(p, 4), g5 = PT_ELLP(p, 5), g6 = PT_ELLP(p, 6), g7 = PT_ELLP(p, 7);
          const P3 m0 = mkp3(mk2(g2.z, g2.w), mk2(g3.x, g3.y), mk2(g3.z, g3.w));
          const P3 m1 = mkp3(mk2(g4.x, g4.y), mk2(g4.z, g4.w), mk2(g5.x, g5.y));
          const P3 m2 = mkp3(mk2(g5.z, g5.w), mk2(g6.x, g6.y), mk2(g6.z, g6.w));
          const P3 y = mkp3(dot3p(m0, x), dot3p(m1, x), dot3p(m2, x));
          const P3 z = mkp3(dot3p(m0, RD), dot3p(m1, RD), dot3p(m2, RD));
          const f2 a = dot3p(z, z);
          const f2 hb = dot3p(y, z);
          const f2 c = fma2(y.z, y.z, fma2(y.y, y.y, fma2(y.x, y.x, mk2(-1.0f, -1.0f))));
          const f2 disc = fma2(hb, hb, -(a * c));
          f2 val = fma2(mk2(g7.x, g7.y) * DD, xx, fma2(mk2(g7.z, g7.w), DD, disc));
          // both tests are necessary conditions: a candidate passes both
          if (!(V.x >= 0.0f)) val.x = -1.0f;
          if (!(V.y >= 0.0f)) val.y = -1.0f;
          if ((ballot64(__builtin_fmaxf(val.x, val.y) >= -1.0e-30f) & qMask) != 0ull) {
#pragma unroll
            for (int u = 0; u < 2; ++u) {
              const int j = 2 * p + u;
              const bool cand = qLane && (u == 0 ? val.x : val.y) >= -1.0e-30f;
              const unsigned long long cm = (j < Ne) ? (ballot64((u == 0 ? val.x : val.y) >= -1.0e-30f) & qMask) : 0ull; // j < Ne is wave-uniform
              if (cm != 0ull) {
                const int pre = lanes_below(cm);
                if (cand) w_queue[(qTail + pre) & 127] = (unsigned int)lane | ((unsigned int)(Ns + j) << 6);
                qTail += __popcll(cm);
                if (qTail - qHead >= 64) { drainE(64); qHead += 64; }
              }
            }
          }
        }
        if (qTail > qHead) { drainE(qTail - qHead); qHead = qTail; }
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      if (qLane || fastLane) { // the slot holds the spheres' winner, and the ellipsoids' if they were deferred
        const unsigned long long kbest = w_slot[lane];
        const unsigned int widx = ~(unsigned int)kbest - 1u; // none -> 0xFFFFFFFF
        if (widx < (unsigned int)(Ns + Ne)) {
          bestT = __uint_as_float((unsigned int)(kbest >> 32));
          bestIdx = (int)widx;
        }
      }
    }

    // Rays that are not finite keep the shader's sequential loop, NaN rules and all.
    const bool seqEll = live && !qLane;

    if (ballot64(seqEll) != 0ull && seqEll) {
      const float ox = rO.x, oy = rO.y, oz = rO.z, dx = rD.x, dy = rD.y, dz = rD.z;
      const float tMin = kEpsilon;
      for (int j = 0; j < Ne; ++j) {
        const float4 e0 = PT_ELL(j, 0), e1 = PT_ELL(j, 1), e2 = PT_ELL(j, 2), e3 = PT_ELL(j, 3);
        const V3 nq = xyz(e2), nq2 = xyz(e3);
        const V3 oc = rotate_q(mk3(ox - e0.x, oy - e0.y, oz - e0.z), nq, nq2, e0.w);
        const V3 dir = rotate_q(mk3(dx, dy, dz), nq, nq2, e0.w);
        const V3 invR = xyz(e1);
        const V3 ocS = mul3(oc, invR), dS = mul3(dir, invR);
        const float a = dot3(dS, dS);
        const float hb = dot3(ocS, dS);
        const float c = dot3(ocS, ocS) - 1.0f;
        const float disc = __builtin_fmaf(hb, hb, -(a * c));
        if (COUNT) cETests++;
        if (ballot64(!(disc < 0.0f)) != 0ull) {
          if (!(disc < 0.0f)) {
            if (COUNT) cERoots++;
            const float sq = __builtin_sqrtf(disc);
            float t = (-hb - sq) / a;
            bool ok = !(t < tMin || t > bestT);
            if (!ok) {
              t = (-hb + sq) / a;
              ok = !(t < tMin || t > bestT);
            }
            if (ok) { bestT = t; bestIdx = Ns + j; }
          }
        }
      }
    }

    PT_PHASE(cycEll)
    // ------------------------------------------------------------------
    // 3. each live lane consumes its result and runs until it has emitted
    //    its next ray or finished its pixel
    // ------------------------------------------------------------------
    // Labels of the per-lane coroutine.  `step` says where to resume.
    enum { GO_PATH_RESULT, GO_NEXT_LIGHT, GO_NEXT_SHADER_LIGHT, GO_NEXT_EM_SPHERE, GO_NEXT_EM_ELLIPSOID,
           GO_AFTER_LIGHTING, GO_FINISH, GO_EMITTED };
    const bool hitAny = bestIdx >= 0;

    // SurfaceShaderData of the hit being shaded (raytrace.comp:298-309), and its material's shader index.
    auto shaderDataAtHit = [&](int &shaderIndex) -> SurfaceShaderData {
      const float4 m0 = PT_MAT(hitMat, 0), m1 = PT_MAT(hitMat, 1), m2 = PT_MAT(hitMat, 2);
      SurfaceShaderData data;
      data.hitPoint = hitP; data.normal = hitN; data.viewDir = viewDir; data.surfaceColor = xyz(m0);
      data.ambient = m0.w; data.diffuse = m1.x; data.specular = m1.y; data.shininess = m1.z;
      data.emissiveStrength = m2.y; data.time = L.time; data.materialIndex = hitMat;
      shaderIndex = (int)m2.z;
      return data;
    };

    // What calculateLighting does with the outcome of one shadow / occlusion ray: `kind` is the
    // lane state the ray was cast in, kk its light index, (dir, range) the ray, pend the
    // contribution an emissive object makes if it is visible.
    auto applyResult = [&](const int kind, const int kk, const bool occluded, const V3 dir, const float range, const V3 pend) {
      if (occluded) return;
      if (kind == ST_LIGHT) {
        // raytrace.comp:348-363
        const float4 Lt = PT_LIGHT(kk);
        const float4 m0 = PT_MAT(hitMat, 0), m1 = PT_MAT(hitMat, 1);
        const V3 surfaceColor = xyz(m0);
        // lightDir and distance are the shadow ray's direction and range: the same
        // expressions on the same operands (getShadowFactor :245-247 vs :349-351)
        const float distance = range;
        const V3 lightDir = dir;
        const float attenuation = Lt.w / ((1.0f + 0.09f * distance) + (0.032f * distance) * distance);
        const float diffuseStrength = glmax(0.0f, dot3(hitN, lightDir));
        const V3 diffuse = scale3(scale3(surfaceColor, m1.x * diffuseStrength), attenuation);
        const V3 halfDir = normalize3(add3(lightDir, viewDir));
        const float specularStrength = highlight_pow(glmax(0.0f, dot3(hitN, halfDir)), m1.z);
        const V3 specular = scale3(scale3(scale3(surfaceColor, m1.y), specularStrength), attenuation);
        lit = add3(lit, add3(diffuse, specular));
      } else if (kind == ST_SHADER_LIGHT) {
        // the body of the shader's light loop for a light that is not shadowed (toon.glsl:22-39)
        const float4 Lt = PT_LIGHT(kk);
        int shaderIndex;
        const SurfaceShaderData data = shaderDataAtHit(shaderIndex);
        SurfaceShaderResult res;
        res.color = lit; res.emissive = mk3(0, 0, 0); res.emissiveStrength = data.emissiveStrength;
        // lightDir and distance are the shadow ray's direction and range (toon.glsl:19 vs :22-25)
        res = shader_light(shaderIndex, data, res, dir, range, Lt.w);
        lit = res.color;
      } else { // ST_EM_SPHERE, ST_EM_ELLIPSOID: raytrace.comp:433-440, 471-477
        lit = add3(lit, pend);
      }
    };

    // The rays of calculateLighting that follow the one just cast: the next light
    // (getShadowFactor :243-253), then the emissive spheres and ellipsoids that face the
    // hit point (:367-480, over the precomputed emissive lists).  Leaves `step` at
    // GO_EMITTED with the ray in (rO, rD, rTmax) or at the step after the lighting.
    auto nextLightingRay = [&](int &step) {
      if (step == GO_NEXT_LIGHT || step == GO_NEXT_SHADER_LIGHT) do {
        if (k < L.numLights) { // getShadowFactor (:243-253)
          const float4 L = PT_LIGHT(k);
          const V3 lightVec = sub3(xyz(L), hitP);
          const float distToLight = length3(lightVec);
          rO = hitP;
          rD = div3s(lightVec, distToLight);
          rTmax = distToLight;
          state = (step == GO_NEXT_LIGHT) ? ST_LIGHT : ST_SHADER_LIGHT;
          if (COUNT) cLight++;
          step = GO_EMITTED;
        } else if (step == GO_NEXT_LIGHT) {
          k = 0;
          step = GO_NEXT_EM_SPHERE;
        } else { // after the shader's light loop (toon.glsl:42-47), then result.color + result.emissive (raytrace.comp:324)
          int shaderIndex;
          const SurfaceShaderData data = shaderDataAtHit(shaderIndex);
          SurfaceShaderResult res;
          res.color = lit; res.emissive = mk3(0, 0, 0); res.emissiveStrength = data.emissiveStrength;
          res = shader_end(shaderIndex, data, res);
          lit = add3(res.color, res.emissive);
          step = GO_AFTER_LIGHTING;
        }
      } while (0);
      if (step == GO_NEXT_EM_SPHERE) do { // raytrace.comp:367-443, over the precomputed emissive list
        step = GO_NEXT_EM_ELLIPSOID;
        while (k < L.numEmSpheres) {
          // The lanes of a wave stand at different entries of the list, and the loop body below costs the wave the
          // same for one lane as for sixty-four, once per entry the slowest lane has to pass over.  So each lane first
          // runs over the entries that surely do not face its hit point (a few instructions each), and the body is
          // then taken by all lanes together, each at the first entry it could not rule out.
          // (Not in the variant for scenes with hundreds of spheres: a pass over those dwarfs this loop, and the registers
          // the run-ahead holds there are spilled ones.)
          if (!POOLED) while (k < L.numEmSpheres && turned_away(xyz(PT_EMS(k, 0)), hitP, hitN) && __float_as_int(PT_EMS(k, 2).x) == 0) k++;
          if (!(k < L.numEmSpheres)) break;
          const float4 s = PT_EMS(k, 0), sm = PT_EMS(k, 1), sx = PT_EMS(k, 2);
          const int si = __float_as_int(sx.y);
          if (si >= Ns) { k = L.numEmSpheres; break; }
          float sphereEmissive = sm.w;
          V3 emissiveColor = xyz(sm);
          const V3 toSphere = sub3(xyz(s), hitP);
          const float distToSphere = length3(toSphere);
          const V3 dirToSphere = div3s(toSphere, distToSphere);
          const int sphereShaderIndex = __float_as_int(sx.x);
          bool skip = false;
          if (sphereShaderIndex != 0) { // :377-414: the sphere's shader gives the emissive colour and strength
            const int sphereMat = PT_MATIDX(si);
            const float4 q0 = PT_MAT(sphereMat, 0), q1 = PT_MAT(sphereMat, 1);
            SurfaceShaderData data;
            data.hitPoint = sub3(xyz(s), scale3(dirToSphere, s.w)); // approximateHitPoint
            data.normal = neg3(dirToSphere);
            data.viewDir = dirToSphere;
            data.surfaceColor = xyz(q0);
            data.ambient = q0.w; data.diffuse = q1.x; data.specular = q1.y; data.shininess = q1.z;
            data.emissiveStrength = sphereEmissive; data.time = L.time; data.materialIndex = sphereMat;
            SurfaceShaderResult res;
            res.color = mk3(0, 0, 0); res.emissive = mk3(0, 0, 0); res.emissiveStrength = sphereEmissive;
            res = shader_begin(sphereShaderIndex, data, res);
            // A light loop only feeds result.color, which this caller drops: its shadow rays are not
            // cast (but counted as issued, since the shader function would cast them).
            if (COUNT && shader_has_light_loop(sphereShaderIndex)) { cLight += (unsigned int)L.numLights; cRays += (unsigned int)L.numLights; }
            res = shader_end(sphereShaderIndex, data, res);
            emissiveColor = res.emissive;
            sphereEmissive = res.emissiveStrength;
            if (length3(emissiveColor) < 0.001f) skip = true;
          }
          if (!skip) {
            const float alignment = dot3(hitN, dirToSphere);
            if (alignment > 0.0f) {
              rO = madd3(hitP, kEpsilon, hitN);
              rD = dirToSphere;
                  rTmax = (distToSphere - s.w) - kEpsilon;
              const float attenuation =
                  sphereEmissive / ((1.0f + 0.5f * distToSphere) + (0.1f * distToSphere) * distToSphere);
              pending = scale3(scale3(emissiveColor, alignment), attenuation);
              state = ST_EM_SPHERE;
              if (COUNT) cEmis++;
              step = GO_EMITTED;
              break;
            }
          }
          k++;
        }
        if (step != GO_EMITTED) k = 0;
      } while (0);
      if (step == GO_NEXT_EM_ELLIPSOID) do { // raytrace.comp:445-480
        step = GO_AFTER_LIGHTING;
        while (k < L.numEmEllipsoids) {
          if (!POOLED) while (k < L.numEmEllipsoids && turned_away(xyz(PT_EME(k, 0)), hitP, hitN)) k++; // as for the spheres
          if (!(k < L.numEmEllipsoids)) break;
          const float4 e0 = PT_EME(k, 0), em = PT_EME(k, 1), ex = PT_EME(k, 2);
          const int ei = __float_as_int(ex.x);
          if (ei >= Ne) { k = L.numEmEllipsoids; break; }
          const V3 toE = sub3(xyz(e0), hitP);
          const float distE = length3(toE);
          const V3 dirE = div3s(toE, distE);
          const float alignment = dot3(hitN, dirE);
          if (alignment > 0.0f) {
            rO = madd3(hitP, kEpsilon, hitN);
            rD = dirE;
              rTmax = (distE - e0.w) - kEpsilon;
            const float attenuation = em.w / ((1.0f + 0.5f * distE) + (0.1f * distE) * distE);
            pending = scale3(scale3(xyz(em), alignment), attenuation);
            state = ST_EM_ELLIPSOID;
            if (COUNT) cEmis++;
            step = GO_EMITTED;
            break;
          }
          k++;
        }
      } while (0);
    };

    if (fanOutOn) {
      // helpers report, then go back to being idle
      if (state == ST_JOB) {
        f_rec[9 * 64 + jobPos] = hitAny ? 1.0f : 0.0f;
        state = ST_IDLE;
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }

    if (state != ST_IDLE) {
      int step;
      // results of the rays this lane handed out last pass, in the order it would have cast them
      for (int r = 0; r < nRec; ++r) {
        const int pos = f_roundBase[r] + lanes_below(f_roundMask[r]);
        const V3 dir = mk3(f_rec[2 * 64 + pos], f_rec[3 * 64 + pos], f_rec[4 * 64 + pos]);
        const V3 pend = mk3(f_rec[6 * 64 + pos], f_rec[7 * 64 + pos], f_rec[8 * 64 + pos]);
        applyResult(__float_as_int(f_rec[pos]), __float_as_int(f_rec[64 + pos]), f_rec[9 * 64 + pos] != 0.0f, dir,
                    f_rec[5 * 64 + pos], pend);
      }
      nRec = 0;
      if (owedLit) { // raytrace.comp:651 / :663, with the throughput and the factor of the hit the sum belongs to
        fin = add3(fin, scale3(scale3(lit, owedThr), owedFactor));
        owedLit = false;
      }
      if (state == ST_PATH) {
        step = GO_PATH_RESULT;
      } else if (state == ST_WAIT) {
        step = GO_AFTER_LIGHTING; // every lighting ray was with another lane
      } else {
        applyResult(state, k, hitAny, rD, rTmax, pending);
        k++;
        step = state == ST_LIGHT ? GO_NEXT_LIGHT
               : state == ST_SHADER_LIGHT ? GO_NEXT_SHADER_LIGHT
               : state == ST_EM_SPHERE ? GO_NEXT_EM_SPHERE : GO_NEXT_EM_ELLIPSOID;
      }

      // The steps only ever lead forward (path result -> lights -> emissive spheres ->
      // emissive ellipsoids -> after lighting -> finish), so one sweep over them in that
      // order takes every lane to its next ray or to the end of its pixel, and lanes
      // that reach a step by different routes share one execution of it.
      {
        if (step == GO_PATH_RESULT) do {
          // traceIterative body after traceRay (raytrace.comp:612-680)
          // --- rayMarchVolume(currentRay, hit ? rec.t : MAX_DISTANCE, 0) (:534-585) ---
          float volX = -1.0f, volY = -1.0f, volZ = -1.0f;
          float volumeAlpha = 0.0f;
          if (VOLUME && L.numVolumes > 0) {
            unsigned int marchSteps;
            const V3 vol = march_volume<false, 0>(P, L, s_div255, [&]() -> float4 { return PT_MAT(L.volMaterial, 4); }, rO, curD,
                                        hitAny ? bestT : kMaxDistance, marchSteps);
            volX = vol.x; volY = vol.y; volZ = vol.z;
            if (COUNT) cVoxel += marchSteps;
            if (volX >= 0.0f) { // :618-620
              volumeAlpha = glclamp(length3(mk3(volX, volY, volZ)) / length3(mk3(1.0f, 1.0f, 1.0f)), 0.0f, 1.0f);
            }
          }
          if (volumeAlpha > 0.0f) { // :624-627
            fin = add3(fin, scale3(mk3(thr * volX, thr * volY, thr * volZ), volumeAlpha));
            thr = thr * (1.0f - volumeAlpha);
          }
          if (!hitAny) { // :672-679
            const V3 unitDir = normalize3(curD);
            const float t = 0.5f * (unitDir.y + 1.0f);
            const float omt = 1.0f - t;
            const V3 bg = mk3(L.bgBottom[0] * omt + L.bgTop[0] * t, L.bgBottom[1] * omt + L.bgTop[1] * t,
                              L.bgBottom[2] * omt + L.bgTop[2] * t);
            fin = add3(fin, scale3(bg, thr));
            step = GO_FINISH;
            break;
          }
          // hit record of the winner
          hitP = madd3(rO, bestT, curD);
          if (bestIdx < Ns) { // hitSphere :161-165
            const float4 s = P.spheres[bestIdx];
            hitN = scale3(sub3(hitP, xyz(s)), 1.0f / s.w);
            hitMat = PT_MATIDX(bestIdx);
          } else { // hitEllipsoid :198-207
            const int j = bestIdx - Ns;
            const float4 e0 = PT_ELL(j, 0), e1 = PT_ELL(j, 1), e2 = PT_ELL(j, 2), e3 = PT_ELL(j, 3);
            const V3 nq = xyz(e2), nq2 = xyz(e3);
            const V3 oc = rotate_q(sub3(rO, xyz(e0)), nq, nq2, e0.w);
            const V3 dir = rotate_q(curD, nq, nq2, e0.w);
            const V3 invR = xyz(e1);
            const V3 localPoint = madd3(oc, bestT, dir);
            const V3 normalLocal = mul3(mul3(localPoint, invR), invR);
            hitN = normalize3(rotate_q(normalLocal, neg3(nq), neg3(nq2), e0.w));
            hitMat = PT_MATIDX(P.matIdxEllBase + j);
          }
          const float4 m0 = PT_MAT(hitMat, 0);
          const float4 m2 = PT_MAT(hitMat, 2);
          const V3 surfaceColor = xyz(m0);
          const float emissiveStrength = m2.y;
          if (emissiveStrength > 0.0f) // :637-639
            fin = add3(fin, scale3(scale3(surfaceColor, thr), emissiveStrength));
          viewDir = normalize3(sub3(rO, hitP)); // :642
          // calculateLighting (:286-336)
          if (COUNT) cShaded++;
          const int shaderFunctionIndex = (int)m2.z;
          if (shaderFunctionIndex != 0) { // :297-326, dispatch through the shader registry
            int shaderIndex;
            const SurfaceShaderData data = shaderDataAtHit(shaderIndex);
            SurfaceShaderResult res;
            res.color = mk3(0, 0, 0); res.emissive = mk3(0, 0, 0); res.emissiveStrength = emissiveStrength;
            res = shader_begin(shaderFunctionIndex, data, res);
            if (shader_has_light_loop(shaderFunctionIndex)) { // resumed once per light, then shader_end
              lit = res.color;
              k = 0;
              step = GO_NEXT_SHADER_LIGHT;
            } else {
              res = shader_end(shaderFunctionIndex, data, res);
              lit = add3(res.color, res.emissive);
              step = GO_AFTER_LIGHTING;
            }
          } else {
            lit = scale3(surfaceColor, m0.w);
            if (emissiveStrength > 0.0f) lit = add3(lit, scale3(surfaceColor, emissiveStrength));
            k = 0;
            step = GO_NEXT_LIGHT;
          }
        } while (0);
        nextLightingRay(step);
        if (step == GO_AFTER_LIGHTING) do { // raytrace.comp:645-671
          const float4 m1 = PT_MAT(hitMat, 1);
          const float4 m2 = PT_MAT(hitMat, 2);
          const float transparency = m2.x;
          const float reflectivity = m1.w;
          bool cont = false;
          V3 nextO = hitP;
          if (reflectivity > 0.01f) {
            const float surfaceOpacity = (1.0f - reflectivity) * (1.0f - transparency);
            fin = add3(fin, scale3(scale3(lit, thr), surfaceOpacity));
            thr = thr * reflectivity;
            const float kk = 2.0f * dot3(hitN, curD); // reflect(I,N) = I - 2 dot(N,I) N
            const V3 reflectDir = sub3(curD, scale3(hitN, kk));
            nextO = madd3(hitP, kEpsilon, hitN);
            curD = reflectDir;
            cont = true;
          } else if (transparency > 0.01f) {
            const float opacity = 1.0f - transparency;
            fin = add3(fin, scale3(scale3(lit, thr), opacity));
            thr = thr * transparency;
            nextO = madd3(hitP, kEpsilon, curD);
            cont = true;
          } else {
            fin = add3(fin, scale3(lit, thr));
          }
          step = GO_FINISH;
          if (cont) {
            bounce++;
            // loop head of traceIterative (:605-606)
            if (bounce < L.maxBounces && !(length3(mk3(thr, thr, thr)) < 0.001f)) {
              rO = nextO; rD = curD; rTmax = kMaxDistance;
              state = ST_PATH;
              if (COUNT) cPath++;
              step = GO_EMITTED;
            }
          }
        } while (0);
        if (step == GO_FINISH) do { // main() :703-704
          write_pixel<SAMPLES>(P, L.accumCount, outIdx, fin);
          state = ST_IDLE;
          step = GO_EMITTED;
        } while (0);
      }
    }
    // Fan-out.  Once the frame has run out of pixels, idle lanes are all that is left of most
    // waves, while a lane inside calculateLighting still has up to numLights + emissive-object
    // rays to cast one after the other.  Which rays those are does not depend on how the earlier
    // ones end, so the lane runs ahead: per round it files its current ray as a record and
    // moves on to its next one, as long as idle lanes remain to take the records; it traces the
    // last ray itself.  Next pass the helpers report hit or miss and the lane applies the
    // results in casting order, which keeps every sum in the order the shader forms it.
    // (Lanes inside a surface shader's light loop stay sequential: the shader's end() adds to the sum before the results are in.)
    if (fanOutOn && exhausted) {
      const unsigned long long idleNow = ballot64(state == ST_IDLE);
      bool gen = state == ST_LIGHT || state == ST_EM_SPHERE || state == ST_EM_ELLIPSOID;
      const int helpers = (int)__popcll(idleNow);
      if (helpers > 0 && ballot64(gen) != 0ull) {
        int total = 0;
        for (int round = 0; round < kFanRounds; ++round) {
          const unsigned long long gm = ballot64(gen);
          const int c = (int)__popcll(gm);
          if (c == 0 || total + c > helpers) break;
          if (gen) {
            const int pos = total + lanes_below(gm);
            f_rec[pos] = __int_as_float(state);
            f_rec[64 + pos] = __int_as_float(k);
            f_rec[2 * 64 + pos] = rD.x; f_rec[3 * 64 + pos] = rD.y; f_rec[4 * 64 + pos] = rD.z;
            f_rec[5 * 64 + pos] = rTmax;
            f_rec[6 * 64 + pos] = pending.x; f_rec[7 * 64 + pos] = pending.y; f_rec[8 * 64 + pos] = pending.z;
            f_rec[10 * 64 + pos] = rO.x; f_rec[11 * 64 + pos] = rO.y; f_rec[12 * 64 + pos] = rO.z;
            nRec++;
          }
          if (lane == 0) { f_roundMask[round] = gm; f_roundBase[round] = total; }
          total += c;
          if (gen) {
            int step = state == ST_LIGHT ? GO_NEXT_LIGHT : (state == ST_EM_SPHERE ? GO_NEXT_EM_SPHERE : GO_NEXT_EM_ELLIPSOID);
            k++;
            nextLightingRay(step);
            if (step != GO_EMITTED) { // no further ray: wait for the helpers, then go on after the lighting
              gen = false;
              state = ST_WAIT;
            }
          }
        }
        // A lane that has handed out every lighting ray of its hit would sit out the next pass.  Where the path goes on
        // does not depend on the lighting (raytrace.comp:645-671: reflectivity, transparency, the normal), so it casts
        // the next path ray now and settles finalColor next pass, first the lighting sum of this hit, then the new hit.
        if (state == ST_WAIT && !(P.tune & (1 << 30))) {
          const float4 m1 = PT_MAT(hitMat, 1);
          const float4 m2 = PT_MAT(hitMat, 2);
          const float transparency = m2.x;
          const float reflectivity = m1.w;
          bool cont = false;
          float factor = 0.0f, nextThr = thr;
          V3 nextO = hitP, nextD = curD;
          if (reflectivity > 0.01f) {
            factor = (1.0f - reflectivity) * (1.0f - transparency);
            nextThr = thr * reflectivity;
            const float kk = 2.0f * dot3(hitN, curD);
            nextD = sub3(curD, scale3(hitN, kk));
            nextO = madd3(hitP, kEpsilon, hitN);
            cont = true;
          } else if (transparency > 0.01f) {
            factor = 1.0f - transparency;
            nextThr = thr * transparency;
            nextO = madd3(hitP, kEpsilon, curD);
            cont = true;
          }
          // only if the loop head of traceIterative (:605-606) lets the next bounce run
          if (cont && bounce + 1 < L.maxBounces && !(length3(mk3(nextThr, nextThr, nextThr)) < 0.001f)) {
            owedLit = true; owedThr = thr; owedFactor = factor;
            thr = nextThr;
            curD = nextD;
            bounce++;
            rO = nextO; rD = curD; rTmax = kMaxDistance;
            state = ST_PATH;
            if (COUNT) cPath++;
          }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (state == ST_IDLE) {
          const int q = lanes_below(idleNow);
          if (q < total) {
            rO = mk3(f_rec[10 * 64 + q], f_rec[11 * 64 + q], f_rec[12 * 64 + q]);
            rD = mk3(f_rec[2 * 64 + q], f_rec[3 * 64 + q], f_rec[4 * 64 + q]);
            rTmax = f_rec[5 * 64 + q];
            jobPos = q;
            state = ST_JOB;
          }
        }
      }
    }
    PT_PHASE(cycShade)
  }
#undef PT_PHASE

  if (PHASES && lane == 0 && P.waveLog) {
    // Timeline run: one record per wave and no atomics (thousands of same-address atomics at the end of the
    // kernel would stretch the very tail the timeline is there to show).
    unsigned long long *rec = P.waveLog + 4ull * ((unsigned long long)blockIdx.x * (blockDim.x >> 6) + (unsigned long long)waveInBlock);
    rec[0] = rtStart;
    rec[1] = exhausted ? rtDry : 0ull;
    rec[2] = __builtin_amdgcn_s_memrealtime();
    rec[3] = (unsigned long long)nPasses | ((unsigned long long)passesAtDry << 32);
  } else if (PHASES && lane == 0) {
    if (coop) atomicAdd(&P.counters[18], __builtin_amdgcn_s_memtime() - tPooled);
    atomicAdd(&P.counters[19], nPooled);
    atomicAdd(&P.counters[20], nPooledRays);
    atomicAdd(&P.counters[13], cycRefill);
    atomicAdd(&P.counters[14], cycSpheres);
    atomicAdd(&P.counters[15], cycEll);
    atomicAdd(&P.counters[16], cycShade);
    atomicAdd(&P.counters[17], cycExact);
    const unsigned long long tEnd = __builtin_amdgcn_s_memtime();
    atomicAdd(&P.counters[21], cycTicket);
    atomicAdd(&P.counters[22], nTickets);
    if (exhausted) atomicAdd(&P.counters[23], tEnd - tExhausted);
    atomicAdd(&P.counters[24], tEnd - tStart);
  }
  if (COUNT) {
    // wave reduction, then one atomic per counter per wave
    unsigned long long v[13];
    v[0] = cRays; v[1] = cPath; v[2] = cLight; v[3] = cEmis;
    v[4] = cTests;  // counted where the tests run: the dense loop's trips, the exact routine's chunks
    v[5] = cETests; // the candidate loop's pairs, the sequential loop
    v[6] = cVoxel; v[7] = cShaded; v[8] = cRoots; v[9] = cERoots;
    // wave-level counts: every live lane of a wave saw the same trips, so take lane-max, not the sum
    v[10] = 0; v[11] = 0; v[12] = cExact;
#pragma unroll
    for (int c = 0; c < 13; ++c) {
      if (c == 10 || c == 11) continue;
      unsigned long long x = v[c];
      for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off, 64);
      if (lane == 0) atomicAdd(&P.counters[c], x);
    }
    {
      unsigned int g = cGroups, sl = cSlow;
      for (int off = 32; off > 0; off >>= 1) {
        unsigned int og = __shfl_down(g, off, 64), os = __shfl_down(sl, off, 64);
        g = g > og ? g : og;
        sl = sl > os ? sl : os;
      }
      if (lane == 0) { atomicAdd(&P.counters[10], (unsigned long long)g); atomicAdd(&P.counters[11], (unsigned long long)sl); }
    }
  }
#undef PT_CTRL_LOAD
#undef PT_CTRL_STORE
#undef PT_IMG
#undef PT_ELL
#undef PT_MAT
#undef PT_LIGHT
#undef PT_EMS
#undef PT_EME
#undef PT_MATIDX
#undef PT_PAIR_DISC
#undef PT_ENQUEUE
}

// De-interleave the compact band-major frames of `bandCount` renderers into
// one height x width frame.  Source frame of renderer p starts at
// p * strideRows rows; elements are uint32 (rgba8) or float4.
} // namespace ptdev
