// pt_volume.hpp — rayMarchVolume of the reference shader (reference shaders/raytrace.comp:485-585): the box test,
// the fixed-step march with nearest-neighbour samples (and the trilinear extension), the colour it returns.
// Shared by the general frame kernel (every path ray marches through the volume up to its hit) and the kernel for
// frames without solids.
#pragma once

#include "pt_frame.hpp"

namespace ptdev {

// The lanes of the wave for which p holds, as a mask in scalar registers straight from the compare.
__device__ __forceinline__ unsigned long long ballot64(bool p) { return __builtin_amdgcn_ballot_w64(p); }
// How many lanes below this one are set in m (v_mbcnt_lo/hi: two instructions, no per-lane mask to keep).
__device__ __forceinline__ int lanes_below(unsigned long long m) {
  return (int)__builtin_amdgcn_mbcnt_hi((unsigned int)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int)m, 0u));
}

// x / 255.0f for 0 <= x <= 255 (an interpolated voxel byte), correctly rounded like the division itself: the product
// with the rounded reciprocal, corrected once through the exact remainder.  Checked against the division for every
// float in [0, 255] (tests/test_div255_exact.py): three instructions instead of the eleven of a division.
__device__ __forceinline__ float div255(float x) {
  const float rc = 1.0f / 255.0f;
  const float q = x * rc;
  const float r = __builtin_fmaf(-255.0f, q, x);
  return __builtin_fmaf(r, rc, q);
}

// P: the kernel's argument (pointers are read from it), L: the copy of the frame constants the kernel reads scalars
// from (the same object, or its copy in LDS), s_div255: the table b / 255 for b = 0 .. 255 in LDS, materialRow4():
// row 4 of the volume's material (colour, absorption coefficient).  Returns rayMarchVolume's colour ((-1,-1,-1): nothing there)
// and the number of samples taken.
// WIDE: trips of eight samples, two per packed operation, where every lane is inside the box (for a kernel with
// registers to spare; the general kernel takes four, one at a time).
// KIND 1 / 2: the caller has checked that the volume is the common case -- bricked copy, every extent a power of two,
// v0 = (0,0,0) (the reference's volume: box (0,0,0)-(2,2,2)) -- and the filter is the nearest-neighbour one (1) or the
// trilinear one (2), so the code for the other cases, and the wave-uniform branches that choose between them per
// sample, are not compiled in (x - 0 is x).  KIND 0: everything is decided at run time.
template <bool WIDE, int KIND, class MaterialRow>
__device__ __forceinline__ V3 march_volume(const FrameParams &P, const FrameParams &L, const float *s_div255, MaterialRow &&materialRow4,
                                           const V3 rO, const V3 curD, const float maxDist, unsigned int &steps) {
  constexpr bool FAST = KIND == 1;  // nearest filter, common volume
  constexpr bool COMMON = KIND != 0; // common volume, filter known
  float volX = -1.0f, volY = -1.0f, volZ = -1.0f;
  steps = 0u;
    // intersectBox (:519-532)
    const V3 invDir = mk3(1.0f / (curD.x + 1e-8f), 1.0f / (curD.y + 1e-8f), 1.0f / (curD.z + 1e-8f));
    const V3 t0 = mul3(sub3(ld3(L.boxMin), rO), invDir);
    const V3 t1 = mul3(sub3(ld3(L.boxMax), rO), invDir);
    const V3 tmn = mk3(glmin(t0.x, t1.x), glmin(t0.y, t1.y), glmin(t0.z, t1.z));
    const V3 tmx = mk3(glmax(t0.x, t1.x), glmax(t0.y, t1.y), glmax(t0.z, t1.z));
    float tNear = glmax(glmax(tmn.x, tmn.y), tmn.z);
    float tFar = glmin(glmin(tmx.x, tmx.y), tmx.z);
    if (tFar >= tNear && tFar > 0.0f) {
      tNear = glmax(tNear, 0.0f);
      tFar = glmin(tFar, maxDist);
      const float4 m4 = materialRow4(); // fetched only by the lanes whose ray meets the box
      const float absorptionCoeff = m4.w;
      const float rxf = (float)L.res[0], ryf = (float)L.res[1], rzf = (float)L.res[2];
      const int rx = L.res[0], ry = L.res[1], rz = L.res[2];
      float t = tNear;
      float acc = 0.0f;
      int stepCount = 0;
      const V3 vpos = ld3(L.volPos), lo = COMMON ? mk3(0.0f, 0.0f, 0.0f) : ld3(L.v0), hi = ld3(L.v1), ext = ld3(L.extent);
      const float rcpEx = L.rcpExtent[0], rcpEy = L.rcpExtent[1], rcpEz = L.rcpExtent[2];
      const float cellTolX = 1.5f * 0x1p-22f * (float)L.res[0], cellTolY = 1.5f * 0x1p-22f * (float)L.res[1],
                  cellTolZ = 1.5f * 0x1p-22f * (float)L.res[2];
      const uint8_t *__restrict__ vox = P.voxels;
      const bool exactExtent = COMMON || L.exactExtent != 0;
      const bool bricked = COMMON || P.brickX > 0; // 8x8x8-voxel bricks, 512 bytes each, x fastest inside and between bricks
      const int nbx = P.brickX, nby = P.brickY;
      if (KIND == 2 || (KIND == 0 && L.volumeFilter == 1)) {
        // Trilinear extension: voxel centres at (i + 0.5) / res; lerp(a, b, t) =
        // fma(t, b - a, a) along x, then y, then z on the byte values, then / 255.
        const bool brk = COMMON || P.brickX > 0;
        const int bX = P.brickX, bY = P.brickY;
        auto fetch = [&](int ix, int iy, int iz) -> float {
          const int id = brk ? ((((iz >> 3) * bY + (iy >> 3)) * bX + (ix >> 3)) << 9) +
                                   (((iz & 7) << 6) | ((iy & 7) << 3) | (ix & 7))
                             : ix + iy * rx + iz * rx * ry;
          return (float)vox[id];
        };
        // WIDE: kTri samples per trip, their voxel loads in flight together; every sample computes its (clamped, hence
        // valid) cells and loads them, the box test and the loop's own condition select afterwards what counts.
        // Each cell index is a sum of one term per axis (for the bricked copy: i + (i >> 3) * 504, 8 i + (i >> 3) *
        // (512 nbx - 64), 64 i + (i >> 3) * (512 nbx nby - 512)), so the eight cells of a sample share six terms.
        constexpr int kTri = 2; // four per trip need 96 registers instead of 53 and are no faster
        while (WIDE && t < tFar && stepCount < 500) {
          // both fit 24 bits as unsigned factors (or the copy is linear); the multiply is the unsigned one, the signed
          // 24-bit multiply would read a factor of 2^23 or more as negative
          const int rowStep = (bX << 9) - 64, sliceStep = ((bX * bY) << 9) - 512;
          float tsT[kTri];
          tsT[0] = t;
#pragma unroll
          for (int s = 1; s < kTri; ++s) tsT[s] = tsT[s - 1] + 0.02f;
          // The two cells of a sample that differ in x only are neighbours in memory (x runs fastest in the linear copy
          // and inside a brick), so one 16-bit load fetches both: four loads per sample instead of eight.  Where
          // the upper cell starts the next brick (ix & 7 = 7, 1 lane in 8) its byte is fetched on its own; where it
          // is the clamped copy of the lower one (ix = res - 1) the weight f is 0 and the second byte does not matter.
          unsigned int base[kTri][4], xStep[kTri];
          float f[kTri][3];
          bool inside[kTri], lone[kTri];
#pragma unroll
          for (int s = 0; s < kTri; ++s) {
            const V3 rel = sub3(madd3(rO, tsT[s], curD), vpos);
            inside[s] = !(rel.x < lo.x || rel.y < lo.y || rel.z < lo.z || rel.x > hi.x || rel.y > hi.y || rel.z > hi.z);
            // where every extent is a power of two the product with its reciprocal is the quotient
            const float u[3] = {exactExtent ? (rel.x - lo.x) * rcpEx : (rel.x - lo.x) / ext.x,
                                exactExtent ? (rel.y - lo.y) * rcpEy : (rel.y - lo.y) / ext.y,
                                exactExtent ? (rel.z - lo.z) * rcpEz : (rel.z - lo.z) / ext.z};
            const float rf[3] = {rxf, ryf, rzf};
            const int ri[3] = {rx, ry, rz};
            int term[2][3]; // [lower / upper cell][axis]
#pragma unroll
            for (int a = 0; a < 3; ++a) {
              const float p = u[a] * rf[a] - 0.5f;
              const float top = (float)(ri[a] - 1);
              // p < 0 ? 0 : (p > top ? top : p), and 0 for a p that is not a number, in one instruction (the median of
              // (p, 0, top); a -0 it may return where the comparisons give +0 has the same cell and weight 0 either way)
              const float q = __builtin_amdgcn_fmed3f(p, 0.0f, top);
              const int i0 = (int)q;
              f[s][a] = __builtin_amdgcn_fractf(q); // q - floor(q), exact for q >= 0: the same as q - float(int(q))
              const int i1 = i0 + 1 > ri[a] - 1 ? ri[a] - 1 : i0 + 1;
              const int lin = a == 0 ? 1 : (a == 1 ? rx : rx * ry);
              const int shift = 3 * a, step = a == 0 ? 504 : (a == 1 ? rowStep : sliceStep);
              term[0][a] = brk ? (i0 << shift) + (int)__umul24((unsigned int)(i0 >> 3), (unsigned int)step) : i0 * lin;
              term[1][a] = brk ? (i1 << shift) + (int)__umul24((unsigned int)(i1 >> 3), (unsigned int)step) : i1 * lin;
              if (a == 0) lone[s] = brk && (i0 & 7) == 7 && i1 != i0;
            }
            xStep[s] = (unsigned int)(term[1][0] - term[0][0]);
#pragma unroll
            for (int c = 0; c < 4; ++c) base[s][c] = (unsigned int)(term[0][0] + term[c & 1][1] + term[c >> 1][2]);
          }
          unsigned int pair[kTri][4];
#pragma unroll
          for (int s = 0; s < kTri; ++s)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
              unsigned short v; // any byte address: the device reads unaligned words of global memory
              __builtin_memcpy(&v, vox + base[s][c], 2);
              pair[s][c] = v;
            }
          bool anyLone = false;
#pragma unroll
          for (int s = 0; s < kTri; ++s) anyLone = anyLone || lone[s];
          if (ballot64(anyLone) != 0ull) {
#pragma unroll
            for (int s = 0; s < kTri; ++s)
              if (lone[s]) {
#pragma unroll
                for (int c = 0; c < 4; ++c) pair[s][c] = (pair[s][c] & 0xFFu) | ((unsigned int)vox[base[s][c] + xStep[s]] << 8);
              }
          }
          const int stepsBefore = stepCount;
#pragma unroll
          for (int s = 0; s < kTri; ++s) {
            // t only grows, so a sample that is not taken ends the lane's march (the ones after it are not taken either)
            const bool taken = s == 0 || (tsT[s] < tFar && stepsBefore + s < 500);
            float b[8];
#pragma unroll
            for (int c = 0; c < 4; ++c) { b[2 * c] = (float)(pair[s][c] & 0xFFu); b[2 * c + 1] = (float)((pair[s][c] >> 8) & 0xFFu); }
            const float c00 = __builtin_fmaf(f[s][0], b[1] - b[0], b[0]), c10 = __builtin_fmaf(f[s][0], b[3] - b[2], b[2]);
            const float c01 = __builtin_fmaf(f[s][0], b[5] - b[4], b[4]), c11 = __builtin_fmaf(f[s][0], b[7] - b[6], b[6]);
            const float c0 = __builtin_fmaf(f[s][1], c10 - c00, c00), c1 = __builtin_fmaf(f[s][1], c11 - c01, c01);
            const float density = inside[s] ? div255(__builtin_fmaf(f[s][2], c1 - c0, c0)) : 0.0f;
            const float next = __builtin_fmaf(1.0f - acc, density * absorptionCoeff, acc);
            acc = taken ? next : acc;
            stepCount += taken ? 1 : 0;
          }
          t = tsT[kTri - 1] + 0.02f;
        }
        while (t < tFar && stepCount < 500) {
          const V3 pos = madd3(rO, t, curD);
          const V3 rel = sub3(pos, vpos);
          float density = 0.0f;
          if (!(rel.x < lo.x || rel.y < lo.y || rel.z < lo.z || rel.x > hi.x || rel.y > hi.y || rel.z > hi.z)) {
            const float u[3] = {(rel.x - lo.x) / ext.x, (rel.y - lo.y) / ext.y, (rel.z - lo.z) / ext.z};
            const float rf[3] = {rxf, ryf, rzf};
            const int ri[3] = {rx, ry, rz};
            int i0[3], i1[3];
            float f[3];
#pragma unroll
            for (int a = 0; a < 3; ++a) {
              const float p = u[a] * rf[a] - 0.5f;
              const float top = (float)(ri[a] - 1);
              float q = p < 0.0f ? 0.0f : (p > top ? top : p);
              if (q != q) q = 0.0f;
              i0[a] = (int)q;
              f[a] = q - (float)i0[a];
              i1[a] = i0[a] + 1 > ri[a] - 1 ? ri[a] - 1 : i0[a] + 1;
            }
            const float b000 = fetch(i0[0], i0[1], i0[2]), b100 = fetch(i1[0], i0[1], i0[2]);
            const float b010 = fetch(i0[0], i1[1], i0[2]), b110 = fetch(i1[0], i1[1], i0[2]);
            const float b001 = fetch(i0[0], i0[1], i1[2]), b101 = fetch(i1[0], i0[1], i1[2]);
            const float b011 = fetch(i0[0], i1[1], i1[2]), b111 = fetch(i1[0], i1[1], i1[2]);
            const float c00 = __builtin_fmaf(f[0], b100 - b000, b000), c10 = __builtin_fmaf(f[0], b110 - b010, b010);
            const float c01 = __builtin_fmaf(f[0], b101 - b001, b001), c11 = __builtin_fmaf(f[0], b111 - b011, b011);
            const float c0 = __builtin_fmaf(f[1], c10 - c00, c00), c1 = __builtin_fmaf(f[1], c11 - c01, c01);
            density = __builtin_fmaf(f[2], c1 - c0, c0) / 255.0f;
          }
          const float sampleAlpha = density * absorptionCoeff;
          acc = __builtin_fmaf(1.0f - acc, sampleAlpha, acc);
          t += 0.02f;
          stepCount++;
        }
      }
      // Index of the (clamped) cell (ix, iy, iz).  Bricked: ((bz * nby + by) * nbx + bx) * 512 + (iz & 7) * 64 +
      // (iy & 7) * 8 + (ix & 7) with b = i >> 3, written as the linear part ix + 8 iy + 64 iz plus what each
      // whole brick adds along its axis; the factors fit 24 bits as unsigned numbers (the host keeps the linear layout
      // otherwise), so the multiplies are unsigned ones: brick coordinates and steps are never negative.
      const int brickRowStep = (nbx << 9) - 64, bricksPerSliceLess1 = nbx * nby - 1;
      auto cellIndex = [&](int ix, int iy, int iz) -> int {
        if (!bricked) return ix + iy * rx + iz * rx * ry;
        int id = ix + (iy << 3) + (iz << 6);
        id += (int)__umul24((unsigned int)(ix >> 3), 504u);
        id += (int)__umul24((unsigned int)(iy >> 3), (unsigned int)brickRowStep);
        return id + (int)(__umul24((unsigned int)(iz >> 3), (unsigned int)bricksPerSliceLess1) << 9);
      };
      // the same as an unsigned offset (the load then takes the base address from scalar registers), with one
      // multiplication for the slices too
      const int brickSliceStep = ((nbx * nby) << 9) - 512;
      auto cellOffset = [&](int ix, int iy, int iz) -> unsigned int {
        if (!bricked) return (unsigned int)(ix + iy * rx + iz * rx * ry);
        // three 24-bit multiply-adds in a chain and two shift-adds, eight instructions with the three shifts (left to
        // itself the compiler multiplies, shifts and sums with three-input adds: ten)
        unsigned int id;
        asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(id) : "v"(ix >> 3), "s"(504), "v"(ix));
        asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(id) : "v"(iy >> 3), "s"(brickRowStep), "v"(id));
        asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(id) : "v"(iz >> 3), "s"(brickSliceStep), "v"(id));
        asm("v_lshl_add_u32 %0, %1, 3, %2" : "=v"(id) : "v"(iy), "v"(id));
        asm("v_lshl_add_u32 %0, %1, 6, %2" : "=v"(id) : "v"(iz), "v"(id));
        return id;
      };
      // clamp(int(p), 0, res - 1) as int(clamp(p, 0, res - 1)): the same integer for every p (the conversion
      // truncates towards zero and saturates; res - 1 is a float exactly)
      const float topX = rxf - 1.0f, topY = ryf - 1.0f, topZ = rzf - 1.0f;
      // Samples a lane takes at tSafeA <= t <= tSafeB are inside the box whatever the rounding of their
      // position, so they skip sampleVolume's six comparisons: the position is linear in t, both ends of that
      // stretch lie inside every slab by a margin of 1e-5 of the magnitudes that enter the position (its
      // rounding error is below 3e-7 of them), hence so does everything between them.  The stretch leaves out
      // the first and the last hundredth of the chord, where the entry and exit faces are closer than that.
      float tSafeA = kMaxDistance, tSafeB = -kMaxDistance;
      if (KIND != 2 && exactExtent && (FAST || L.volumeFilter != 1)) {
        const float ta = tNear + 0.01f, tb = tFar - 0.01f;
        const V3 pa = sub3(madd3(rO, ta, curD), vpos), pb = sub3(madd3(rO, tb, curD), vpos);
        const float mx = 1e-5f * (__builtin_fabsf(rO.x) + __builtin_fabsf(vpos.x) + __builtin_fabsf(lo.x) + __builtin_fabsf(hi.x) + tFar * __builtin_fabsf(curD.x));
        const float my = 1e-5f * (__builtin_fabsf(rO.y) + __builtin_fabsf(vpos.y) + __builtin_fabsf(lo.y) + __builtin_fabsf(hi.y) + tFar * __builtin_fabsf(curD.y));
        const float mz = 1e-5f * (__builtin_fabsf(rO.z) + __builtin_fabsf(vpos.z) + __builtin_fabsf(lo.z) + __builtin_fabsf(hi.z) + tFar * __builtin_fabsf(curD.z));
        const bool inside = glmin(pa.x, pb.x) >= lo.x + mx && glmax(pa.x, pb.x) <= hi.x - mx &&
                            glmin(pa.y, pb.y) >= lo.y + my && glmax(pa.y, pb.y) <= hi.y - my &&
                            glmin(pa.z, pb.z) >= lo.z + mz && glmax(pa.z, pb.z) <= hi.z - mz;
        if (ta <= tb && inside) { tSafeA = ta; tSafeB = tb; }
      }
      const float scaleX = rcpEx * rxf, scaleY = rcpEy * ryf, scaleZ = rcpEz * rzf; // exact where the extents are powers of two
      // A sample inside a lane's safe stretch lies inside every slab by 0.9e-5 of the magnitudes involved, so its scaled
      // position p satisfies 0 < p < res on every axis and int(p) is a cell as it stands: clamp(int(p), 0, res - 1)
      // of the shader changes nothing there.  One unsigned bound on the finished offset stands in for the three clamps
      // (it never binds; it keeps a load inside the volume whatever happens).
      const unsigned int lastByte = bricked ? (unsigned int)(nbx * nby * ((rz + 7) >> 3)) * 512u - 1u : (unsigned int)(rx * ry * rz) - 1u;
      auto safeOffset = [&](int ix, int iy, int iz) -> unsigned int {
        const unsigned int o = cellOffset(ix, iy, iz);
        return o < lastByte ? o : lastByte;
      };
      // two samples of a safe stretch, one packed operation for both (sampleVolume's arithmetic, :485-517)
      auto safePair = [&](const float ta, const float tb, unsigned int &ra, unsigned int &rb) {
        const f2 tt = mk2(ta, tb);
        f2 qx = fma2(tt, mk2(curD.x, curD.x), mk2(rO.x, rO.x)) - mk2(vpos.x, vpos.x);
        f2 qy = fma2(tt, mk2(curD.y, curD.y), mk2(rO.y, rO.y)) - mk2(vpos.y, vpos.y);
        f2 qz = fma2(tt, mk2(curD.z, curD.z), mk2(rO.z, rO.z)) - mk2(vpos.z, vpos.z);
        if (!FAST) { qx = qx - mk2(lo.x, lo.x); qy = qy - mk2(lo.y, lo.y); qz = qz - mk2(lo.z, lo.z); }
        qx = qx * mk2(scaleX, scaleX); qy = qy * mk2(scaleY, scaleY); qz = qz * mk2(scaleZ, scaleZ);
        ra = (unsigned int)vox[safeOffset((int)qx.x, (int)qy.x, (int)qz.x)];
        rb = (unsigned int)vox[safeOffset((int)qx.y, (int)qy.y, (int)qz.y)];
      };
      // The march has no data-dependent exit, so four steps are taken per
      // trip: their voxel loads are in flight together and the accumulation
      // then runs in step order, exactly as one step per trip would.
      constexpr int kTrip = WIDE ? 8 : 4; // samples per trip inside the lane's safe stretch
      while (t < tFar && stepCount < 500) {
        // Trips of kTrip samples without tests, masks or per-sample validity, for as long as this lane has kTrip
        // samples inside its safe stretch (ts[kTrip - 1] <= tSafeB < tFar: all are taken).  The lanes of a wave
        // enter and leave their stretches at different steps; a lane that cannot take such a trip sits this loop
        // out (it costs the others nothing) and takes its few samples at the two ends of its chord in the tested
        // trips below, so a tile needs about (longest chord / kTrip) of these trips plus a handful of the others,
        // whatever the spread of its chords.  ((a * rcpE) * res = a * (rcpE * res), both scalings exact.)
        for (;;) {
          float ts[kTrip];
          ts[0] = t;
#pragma unroll
          for (int u = 1; u < kTrip; ++u) ts[u] = ts[u - 1] + 0.02f;
          if (!(ts[0] >= tSafeA && ts[kTrip - 1] <= tSafeB && stepCount <= 500 - kTrip)) break;
          unsigned int raw[kTrip];
          if (WIDE) {
#pragma unroll
            for (int u = 0; u < kTrip; u += 2) safePair(ts[u], ts[u + 1], raw[u], raw[u + 1]); // two samples per packed operation
          } else {
#pragma unroll
            for (int u = 0; u < kTrip; ++u) {
              const V3 rel = sub3(madd3(rO, ts[u], curD), vpos);
              raw[u] = (unsigned int)vox[safeOffset((int)((rel.x - lo.x) * scaleX), (int)((rel.y - lo.y) * scaleY), (int)((rel.z - lo.z) * scaleZ))];
            }
          }
#pragma unroll
          for (int u = 0; u < kTrip; ++u) acc = __builtin_fmaf(1.0f - acc, s_div255[raw[u]] * absorptionCoeff, acc);
          t = ts[kTrip - 1] + 0.02f;
          stepCount += kTrip;
        }
        // what is left of a lane's safe stretch when it is shorter than a trip: two samples at a time, still without tests
        if (WIDE) for (;;) {
          const float tb = t + 0.02f;
          if (!(t >= tSafeA && tb <= tSafeB && stepCount <= 498)) break;
          unsigned int ra, rb;
          safePair(t, tb, ra, rb);
          acc = __builtin_fmaf(1.0f - acc, s_div255[ra] * absorptionCoeff, acc);
          acc = __builtin_fmaf(1.0f - acc, s_div255[rb] * absorptionCoeff, acc);
          t = tb + 0.02f;
          stepCount += 2;
        }
        if (!(t < tFar && stepCount < 500)) break;
        float ts[4];
        ts[0] = t;
#pragma unroll
        for (int u = 1; u < 4; ++u) ts[u] = ts[u - 1] + 0.02f;
        int index[4];
        int n = 1;
        // A tested trip ends early at a sample from which the lane can take trips without tests (the first sample of a
        // chord is the only one before its safe stretch): such a sample and the ones after it are not taken here.
        // (0.145: seven steps of 0.02 and what their rounding can add; a lane this misses just stays with the tests.)
#pragma unroll
        for (int u = 1; u < 4; ++u)
          if (n == u && ts[u] < tFar && stepCount + u < 500 &&
              !(WIDE && ts[u] >= tSafeA && ts[u] + 0.145f <= tSafeB && stepCount + u <= 500 - kTrip))
            n = u + 1;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          // sampleVolume (:485-517), nearest neighbour
          const V3 pos = madd3(rO, ts[u], curD);
          const V3 rel = sub3(pos, vpos);
          index[u] = -1;
          if (u < n && !(rel.x < lo.x || rel.y < lo.y || rel.z < lo.z || rel.x > hi.x || rel.y > hi.y || rel.z > hi.z)) {
            // Only the integer part of ((rel - lo) / extent) * resolution is used.  With the reciprocal of the
            // extent (rounded once, from the host) the product differs from the shader's by less than
            // 1.26 * 2^-22 * resolution (one rounding of the reciprocal, one of each product, against one of
            // the quotient), so away from an integer by more than cellTol = 1.5 * 2^-22 * resolution it has the
            // same integer part; the few samples closer than that take the division.
            const float ax = rel.x - lo.x, ay = rel.y - lo.y, az = rel.z - lo.z;
            float px = (ax * rcpEx) * rxf, py = (ay * rcpEy) * ryf, pz = (az * rcpEz) * rzf;
            if (!exactExtent) { // where every extent is a power of two the product IS the shader's quotient
              const bool sure = __builtin_fabsf(__builtin_amdgcn_fractf(px) - 0.5f) < 0.5f - cellTolX &&
                                __builtin_fabsf(__builtin_amdgcn_fractf(py) - 0.5f) < 0.5f - cellTolY &&
                                __builtin_fabsf(__builtin_amdgcn_fractf(pz) - 0.5f) < 0.5f - cellTolZ;
              if (!sure) { // also where a product is not a number
                px = (ax / ext.x) * rxf; py = (ay / ext.y) * ryf; pz = (az / ext.z) * rzf;
              }
            }
            index[u] = cellIndex((int)__builtin_amdgcn_fmed3f(px, 0.0f, topX), (int)__builtin_amdgcn_fmed3f(py, 0.0f, topY),
                                 (int)__builtin_amdgcn_fmed3f(pz, 0.0f, topZ));
          }
        }
        unsigned int raw[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) raw[u] = index[u] >= 0 ? (unsigned int)vox[index[u]] : 0u;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          if (u < n) {
            const float density = s_div255[raw[u]];
            const float sampleAlpha = density * absorptionCoeff;
            acc = __builtin_fmaf(1.0f - acc, sampleAlpha, acc);
          }
        }
        t = ts[n - 1] + 0.02f;
        stepCount += n;
      }
      steps = (unsigned int)stepCount;
      // mix(vec3(-1), materialColor*acc, float(acc >= 1e-4)) = x*(1-a) + y*a
      const float a = (acc >= 0.0001f) ? 1.0f : 0.0f;
      const float oma = 1.0f - a;
      volX = -1.0f * oma + (m4.x * acc) * a;
      volY = -1.0f * oma + (m4.y * acc) * a;
      volZ = -1.0f * oma + (m4.z * acc) * a;
    }
  return mk3(volX, volY, volZ);
}

// main()'s last lines (raytrace.comp:703-704): the clamped colour goes to the rgba8 frame, and to the float frame
// if there is one; with accumulation on, what goes there is the mean of the clamped samples so far.
// PLANES (a launch that draws several samples): the clamped colour goes to this sample's plane, P.accum + the offset
// outIdx carries, and nowhere else.
template <bool PLANES = false>
__device__ __forceinline__ void write_pixel(const FrameParams &P, const int accumCount, const int outIdx, const V3 fin) {
  float r = glclamp(fin.x, 0.0f, 1.0f), g = glclamp(fin.y, 0.0f, 1.0f), b = glclamp(fin.z, 0.0f, 1.0f);
  if (PLANES) {
    P.accum[outIdx] = make_float4(r, g, b, 0.0f);
    return;
  }
  if (P.accum) {
    float4 sum = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    if (accumCount > 1) sum = P.accum[outIdx];
    sum = make_float4(sum.x + r, sum.y + g, sum.z + b, 0.0f);
    P.accum[outIdx] = sum;
    if (accumCount > 1) {
      const float n = (float)accumCount;
      r = sum.x / n; g = sum.y / n; b = sum.z / n;
    }
  }
  P.outRgba8[outIdx] = unorm8(r) | (unorm8(g) << 8) | (unorm8(b) << 16) | 0xFF000000u;
  if (P.outF32) P.outF32[outIdx] = make_float4(r, g, b, 1.0f);
}

// The sample planes of one launch added to the running sum in sample order, one pixel per thread: the additions, the
// division and the stores write_pixel performs when every sample has a launch of its own, on the same operands.
__global__ void __launch_bounds__(256) pt_sum_samples_kernel(const float4 *planes, const int planeStride, const int nSamples, float4 *accum,
                                                             const int countBefore, uint32_t *outRgba8, float4 *outF32) {
  const int i = (int)(blockIdx.x * 256u + threadIdx.x);
  if (i >= planeStride) return;
  float4 sum = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
  if (countBefore > 0) sum = accum[i];
  const float4 *p = planes + i;
#pragma unroll 8
  for (int s = 0; s < nSamples; ++s) {
    const float4 c = p[(size_t)s * (size_t)planeStride];
    sum = make_float4(sum.x + c.x, sum.y + c.y, sum.z + c.z, 0.0f);
  }
  accum[i] = sum;
  float r = sum.x, g = sum.y, b = sum.z;
  const int count = countBefore + nSamples;
  if (count > 1) {
    const float n = (float)count;
    r = sum.x / n; g = sum.y / n; b = sum.z / n;
  }
  outRgba8[i] = unorm8(r) | (unorm8(g) << 8) | (unorm8(b) << 16) | 0xFF000000u;
  if (outF32) outF32[i] = make_float4(r, g, b, 1.0f);
}

// Frames without solids (no sphere, no ellipsoid, a volume): every pixel is its camera ray, the march of that ray
// through the volume and the background behind it (traceIterative when traceRay finds nothing: raytrace.comp:610-627,
// 672-679).  One wave per 8x8 tile, four tiles per workgroup, no state machine and no tickets: the kernel needs a
// third of the general kernel's registers, so twice as many waves are resident to cover the latency of the voxel
// loads, which is what bounds the march there.
// KIND: see march_volume; the host launches the variant the volume and the filter qualify for.
// SAMPLES: the variant for launches that draw several samples of every pixel (P.sampleCount): a wave takes one
// (tile, sample) pair, the samples of a tile next to each other, and the colour goes to the sample's plane.
// (rx0, ry0) - (rx1, ry1): pixels outside this rectangle cannot see the box (the host projects the box's corners and
// adds a margin of two pixels; the whole frame where it cannot tell).  A tile without a pixel inside skips the march and
// its slab test: the march's answer for such a ray is (-1, -1, -1) with no sample taken.
template <int KIND, bool SAMPLES = false>
__global__ void __launch_bounds__(256) pt_volume_frame_kernel(const FrameParams P, const int rx0, const int ry0, const int rx1, const int ry1) {
  __shared__ float s_div255[256];
  s_div255[threadIdx.x] = (float)threadIdx.x / 255.0f; // the workgroup has 256 threads
  __syncthreads();
  int tile = (int)blockIdx.x * 4 + (int)(threadIdx.x >> 6);
  int sample = 0;
  if (SAMPLES) {
    if (tile >= P.numTiles * P.sampleCount) return;
    const int q = tile / P.sampleCount;
    sample = tile - q * P.sampleCount;
    tile = q;
  }
  if (tile >= P.numTiles) return;
  int x, ly, gy;
  if (!tile_pixel(P, tile, (int)(threadIdx.x & 63), x, ly, gy)) return;
  const V3 curD = camera_dir(P, x, gy, P.sampleIndex + sample), rO = ld3(P.cam);
  V3 fin = mk3(0, 0, 0);
  float thr = 1.0f;
  unsigned int steps = 0u;
  V3 vol = mk3(-1.0f, -1.0f, -1.0f);
  const bool maySee = x + 1 >= rx0 && x <= rx1 && gy + 1 >= ry0 && gy <= ry1; // a sample lies in [x, x + 1) x [gy, gy + 1)
  if (__builtin_amdgcn_ballot_w64(maySee) != 0ull)
    vol = march_volume<true, KIND>(P, P, s_div255, [&]() -> float4 { return P.image[P.offMat + 5 * P.volMaterial + 4]; }, rO, curD,
                                   kMaxDistance, steps);
  float volumeAlpha = 0.0f;
  if (vol.x >= 0.0f) volumeAlpha = glclamp(length3(vol) / length3(mk3(1.0f, 1.0f, 1.0f)), 0.0f, 1.0f); // :618-620
  if (volumeAlpha > 0.0f) {                                                                              // :624-627
    fin = add3(fin, scale3(mk3(thr * vol.x, thr * vol.y, thr * vol.z), volumeAlpha));
    thr = thr * (1.0f - volumeAlpha);
  }
  { // :672-679
    const V3 unitDir = normalize3(curD);
    const float t = 0.5f * (unitDir.y + 1.0f);
    const float omt = 1.0f - t;
    const V3 bg = mk3(P.bgBottom[0] * omt + P.bgTop[0] * t, P.bgBottom[1] * omt + P.bgTop[1] * t, P.bgBottom[2] * omt + P.bgTop[2] * t);
    fin = add3(fin, scale3(bg, thr));
  }
  write_pixel<SAMPLES>(P, P.accumCount, ly * P.width + x + (SAMPLES ? sample * P.planeStride : 0), fin);
}

} // namespace ptdev
