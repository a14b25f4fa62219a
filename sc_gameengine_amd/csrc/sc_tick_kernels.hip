// sc_tick_kernels.hip -- hand-written gfx950 (CDNA4, wave64) kernels of the world tick.
//
// Build with -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt: the reference is
// /fp:precise C++ (no FMA contraction, correctly rounded sqrt) and visibility is decided by an
// fp32 compare, so every rounding here has to be the host's rounding.
//
// The path is HBM-bound integer/fp32 streaming work (3-5 flop/B): no MFMA.  What matters is
// coalesced SoA streams (64 lanes x 4 B = 256 B per wave load, 64 x 16 B = 1 KiB per row store),
// no inter-workgroup dependency inside the big kernel (a child re-derives its ancestors' matrices
// instead of waiting for them), and wave-ballot compaction.
#include "sc_tick_internal.h"
#include <hip/hip_ext.h>
#include "../../include/sc_tick.h"


namespace sctick {

// The pair role is bound by instruction issue and LDS latency: five waves per SIMD hide it measurably better than four
// (config 5: 66 -> 55 us for the end-of-tick kernel).  Its rarely entered overflow phase would push the allocation to 107
// VGPRs (four waves); held to five waves the compiler spills eight dwords instead, on that rare path.
#ifndef SC_XFORM_OCC
#define SC_XFORM_OCC      // (an occupancy attribute for the fused kernel, for experiments: seven and eight waves per SIMD by force both lost -- profiles/r03/ab_eight_waves.log, profiles/r04/ab_fused_kernel_slp_fma.log)
#endif
#ifndef SC_PAIR_OCC
#define SC_PAIR_OCC __attribute__((amdgpu_waves_per_eu(5, 5)))
#endif

// ------------------------------------------------------------------------------------------
// 3x4 affine helpers.  Row r of the world matrix is (M[r,0], M[r,1], M[r,2], M[r,3]); the
// reference's column-major Mat4 has m[c*4+r] == M[r,c] and a constant last row (0,0,0,1).
// ------------------------------------------------------------------------------------------
struct Aff { float r0[4], r1[4], r2[4]; };

__device__ __forceinline__ bool dirtyBit(const uint32_t* __restrict__ bits, uint32_t i)
{
  return (bits[i >> 5] >> (i & 31u)) & 1u;
}

// slab accessors: base + (stream * capBytes + i * 4) with 32-bit offsets (saddr + voffset addressing)
__device__ __forceinline__ float ldF(const DeviceState& d, uint32_t stream, uint32_t i)
{
  return *reinterpret_cast<const float*>(d.fslab + (stream * d.capBytes + i * 4u));
}
__device__ __forceinline__ uint32_t ldU(const DeviceState& d, uint32_t stream, uint32_t i)
{
  return *reinterpret_cast<const uint32_t*>(d.fslab + (stream * d.capBytes + i * 4u));
}
__device__ __forceinline__ float4 ldRow(const DeviceState& d, uint32_t row, uint32_t i)
{
  return *reinterpret_cast<const float4*>(d.rslab + (row * d.capBytes16 + i * 16u));
}
typedef float V4f __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void storeStream16(void* p, const float4& v)
{
  // written once per tick and not read again before the NEXT launch: a non-temporal store does not leave the line dirty in
  // L2, which the following kernel boundary would otherwise have to write back before the end-of-tick kernel starts
  V4f x = { v.x, v.y, v.z, v.w };
  __builtin_nontemporal_store(x, reinterpret_cast<V4f*>(p));
}
__device__ __forceinline__ void stRow(const DeviceState& d, uint32_t row, uint32_t i, const float4& v)
{
  *reinterpret_cast<float4*>(d.rslab + (row * d.capBytes16 + i * 16u)) = v;
}

// local = T * (R * S) with R = (Rz * Ry) * Rx  (sc_math.cpp:100-142).
// The reference forms it with four full 4x4 products; multiplying by the exact 0 / 1 entries of
// T, S and the axis rotations contributes +-0 terms or x*1, so the non-zero results below are the
// same fp32 values (each sum keeps the reference's left-to-right order); only the sign of an
// exact zero can differ.
// lkj = link word of entity j: its kRotTrivial* bits say which axes were uploaded with sin == 0 and
// cos == 1 exactly; those constants then replace the loads (same values, same arithmetic, same bits).
__device__ __forceinline__ Aff loadLocal(const DeviceState& d, uint32_t j, uint32_t lkj)
{
  const uint32_t triv = lkj >> 29;                      // bit0 X, bit1 Y, bit2 Z
  float sx = 0.0f, cx = 1.0f, sy = 0.0f, cy = 1.0f, sz = 0.0f, cz = 1.0f;
  if (!(triv & 1u)) { sx = ldF(d, kRSX, j); cx = ldF(d, kRCX, j); }
  if (!(triv & 2u)) { sy = ldF(d, kRSY, j); cy = ldF(d, kRCY, j); }
  if (!(triv & 4u)) { sz = ldF(d, kRSZ, j); cz = ldF(d, kRCZ, j); }
  const float kx = ldF(d, kSX, j), ky = ldF(d, kSY, j), kz = ldF(d, kSZ, j);
  // A = Rz * Ry
  const float a00 = cz * cy, a10 = sz * cy, a20 = -sy;
  const float a01 = -sz,     a11 = cz;                 // a21 = 0
  const float a02 = cz * sy, a12 = sz * sy, a22 = cy;
  // R = A * Rx : col0 = A col0; col1 = A1*cx + A2*sx; col2 = A1*(-sx) + A2*cx
  const float nsx = -sx;
  const float r01 = a01 * cx + a02 * sx, r11 = a11 * cx + a12 * sx, r21 = a22 * sx;
  const float r02 = a01 * nsx + a02 * cx, r12 = a11 * nsx + a12 * cx, r22 = a22 * cx;
  Aff L;
  L.r0[0] = a00 * kx; L.r0[1] = r01 * ky; L.r0[2] = r02 * kz; L.r0[3] = ldF(d, kPX, j);
  L.r1[0] = a10 * kx; L.r1[1] = r11 * ky; L.r1[2] = r12 * kz; L.r1[3] = ldF(d, kPY, j);
  L.r2[0] = a20 * kx; L.r2[1] = r21 * ky; L.r2[2] = r22 * kz; L.r2[3] = ldF(d, kPZ, j);
  return L;
}

// world = parent * local  (mat4_mul, sc_math.cpp:52-68): ((p0*l0 + p1*l1) + p2*l2) + p3*l3 with
// l3 the local's last row (0,0,0,1): the fourth term is +-0 for columns 0..2 and p3 for column 3.
__device__ __forceinline__ void mulRow(const float p[4], const Aff& L, float out[4])
{
  out[0] = (p[0] * L.r0[0] + p[1] * L.r1[0]) + p[2] * L.r2[0];
  out[1] = (p[0] * L.r0[1] + p[1] * L.r1[1]) + p[2] * L.r2[1];
  out[2] = (p[0] * L.r0[2] + p[1] * L.r1[2]) + p[2] * L.r2[2];
  out[3] = ((p[0] * L.r0[3] + p[1] * L.r1[3]) + p[2] * L.r2[3]) + p[3];
}
__device__ __forceinline__ Aff mulAff(const Aff& P, const Aff& L)
{
  Aff W;
  mulRow(P.r0, L, W.r0); mulRow(P.r1, L, W.r1); mulRow(P.r2, L, W.r2);
  return W;
}

__device__ __forceinline__ Aff loadRows(const DeviceState& d, uint32_t j)
{
  const float4 a = ldRow(d, 0, j), b = ldRow(d, 1, j), c = ldRow(d, 2, j);
  Aff M;
  M.r0[0] = a.x; M.r0[1] = a.y; M.r0[2] = a.z; M.r0[3] = a.w;
  M.r1[0] = b.x; M.r1[1] = b.y; M.r1[2] = b.z; M.r1[3] = b.w;
  M.r2[0] = c.x; M.r2[1] = c.y; M.r2[2] = c.z; M.r2[3] = c.w;
  return M;
}
__device__ __forceinline__ void storeRows(const DeviceState& d, uint32_t j, const Aff& M)
{
  stRow(d, 0, j, make_float4(M.r0[0], M.r0[1], M.r0[2], M.r0[3]));
  stRow(d, 1, j, make_float4(M.r1[0], M.r1[1], M.r1[2], M.r1[3]));
  stRow(d, 2, j, make_float4(M.r2[0], M.r2[1], M.r2[2], M.r2[3]));
}

struct BoundsCE { float cx, cy, cz, ex, ey, ez; };
__device__ __forceinline__ BoundsCE loadBounds(const DeviceState& d, uint32_t j)
{
  const float x0 = ldF(d, kBMINX, j), y0 = ldF(d, kBMINY, j), z0 = ldF(d, kBMINZ, j);
  const float x1 = ldF(d, kBMAXX, j), y1 = ldF(d, kBMAXY, j), z1 = ldF(d, kBMAXZ, j);
  BoundsCE b;
  b.cx = (x0 + x1) * 0.5f; b.cy = (y0 + y1) * 0.5f; b.cz = (z0 + z1) * 0.5f;
  b.ex = (x1 - x0) * 0.5f; b.ey = (y1 - y0) * 0.5f; b.ez = (z1 - z0) * 0.5f;
  return b;
}

// computeWorldBoundsSphere + sphereInFrustum (sc_world_partition.cpp:1105-1144).  All six planes
// are evaluated; OR-ing "d < -radius" equals the reference's early-out (NaN compares false in both).
__device__ __forceinline__ bool sphereVisible(const Aff& M, const BoundsCE& b, const Frustum6& fr,
                                              float& c0, float& c1, float& c2)
{
  c0 = M.r0[0] * b.cx + M.r0[1] * b.cy + M.r0[2] * b.cz + M.r0[3];
  c1 = M.r1[0] * b.cx + M.r1[1] * b.cy + M.r1[2] * b.cz + M.r1[3];
  c2 = M.r2[0] * b.cx + M.r2[1] * b.cy + M.r2[2] * b.cz + M.r2[3];
  // max of the three column norms (std::max(a,b) == (a<b)?b:a): the correctly rounded square root is monotone and keeps
  // NaN, so selecting among the SQUARED norms with the same comparisons and taking one root gives the value the
  // reference gets from three roots (equal roots of unequal squares are the same value either way)
  const float nx = M.r0[0] * M.r0[0] + M.r1[0] * M.r1[0] + M.r2[0] * M.r2[0];
  const float ny = M.r0[1] * M.r0[1] + M.r1[1] * M.r1[1] + M.r2[1] * M.r2[1];
  const float nz = M.r0[2] * M.r0[2] + M.r1[2] * M.r1[2] + M.r2[2] * M.r2[2];
  const float nyz = (ny < nz) ? nz : ny;
  const float maxScale = sqrtf((nx < nyz) ? nyz : nx);
  const float radius = sqrtf(b.ex * b.ex + b.ey * b.ey + b.ez * b.ez) * maxScale;
  bool out = false;
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    const float dist = fr.p[k][0] * c0 + fr.p[k][1] * c1 + fr.p[k][2] * c2 + fr.p[k][3];
    out = out || (dist < -radius);
  }
  return !out;
}

// Same test with the planes read through a uniform pointer into the constant address space: scalar loads at
// the point of use.  The fused kernel uses it so that the 24 coefficients are not held in SGPRs across the tile.
typedef const __attribute__((address_space(4))) float* ConstF;
__device__ __forceinline__ bool sphereVisibleAt(const Aff& M, const BoundsCE& b, ConstF fr)
{
  const float c0 = M.r0[0] * b.cx + M.r0[1] * b.cy + M.r0[2] * b.cz + M.r0[3];
  const float c1 = M.r1[0] * b.cx + M.r1[1] * b.cy + M.r1[2] * b.cz + M.r1[3];
  const float c2 = M.r2[0] * b.cx + M.r2[1] * b.cy + M.r2[2] * b.cz + M.r2[3];
  // (one root of the largest squared column norm instead of three roots: see sphereVisible)
  const float nx = M.r0[0] * M.r0[0] + M.r1[0] * M.r1[0] + M.r2[0] * M.r2[0];
  const float ny = M.r0[1] * M.r0[1] + M.r1[1] * M.r1[1] + M.r2[1] * M.r2[1];
  const float nz = M.r0[2] * M.r0[2] + M.r1[2] * M.r1[2] + M.r2[2] * M.r2[2];
  const float nyz = (ny < nz) ? nz : ny;
  const float maxScale = sqrtf((nx < nyz) ? nyz : nx);
  const float radius = sqrtf(b.ex * b.ex + b.ey * b.ey + b.ez * b.ez) * maxScale;
  bool out = false;
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    const float dist = fr[4 * k] * c0 + fr[4 * k + 1] * c1 + fr[4 * k + 2] * c2 + fr[4 * k + 3];
    out = out || (dist < -radius);
  }
  return !out;
}

// world AABB of the bounds box (broadphase spec): centre as above, half extent
// h_r = |M[r,0]|*ex + |M[r,1]|*ey + |M[r,2]|*ez
__device__ __forceinline__ void worldAabb(const Aff& M, const BoundsCE& b, float mn[3], float mx[3])
{
  const float c0 = M.r0[0] * b.cx + M.r0[1] * b.cy + M.r0[2] * b.cz + M.r0[3];
  const float c1 = M.r1[0] * b.cx + M.r1[1] * b.cy + M.r1[2] * b.cz + M.r1[3];
  const float c2 = M.r2[0] * b.cx + M.r2[1] * b.cy + M.r2[2] * b.cz + M.r2[3];
  const float h0 = fabsf(M.r0[0]) * b.ex + fabsf(M.r0[1]) * b.ey + fabsf(M.r0[2]) * b.ez;
  const float h1 = fabsf(M.r1[0]) * b.ex + fabsf(M.r1[1]) * b.ey + fabsf(M.r1[2]) * b.ez;
  const float h2 = fabsf(M.r2[0]) * b.ex + fabsf(M.r2[1]) * b.ey + fabsf(M.r2[2]) * b.ez;
  mn[0] = c0 - h0; mn[1] = c1 - h1; mn[2] = c2 - h2;
  mx[0] = c0 + h0; mx[1] = c1 + h1; mx[2] = c2 + h2;
}

// ------------------------------------------------------------------------------------------
// Broadphase binning.  A box is entered into every sector its xz range
// [floor(min*inv), floor(max*inv)] touches (worldToSector arithmetic, sc_world_partition.cpp:268-275)
// when that is at most 2x2 sectors inside the grid rectangle; a pair is later reported only from
// the sector holding the low corner of the two boxes' intersection, so every bin is self-contained:
// no halo, no duplicate.  A box that finds a sector's bin full goes to the sector OVERFLOW list, tagged with
// that sector (the wave that searches the sector gathers its overflow back and treats it like the bin's
// records); a box larger than 2x2 sectors or outside the rectangle goes to the "big" list, which the pair
// kernel tests against everything.
// ------------------------------------------------------------------------------------------
struct BinPlan { bool collide, big; float x0, z0; uint32_t nx, nz; };

__device__ __forceinline__ BinPlan planBins(const TickParams& p, const float mn[3], const float mx[3])
{
  BinPlan b; b.collide = false; b.big = false; b.x0 = b.z0 = 0.0f; b.nx = b.nz = 0;
  // a NaN bound overlaps nothing under the closed-interval test: no collider
  if (!(mn[0] == mn[0] && mn[1] == mn[1] && mn[2] == mn[2] && mx[0] == mx[0] && mx[1] == mx[1] && mx[2] == mx[2])) return b;
  b.collide = true;
  const float x0 = floorf(mn[0] * p.invSector) - p.binOx, x1 = floorf(mx[0] * p.invSector) - p.binOx;
  const float z0 = floorf(mn[2] * p.invSector) - p.binOz, z1 = floorf(mx[2] * p.invSector) - p.binOz;
  const float nx = x1 - x0 + 1.0f, nz = z1 - z0 + 1.0f;
  const bool ok = x0 >= 0.0f && z0 >= 0.0f && x1 < (float)p.binSX && z1 < (float)p.binSZ &&
                  nx >= 1.0f && nx <= 2.0f && nz >= 1.0f && nz <= 2.0f;
  if (!ok) { b.big = true; return b; }
  b.x0 = x0; b.z0 = z0; b.nx = (uint32_t)nx; b.nz = (uint32_t)nz;
  return b;
}

__device__ __forceinline__ void appendBig(const DeviceState& d, const TickParams& p, const float4& rmin, const float4& rmax)
{
  const uint32_t slot = atomicAdd(&d.counters[kCtrPar + 8u * p.parity + kCtrBig], 1u);
  if (slot >= p.bigCap) return;                    // cannot happen with a sane counter (one entry per entity at most): never write outside
  d.bigList[2u * (size_t)slot] = rmin;
  d.bigList[2u * (size_t)slot + 1u] = rmax;
}

// A sector becomes CROWDED the moment slot 64 of its bin is handed out (its counter keeps counting; the record goes to the
// sector overflow list).  Whoever is handed that slot -- exactly one record per sector and tick -- puts the sector on the tick's
// queue of crowded sectors, so that the queue is complete before the pair search starts (it is a launch of its own).
__device__ __forceinline__ void queueCrowded(const DeviceState& d, const TickParams& p, uint32_t sector)
{
  const uint32_t at = atomicAdd(&d.counters[kCtrPar + 8u * p.parity + kCtrCrowdTail], 1u);
  const uint32_t sectors = p.binSX * p.binSZ;
  if (at < sectors) d.crowdQueue[p.parity * sectors + at] = sector;          // (one entry per sector at most: always room)
}

// One record into one sector's bin by the lane itself (or into the sector overflow list when the bin is full).
__device__ __forceinline__ void binInsertLane(const DeviceState& d, const TickParams& p, uint32_t sector, const float4& rmin, const float4& rm)
{
  const uint32_t slot = atomicAdd(&d.binCount[sector], 1u);
  atomicOr(&d.binLayers[sector], __float_as_uint(rmin.w));
  if (slot < kBinCap) {
    float4* r = d.bins + 2u * ((size_t)sector * kBinCap + slot);
    r[0] = rmin; r[1] = rm;
  } else {
    if (slot == kBinCap) queueCrowded(d, p, sector);
    const uint32_t ctr = kCtrPar + 8u * p.parity;
    const uint32_t at = atomicAdd(&d.counters[ctr + kCtrSpill], 1u);
    if (at < p.ovfCap) {
      d.spill[2u * (size_t)at] = rmin; d.spill[2u * (size_t)at + 1u] = rm; d.spillSector[at] = sector;
      atomicMin(&d.ovfLo[sector], at); atomicMax(&d.ovfHi[sector], at + 1u);
    } else atomicAdd(&d.counters[ctr + kCtrBorderLost], 1u);               // (cannot happen with the list sized by ovfRecords(): never silent anyway)
  }
}

// a record that overlaps nothing and passes no filter: a remembered bin slot whose copy does not exist this tick; padding of a border message
__device__ __forceinline__ void nullRecord(float4& lo, float4& hi, uint32_t id = 0x00FFFFFFu)
{
  lo = make_float4(INFINITY, INFINITY, INFINITY, __uint_as_float(0u));
  hi = make_float4(-INFINITY, -INFINITY, -INFINITY, __uint_as_float(id));
}

// Lazy records.  The pair search reads a bin only if two of its records can pass the group/mask filter against each other
// (or a big box is about); in a city most bins hold static props only, which never do.  The records a bin's OWNERS keep in
// their remembered slots (home slots, below) have a layer summary that is known from the learn tick on -- homeLayers -- so
// when that summary admits no pair, the owners do not write their records at all on the ticks in between: nothing would read
// them.  If a record from elsewhere then makes such a bin admissible after all (a vehicle drives in; a big box appears), the
// wave that searches the bin rebuilds the owners' records from their world matrices (rebuildHomeRecord: the arithmetic of
// binEntityWave) -- rare, and only where it is needed.  Ring sectors are always written: the border pack copies them.
// a summary that passes the "two of these records can meet" test exactly when a record of the bin (summary H) and a record of
// the world's vocabulary V (which includes the bin's own) can: groups of one side against masks of the other, both ways
__host__ __device__ __forceinline__ uint32_t layersThatCanMeet(uint32_t H, uint32_t V)
{
  V |= H;
  const bool meet = ((V & 0xFFFFu) & (H >> 16)) != 0u && ((H & 0xFFFFu) & (V >> 16)) != 0u;
  return meet ? 0xFFFFFFFFu : 0u;
}
__device__ __forceinline__ bool binWrittenEveryTick(uint32_t homeLayers, uint32_t sector, uint32_t binSX, uint32_t binSZ)
{
  const uint32_t gx = sector % binSX, gz = sector / binSX;
  return ((homeLayers & 0xFFFFu) & (homeLayers >> 16)) != 0u || gx == 0u || gz == 0u || gx + 1u == binSX || gz + 1u == binSZ;
}

// Whole-wave broadphase step for one entity per lane: world AABB -> bins / big list.
//
// Round 3: ONE reservation round trip per tile.  Ablations of the round-2 form (profiles/r03/ab_ablation_binning.log) showed
// that binning was 15 of the fused kernel's 36 us, and that its cost was not bytes: the copies of a box in the sectors next to
// its own -- every ground slab has three, a straddling prop one -- each reserved their slot with a returning atomic of their
// own, one AFTER the other behind the primary copy's (four dependent round trips to L2 at the end of every tile: 5.8 us);
// the primary's returning atomic another 2.9.  Now every returning atomic of the tile -- the run heads' for the primary
// copies, up to three per straddling lane for the others -- is issued before any result is used, the layer summaries'
// atomicOr (nothing returns) behind them, and the stores follow one wait.
__device__ __forceinline__ void spillLane(const DeviceState& d, const TickParams& p, uint32_t sector, uint32_t slot, const float4& rmin, const float4& rm)
{
  // one record that found its sector's bin full (it was handed `slot` >= 64): it joins the sector overflow list
  if (slot == kBinCap) queueCrowded(d, p, sector);
  const uint32_t ctr = kCtrPar + 8u * p.parity;
  const uint32_t at = atomicAdd(&d.counters[ctr + kCtrSpill], 1u);
  if (at < p.ovfCap) {
    d.spill[2u * (size_t)at] = rmin; d.spill[2u * (size_t)at + 1u] = rm; d.spillSector[at] = sector;
    atomicMin(&d.ovfLo[sector], at); atomicMax(&d.ovfHi[sector], at + 1u);
  } else atomicAdd(&d.counters[ctr + kCtrBorderLost], 1u);               // (cannot happen with the list sized by ovfRecords(): never silent anyway)
}

// storeM: the lane's freshly built world matrix is stored HERE, behind the reservations (they do not depend on it): the wait
// for the reservations then does not include the stores' way to memory (vmcnt retires in order), and the bounds loads in
// front of this call are not held behind the stores either.
//
// HOME SLOTS (round 3).  A reservation is a returning atomic, and a global atomic executes at the memory side of this chip
// (MI355X_MICROARCH.md, Global atomics): ~1 us under load, a read-modify-write of a 64-byte line per lane.  With every
// reservation switched off -- records stored at made-up slots -- the fused kernel ran 28 instead of 39 us
// (profiles/r03/ab_noatomics.log).  Boxes rarely change sector from one tick to the next, so the slots are REMEMBERED: on a
// learn tick (TickParams::homeMode 1; the first tick, then now and then) every record reserves its slot as before and the lane
// notes, per entity, its primary sector and the slots of its up to four copies (homeA / homeB); a small kernel then notes how
// many slots of each bin were handed out (homeCount) and the layer summary they gave (homeLayers), and the pair search leaves
// the bins' counters at those values instead of zero.  On the ticks in between (homeMode 2) a record whose sector is still
// the one its slot was reserved in is stored straight there -- no atomic, no wait; a copy that no longer exists (the box left
// the sector, shrank, lost its bounds) stores a NULL record in its slot, so that every reserved slot is written by its owner
// on every tick and no parity copy of the bins ever shows a stale record; a copy without a reservation (the box entered a
// new sector, a bin was full at the learn tick, level kernels, border records) reserves behind the remembered slots as before.
// Same records in the same bins -- the pair search sees a few null records more, which pass no filter and overlap nothing.
template <uint32_t kHome>
__device__ __forceinline__ void binEntityWave(const DeviceState& d, const TickParams& p, uint32_t i, bool collider,
                                              const Aff& M, const BoundsCE& b, bool storeM, bool lazyOn)
{
  float mn[3] = {0, 0, 0}, mx[3] = {0, 0, 0};
  BinPlan plan; plan.collide = false; plan.big = false; plan.x0 = plan.z0 = 0.0f; plan.nx = plan.nz = 0;
  float4 rmin = make_float4(0, 0, 0, 0), rmax = make_float4(0, 0, 0, 0);
  constexpr uint32_t mode = kHome;                       // (a kernel instance per mode: the learn tick's bookkeeping costs the other ticks no registers)
  uint32_t hA = kNoHome, hB = 0xFFFFFFFFu;
  if (mode == kHomeUse && i < p.n) { hA = d.homeA[i]; hB = d.homeB[i]; }
  if (collider) {
    worldAabb(M, b, mn, mx);
    plan = planBins(p, mn, mx);
    rmin = make_float4(mn[0], mn[1], mn[2], __uint_as_float(ldU(d, kLAYERS, i)));
    rmax = make_float4(mx[0], mx[1], mx[2], __uint_as_float(i | p.rankBits));
  }
  const bool want = plan.collide && !plan.big;
  const uint32_t sector = (uint32_t)plan.z0 * p.binSX + (uint32_t)plan.x0;
  const uint32_t lane = threadIdx.x & 63u;
  // remembered slots count while the box's primary sector is the one they were reserved from
  const bool atHome = want && hA == sector;
  const bool r0 = atHome && (hB & 0xFFu) != kNoSlot;
  const bool a0 = want && !r0;                                     // the primary copy reserves its slot
  uint32_t slot = hB & kSlotMask;
  // lazy records: a remembered slot of a bin that cannot produce a pair is not written (binWrittenEveryTick)
  // ... and neither is a slot that is written on every tick when its owner's matrix did not change: it holds this very record
  // (TickParams::cleanStay; storeM = the matrix was rebuilt)
  const bool stay = p.cleanStay != 0u && !storeM;
  const bool skip0 = r0 && ((hB & kSlotAlways) ? stay : lazyOn);
  float4 rm = rmax;
  rm.w = __uint_as_float(i | p.rankBits | kPrimary);
  const uint32_t myLay = __float_as_uint(rmin.w);

  const unsigned long long act = ballot64(a0);
  uint32_t myHead = lane, runEnd = lane + 1u;
  if (act) {
    // ---- primary copies that reserve: consecutive lanes that target the same sector form a run, whose first lane reserves
    // the slots of the whole run
    const uint32_t key = a0 ? sector : 0xFFFFFFFFu;
    const uint32_t prev = __shfl_up(key, 1, 64);
    const bool head = a0 && (lane == 0 || prev != key);
    const unsigned long long heads = ballot64(head);
    const unsigned long long upto = (lane == 63u) ? ~0ull : ((2ull << lane) - 1ull);
    myHead = (63u - (uint32_t)__clzll(heads & upto)) & 63u;
    const unsigned long long above = (myHead == 63u) ? 0ull : ~((2ull << myHead) - 1ull);
    const unsigned long long ends = (heads | ~act) & above;
    runEnd = ends ? (uint32_t)__ffsll((long long)ends) - 1u : 64u;
    // OR of the run's collision layers for the head lane.  Runs are almost always uniform (one layer
    // word): one cross-lane read decides; only a wave with a mixed run pays the segmented OR.
    uint32_t lay = a0 ? myLay : 0u;
    const uint32_t headLay = (uint32_t)__shfl((int)lay, (int)myHead, 64);
    if (ballot64(a0 && headLay != lay)) {
#pragma unroll
      for (uint32_t o = 1; o < 64u; o <<= 1) {
        const uint32_t other = (uint32_t)__shfl_down((int)lay, o, 64);
        if (a0 && lane + o < runEnd) lay |= other;
      }
    }
    uint32_t base = 0;
    if (head) { base = atomicAdd(&d.binCount[sector], runEnd - lane); atomicOr(&d.binLayers[sector], lay); }
    if (storeM) storeRows(d, i, M);
    base = __shfl(base, myHead, 64);
    if (a0) slot = base + (lane - myHead);
  } else if (storeM) storeRows(d, i, M);
  if (want && slot < kBinCap && !skip0) {
    float4* r = d.bins + 2u * ((size_t)sector * kBinCap + slot);
    r[0] = rmin; r[1] = rm;
  }
  uint32_t learned = (want && slot < kBinCap) ? slot : kNoSlot;    // (learn tick: what homeB will hold)
  // ---- the copies in the neighbouring sectors (a few lanes per wave; every ground slab).  With remembered slots (the usual
  // case on a kHomeUse tick) a copy is a plain store; on the other ticks the (up to three) reservations of a lane are issued
  // together and waited for once.
  const bool c1 = want && plan.nx > 1u, c2 = want && plan.nz > 1u, c3 = c1 && c2;
  if (mode == kHomeUse) {
    // (the reservations of the copies that have no remembered slot -- boxes that entered a sector, crowded bins -- are issued
    //  together and waited for once, like on the other ticks: one after the other they cost config 5's fused kernel 2.3 us)
    const uint32_t sec1 = sector + 1u, sec2 = sector + p.binSX, sec3 = sec2 + 1u;
    const uint32_t b1 = (hB >> 8) & 0xFFu, b2 = (hB >> 16) & 0xFFu, b3 = hB >> 24;
    uint32_t q1 = b1 & kSlotMask, q2 = b2 & kSlotMask, q3 = b3 & kSlotMask;
    const bool h1 = atHome && b1 != kNoSlot, h2 = atHome && b2 != kNoSlot, h3 = atHome && b3 != kNoSlot;    // remembered
    const bool a1 = c1 && !h1, a2 = c2 && !h2, a3 = c3 && !h3;
    const bool w1 = c1 && !(h1 && ((b1 & kSlotAlways) ? stay : lazyOn)), w2 = c2 && !(h2 && ((b2 & kSlotAlways) ? stay : lazyOn)),
               w3 = c3 && !(h3 && ((b3 & kSlotAlways) ? stay : lazyOn));
    if (a1) q1 = atomicAdd(&d.binCount[sec1], 1u);
    if (a2) q2 = atomicAdd(&d.binCount[sec2], 1u);
    if (a3) q3 = atomicAdd(&d.binCount[sec3], 1u);
    if (w1 && q1 < kBinCap) { float4* r = d.bins + 2u * ((size_t)sec1 * kBinCap + q1); r[0] = rmin; r[1] = rmax; }
    if (w2 && q2 < kBinCap) { float4* r = d.bins + 2u * ((size_t)sec2 * kBinCap + q2); r[0] = rmin; r[1] = rmax; }
    if (w3 && q3 < kBinCap) { float4* r = d.bins + 2u * ((size_t)sec3 * kBinCap + q3); r[0] = rmin; r[1] = rmax; }
    if (a1) atomicOr(&d.binLayers[sec1], myLay);
    if (a2) atomicOr(&d.binLayers[sec2], myLay);
    if (a3) atomicOr(&d.binLayers[sec3], myLay);
    if (ballot64((a1 && q1 >= kBinCap) || (a2 && q2 >= kBinCap) || (a3 && q3 >= kBinCap))) {
      if (a1 && q1 >= kBinCap) spillLane(d, p, sec1, q1, rmin, rmax);
      if (a2 && q2 >= kBinCap) spillLane(d, p, sec2, q2, rmin, rmax);
      if (a3 && q3 >= kBinCap) spillLane(d, p, sec3, q3, rmin, rmax);
    }
    // remembered slots whose copy does not exist this tick: the owner writes a null record there (under its own id: whoever
    // rebuilds an unwritten bin goes by the ids in the slots)
    if (hA != kNoHome && !(atHome && c1 && c2)) {
      float4 lo, hi; nullRecord(lo, hi, i | p.rankBits);
#pragma unroll
      for (uint32_t k = 0; k < 4u; ++k) {
        const bool ck = k == 0u ? want : (k == 1u ? c1 : (k == 2u ? c2 : c3));
        const uint32_t gk = (hB >> (8u * k)) & 0xFFu;
        if (gk != kNoSlot && !(atHome && ck) && !(lazyOn && !(gk & kSlotAlways))) {
          float4* r = d.bins + 2u * ((size_t)(hA + (k & 1u) + (k >> 1) * p.binSX) * kBinCap + (gk & kSlotMask)); r[0] = lo; r[1] = hi;
        }
      }
    }
  } else {
    const uint32_t sec1 = sector + 1u, sec2 = sector + p.binSX, sec3 = sec2 + 1u;
    uint32_t q1 = 0, q2 = 0, q3 = 0;
    if (c1) q1 = atomicAdd(&d.binCount[sec1], 1u);
    if (c2) q2 = atomicAdd(&d.binCount[sec2], 1u);
    if (c3) q3 = atomicAdd(&d.binCount[sec3], 1u);
    if (c1 && q1 < kBinCap) { float4* r = d.bins + 2u * ((size_t)sec1 * kBinCap + q1); r[0] = rmin; r[1] = rmax; }
    if (c2 && q2 < kBinCap) { float4* r = d.bins + 2u * ((size_t)sec2 * kBinCap + q2); r[0] = rmin; r[1] = rmax; }
    if (c3 && q3 < kBinCap) { float4* r = d.bins + 2u * ((size_t)sec3 * kBinCap + q3); r[0] = rmin; r[1] = rmax; }
    if (c1) atomicOr(&d.binLayers[sec1], myLay);
    if (c2) atomicOr(&d.binLayers[sec2], myLay);
    if (c3) atomicOr(&d.binLayers[sec3], myLay);
    learned |= ((c1 && q1 < kBinCap ? q1 : kNoSlot) << 8) | ((c2 && q2 < kBinCap ? q2 : kNoSlot) << 16) | ((c3 && q3 < kBinCap ? q3 : kNoSlot) << 24);
    if (ballot64((c1 && q1 >= kBinCap) || (c2 && q2 >= kBinCap) || (c3 && q3 >= kBinCap))) {
      if (c1 && q1 >= kBinCap) spillLane(d, p, sec1, q1, rmin, rmax);
      if (c2 && q2 >= kBinCap) spillLane(d, p, sec2, q2, rmin, rmax);
      if (c3 && q3 >= kBinCap) spillLane(d, p, sec3, q3, rmin, rmax);
    }
    // ---- learn tick: remember where this entity's records went (reserved slots inside the bin only)
    if (mode == kHomeLearn && i < p.n) { d.homeA[i] = want ? sector : kNoHome; d.homeB[i] = learned; }
  }
  // bin full: the record joins the sector overflow list (primary copies: one reservation per wave)
  const bool over = a0 && slot >= kBinCap;
  const unsigned long long mo = ballot64(over);
  if (mo) {
    const uint32_t ctr = kCtrPar + 8u * p.parity;
    const uint32_t first = (uint32_t)__ffsll((long long)mo) - 1u;
    uint32_t at = 0;
    if (lane == first) at = atomicAdd(&d.counters[ctr + kCtrSpill], (uint32_t)__popcll(mo));     // (every wave of the chip meets on this word: one atomic, not two)
    at = __shfl(at, (int)first, 64) + (uint32_t)__popcll(mo & ((1ull << lane) - 1ull));
    if (over && slot == kBinCap) queueCrowded(d, p, sector);
    if (over && at < p.ovfCap) {
      d.spill[2u * (size_t)at] = rmin; d.spill[2u * (size_t)at + 1u] = rm; d.spillSector[at] = sector;
      // the overflowing lanes of a run are its tail and their list indices ascend with the lane: the first one lowers the
      // sector's slice bound, the last one raises it (two atomics per run, not per record)
      if (slot == kBinCap || lane == myHead) atomicMin(&d.ovfLo[sector], at);
      if (lane + 1u == runEnd) atomicMax(&d.ovfHi[sector], at + 1u);
    } else if (over) atomicAdd(&d.counters[ctr + kCtrBorderLost], 1u);      // (cannot happen with the list sized by ovfRecords(): never silent anyway)
  }
  if (plan.collide && plan.big) appendBig(d, p, rmin, rmax);
}

// Same for a single lane (level kernels: entities of one level are scattered, no runs to aggregate).
__device__ __forceinline__ void binEntitySingle(const DeviceState& d, const TickParams& p, uint32_t i, const Aff& M, const BoundsCE& b)
{
  float mn[3], mx[3];
  worldAabb(M, b, mn, mx);
  const BinPlan plan = planBins(p, mn, mx);
  const float4 rmin = make_float4(mn[0], mn[1], mn[2], __uint_as_float(ldU(d, kLAYERS, i)));
  float4 rmax = make_float4(mx[0], mx[1], mx[2], __uint_as_float(i | p.rankBits));
  if (!plan.collide) return;
  if (plan.big) { appendBig(d, p, rmin, rmax); return; }
  const uint32_t sx = (uint32_t)plan.x0, sz = (uint32_t)plan.z0;
  for (uint32_t k = 0; k < 4; ++k) {
    const uint32_t dx = k & 1u, dz = k >> 1;
    if (!(dx < plan.nx && dz < plan.nz)) continue;
    float4 rm = rmax; if (k == 0) rm.w = __uint_as_float(i | p.rankBits | kPrimary);
    binInsertLane(d, p, (sz + dz) * p.binSX + (sx + dx), rmin, rm);
  }
}

// ------------------------------------------------------------------------------------------
// K1: fused TransformSystem + CullingSystem mask (+ world AABB), one launch over all entities in
// Transform-pool dense order.
//
// Hierarchy without level barriers: an entity of depth <= kMaxChain walks its own parent chain
// (link words, L1/L2-resident), finds the dirty ancestor nearest the root ("top"), and rebuilds
// world = world'(parent(top)) * local(top) * ... * local(self) left to right -- exactly the products
// TransformSystem's DFS performs (sc_ecs.cpp:178-209), so the result is bit-identical, while no
// workgroup ever waits for another one.  Everything above `top` is clean, hence its stored matrix
// is not written by anyone this tick and can be read race-free.
// ------------------------------------------------------------------------------------------
// (Variants that were measured and dropped -- a wave-cooperative hierarchy resolve through ds_bpermute, bounds
// loaded before the walk, binning with a single atomic round trip -- are in the history at 72ae167; all of
// them lost to this form because they cost VGPRs, and this kernel is occupancy-bound.  DESIGN.md section 5.)
// The body is a device function taking the arguments by reference on purpose: written directly in the
// __global__ function the same code allocates 84/91 VGPRs instead of 64/77 (hipcc, ROCm 7.2).
// kChain = how many ancestors a lane may have to walk: min(deepest level in the world, kMaxChain), known to the
// host after scTickSetTopology.  Specialising on it removes dead levels from worlds that are flat or shallow.
// (Requesting all levels' locals before multiplying -- one round trip instead of one per level -- was measured:
// it needs 12 more VGPRs per level, 104-116 in all, and lost 5-25 %)
template <bool kCull, bool kAabb, uint32_t kChain, uint32_t kHome>
__device__ __forceinline__ void xformCullBody(const DeviceState& d, const TickParams& p);

template <bool kCull, bool kAabb, uint32_t kChain, uint32_t kHome>
__global__ __launch_bounds__(kTile) SC_XFORM_OCC void k_xform_cull(const DeviceState d, const TickParams p) { xformCullBody<kCull, kAabb, kChain, kHome>(d, p); }

template <bool kCull, bool kAabb, uint32_t kChain, uint32_t kHome>
__device__ __forceinline__ void xformCullBody(const DeviceState& d, const TickParams& p)
{
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t wave = threadIdx.x >> 6;
  const uint32_t begin = blockIdx.x * p.span;
  const uint32_t end = (begin + p.span < p.n) ? begin + p.span : p.n;
  const bool doXform = (p.flags & SC_TICK_XFORM) != 0;
  const bool wantCand = (p.flags & SC_TICK_CULLED_LIST) != 0;
  const bool hasDeep = (p.flags & kFlagHasDeep) != 0;

  uint32_t visCount = 0, candCount = 0;      // wave-uniform running sums
  // lazy records (binWrittenEveryTick): off while big boxes are about -- they are tested against every bin; the tick's pair
  // search learns from lazyCtl[1 + parity] what this kernel did
  bool lazyOn = false;
  if (kAabb) {
    // (p.lazy 2: the unwritten bins are those nothing in the world can meet -- big boxes included, so no guard)
    lazyOn = kHome == kHomeUse && (p.lazy == 2u || (p.lazy == 1u && d.lazyCtl[0] == 0u));      // (first needed at the end of the first tile; noted for the pair search below)
  }

  for (uint32_t base = begin; base < end; base += kTile) {
    const uint32_t i = base + threadIdx.x;
    const bool active = i < p.n;
    const uint32_t lk = active ? ldU(d, kLINK, i) : ((kUnreachable << kDepthShift) | kNoParent);
    // requested together with the link word (it must not wait behind the depth test below: that costs
    // a whole memory round trip per tile)
    const uint32_t ownDirtyWord = (doXform && active) ? d.dirty[i >> 5] : 0u;
    const uint32_t depth = linkDepth(lk);
    const bool chain = depth <= kChain;
    // ---- walk up: ancestors a[1..depth], top = dirty level nearest the root
    uint32_t a[kChain + 1];                            // ancestors
    uint32_t rotFlags = lk >> 29;                      // 3 rotation-triviality bits per level, level k at bits 3k..3k+2
    a[0] = i;
    int top = -1;
    if (doXform && chain) {
      if ((ownDirtyWord >> (i & 31u)) & 1u) top = 0;
      uint32_t cur = lk;
#pragma unroll
      for (uint32_t k = 1; k <= kChain; ++k) {
        a[k] = i;
        if (k <= depth) {
          a[k] = cur & kParentMask;
          cur = ldU(d, kLINK, a[k]);
          rotFlags |= (cur >> 29) << (3u * k);
          if (dirtyBit(d.dirty, a[k])) top = (int)k;
        }
      }
    } else {
#pragma unroll
      for (uint32_t k = 1; k <= kChain; ++k) a[k] = i;
    }
    const bool recompute = top >= 0;

    Aff M;
    if (recompute) {
      const bool fromRoot = (uint32_t)top == depth;       // the chain's root itself is rebuilt: world = local
      uint32_t seed = i;
#pragma unroll
      for (uint32_t k = 0; k < kChain; ++k) if ((uint32_t)top == k) seed = a[k + 1];
      // clean parent of the top dirty ancestor: its stored (possibly stale) matrix is the seed
      if (!fromRoot) M = loadRows(d, seed);
#pragma unroll
      for (int lev = (int)kChain; lev >= 0; --lev) {
        if (lev <= top) {
          const Aff L = loadLocal(d, a[lev], (rotFlags >> (3 * lev)) << 29);
          if (lev == top && fromRoot) M = L;
          else M = mulAff(M, L);
        }
      }
      if (!kAabb) storeRows(d, i, M);                     // (with binning the store rides behind the bin reservations: binEntityWave)
    } else if ((kCull || kAabb) && active) {
      M = loadRows(d, i);
    }

    if (hasDeep) {
      const unsigned long long rm = ballot64(recompute);
      if (lane == 0 && (base + wave * 64u) < p.n) d.recomp[(base >> 6) + wave] = rm;
    }

    if (kCull || kAabb) {
      const bool cand = active && (lk & kHasMesh);
      const bool hb = active && (lk & kHasBounds);
      BoundsCE b = {0, 0, 0, 0, 0, 0};
      if (hb) b = loadBounds(d, i);

      if (kCull) {
        bool visible = cand;
        if (cand && hb && !p.freeze && p.frustumValid) {
          const float* fr = d.frustum;
          asm volatile("" : "+s"(fr));                 // opaque: keeps the plane loads inside the loop, at their use
          visible = sphereVisibleAt(M, b, (ConstF)fr);
        }
        // deeper entities get their matrix (and their bit) from the level kernels
        if (doXform && depth > kChain && depth != kUnreachable) visible = false;
        const unsigned long long vm = ballot64(visible);
        const unsigned long long cm = ballot64(cand);
        if (lane == 0 && (base + wave * 64u) < p.n) {
          d.vis[(base >> 6) + wave] = vm;
          if (wantCand) d.cand[(base >> 6) + wave] = cm;
        }
        visCount += (uint32_t)__popcll(vm);
        candCount += (uint32_t)__popcll(cm);
      }
      if (kAabb) {
        // deeper entities are binned by the level kernels once their matrix is final
        const bool collider = hb && !(doXform && depth > kChain && depth != kUnreachable);
        binEntityWave<kHome>(d, p, i, collider, M, b, recompute, lazyOn);
      }
    }
  }

  if (kAabb && blockIdx.x == 0 && threadIdx.x == 0) d.lazyCtl[1u + p.parity] = lazyOn ? 0u : 1u;      // (per tick parity: a pipelined tile's pair half of tick t runs under the fused kernel of tick t + 1)
  if (kCull) {
    __shared__ uint32_t sVis[kTile / 64], sCand[kTile / 64];
    if (lane == 0) { sVis[wave] = visCount; sCand[wave] = candCount; }
    __syncthreads();
    if (threadIdx.x == 0) {
      d.blockVis[blockIdx.x] = sVis[0] + sVis[1] + sVis[2] + sVis[3];
      d.blockCand[blockIdx.x] = sCand[0] + sCand[1] + sCand[2] + sCand[3];
    }
  }
}

// (Round 3 built a second form of this kernel in which a workgroup's tile shares the walk through LDS -- link words, dirty
// words and the local matrices of the entities that rebuild, so that ancestors cost no memory round trips: three dependent
// trips per tile instead of eight, bit-identical results, all GPU tests green -- and it was no faster: 39.6 against 38.7 us at six
// and seven waves per SIMD, 48 at five (profiles/r03/ab_lds_kernel.log), 35.6 against 33.0 us once the bins' atomics were gone
// (ab_kernels_home.log).  The kernel's time was never its chain of round trips; it was the memory-side atomics of the binning
// (binEntityWave, "home slots").  The variant is in the history at commit 3b44da7.)
// ------------------------------------------------------------------------------------------
// Level kernel for entities deeper than kMaxChain (rare): one launch per level, parents final.
// nodeDirty = dirty || parent recomputed this tick (sc_ecs.cpp:184).
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kTile) void k_deep_level(const DeviceState d, const TickParams p,
                                                      const uint32_t* __restrict__ list, uint32_t count)
{
  const uint32_t t = blockIdx.x * kTile + threadIdx.x;
  if (t >= count) return;
  const uint32_t i = list[t];
  const uint32_t lk = ldU(d, kLINK, i);
  const uint32_t par = lk & kParentMask;
  bool nodeDirty = false;
  Aff M;
  if (p.flags & SC_TICK_XFORM) {
    nodeDirty = dirtyBit(d.dirty, i) || ((d.recomp[par >> 6] >> (par & 63u)) & 1ull);
    if (nodeDirty) {
      const Aff P = loadRows(d, par);
      const Aff L = loadLocal(d, i, lk);
      M = mulAff(P, L);
      storeRows(d, i, M);
      atomicOr((unsigned long long*)&d.recomp[i >> 6], 1ull << (i & 63u));
    }
  }
  if (!nodeDirty) M = loadRows(d, i);
  const bool hb = (lk & kHasBounds) != 0;
  BoundsCE b = {0, 0, 0, 0, 0, 0};
  if (hb) b = loadBounds(d, i);
  if (p.flags & SC_TICK_CULL) {
    bool visible = (lk & kHasMesh) != 0;
    if (visible && hb && !p.freeze && p.frustumValid) visible = sphereVisibleAt(M, b, (ConstF)d.frustum);
    if (visible) {
      atomicOr((unsigned long long*)&d.vis[i >> 6], 1ull << (i & 63u));
      atomicAdd(&d.blockVis[i / p.span], 1u);
    }
  }
  if (p.flags & SC_TICK_BROADPHASE) {
    if (hb) binEntitySingle(d, p, i, M, b);
  }
}

// ------------------------------------------------------------------------------------------
// K2: ordered (stable) compaction of the visibility bits into CullingState::visible (and ::culled),
// sc_world_partition.cpp:1273-1280, plus the end-of-tick dirty clear (sc_ecs.cpp:201).
// Same span partition as K1: workgroup b owns entities [b*span, (b+1)*span); its output offset is
// the sum of the preceding spans' counts; inside a tile each wave places its lanes by popcount.
// ------------------------------------------------------------------------------------------
// ---- frame producers, one entity per lane (a wave owns exactly the two dirty words of its 64 entities) ----
// SynthWorld dirty regime (ii): localPos.x += dx on every root, marked dirty.
__device__ __forceinline__ bool nudgePosition(const DeviceState& d, uint32_t i, uint32_t n, float dx)
{
  const bool in = i < n;
  const uint32_t lk = in ? d.link[i] : ((kUnreachable << kDepthShift) | 1u);
  const float x = in ? d.px[i] : 0.0f;
  const bool root = in && (lk & kParentMask) == kNoParent && linkDepth(lk) != kUnreachable;
  if (root) d.px[i] = x + dx;
  return root;
}
// a wave ORs the ballot of its 64 entities into their two dirty words
__device__ __forceinline__ void markDirtyWave(const DeviceState& d, uint32_t i, uint32_t n, bool moved)
{
  const uint32_t lane = threadIdx.x & 63u;
  const unsigned long long m = ballot64(moved);
  const uint32_t mine = lane < 32u ? (uint32_t)m : (uint32_t)(m >> 32);
  if (i < n && (lane & 31u) == 0u && mine) d.dirty[i >> 5] |= mine;
}
// Upstream movers (include/sc_tick.h "upstream movers"): straight-line advance inside the agent's
// sector, vehicles wrap, peds reflect.  pos + vel*dt is two roundings (no FMA), as the oracle's.
// ---- on-rails traffic (the step before the path) -------------------------------------------------------
// advanceAlongLane, src/engine/traffic/sc_traffic_lanes.cpp:291-352, with chooseNextSegment :137-156 inside: walk `distance`
// along the lane graph from (lane, s), at most eight segments; a dead end parks the agent on the end node.
struct LaneStep { bool ok; uint32_t lane; float s; float pos[3]; float dir[3]; };
__device__ __forceinline__ LaneStep advanceAlongLane(const LaneGraphDev& g, uint32_t lane, float s, float distance)
{
  LaneStep r; r.ok = false; r.lane = lane; r.s = s;
  r.pos[0] = r.pos[1] = r.pos[2] = 0.0f; r.dir[0] = r.dir[1] = r.dir[2] = 0.0f;
  if (lane == kInvalidLane || lane >= g.segments) return r;
  float remaining = distance; uint32_t current = lane; float currentS = s;
  for (uint32_t guard = 0; guard < 8u; ++guard) {
    const float4 A = g.segA[current], B = g.segB[current]; const uint4 C = g.segC[current];
    if (!C.y) return r;                                     // !seg.active
    const float len = A.w;
    if (len <= 1e-5f) return r;
    const float available = len - currentS;
    if (remaining <= available) {
      currentS += remaining;
      r.pos[0] = A.x + B.x * currentS; r.pos[1] = A.y + B.y * currentS; r.pos[2] = A.z + B.z * currentS;
      r.dir[0] = B.x; r.dir[1] = B.y; r.dir[2] = B.z;
      r.lane = current; r.s = currentS; r.ok = true;
      return r;
    }
    remaining -= available; currentS = 0.0f;
    // chooseNextSegment: the active connection of the end node whose direction agrees best (dot > -1)
    uint32_t best = kInvalidLane; float bestDot = -1.0f;
    const uint32_t c0 = g.nodeConnOff[C.x], c1 = g.nodeConnOff[C.x + 1u];
    for (uint32_t k = c0; k < c1; ++k) {
      const uint32_t segId = g.nodeConn[k];
      if (segId >= g.segments) continue;
      if (!g.segC[segId].y) continue;
      const float4 D = g.segB[segId];
      const float dot = B.x * D.x + B.y * D.y + B.z * D.z;
      if (dot > bestDot) { bestDot = dot; best = segId; }
    }
    if (best == kInvalidLane) {
      const float4 E = g.nodePos[C.x];
      r.pos[0] = E.x; r.pos[1] = E.y; r.pos[2] = E.z; r.dir[0] = B.x; r.dir[1] = B.y; r.dir[2] = B.z;
      r.lane = current; r.s = len; r.ok = true;
      return r;
    }
    current = best;
  }
  return r;
}

// TrafficLaneGraph::queryNearestLane (sc_traffic_lanes.cpp:240-279) for ONE position, by a whole wave (every lane must call):
// lanes take the segments 64 at a time; each keeps its closest (strict <, so the lowest index among equals within the lane),
// and one 64-bit min over (bits of the squared distance, index) elects the winner -- squared distances are >= 0, so their
// bit patterns order like the values, and equal distances go to the lower index as in the reference's ascending scan.
// Returns the lane id (kInvalidLane: no active segment) and its s, wave-uniform.
__device__ __forceinline__ uint32_t queryNearestLaneWave(const LaneGraphDev& g, float x, float y, float z, float& sOut)
{
  const uint32_t lane = threadIdx.x & 63u;
  float bestD = INFINITY, bestS = 0.0f; uint32_t best = kInvalidLane;
  for (uint32_t i = lane; i < g.segments; i += 64u) {
    const uint4 C = g.segC[i];
    const float4 A = g.segA[i], B = g.segB[i];
    if (!C.y || A.w <= 1e-5f) continue;                              // !seg.active || seg.length <= 1e-5f
    const float tx = x - A.x, ty = y - A.y, tz = z - A.z;
    const float proj = tx * B.x + ty * B.y + tz * B.z;
    const float mn = (proj < A.w) ? proj : A.w;                      // std::min(seg.length, proj)
    const float sv = (0.0f < mn) ? mn : 0.0f;                        // std::max(0.0f, ...)
    const float cx = A.x + B.x * sv, cy = A.y + B.y * sv, cz = A.z + B.z * sv;
    const float dx = x - cx, dy = y - cy, dz = z - cz;
    const float distSq = dx * dx + dy * dy + dz * dz;
    if (best == kInvalidLane || distSq < bestD) { bestD = distSq; best = i; bestS = sv; }
  }
  unsigned long long key = ((unsigned long long)__float_as_uint(bestD) << 32) | best;
  unsigned long long low = key;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { const unsigned long long other = __shfl_xor(low, o, 64); low = other < low ? other : low; }
  const unsigned long long winners = ballot64(best != kInvalidLane && key == low);
  sOut = 0.0f;
  if (!winners) return kInvalidLane;
  const int win = __ffsll((long long)winners) - 1;
  sOut = __shfl(bestS, win, 64);
  return (uint32_t)__shfl((int)best, win, 64);
}

// TrafficAISystem for one agent of the OnRails tier: the per-agent preamble src/engine/traffic/sc_traffic_ai.cpp:274-299 (valid
// active lane, look-ahead point, the 1e-4 early out, desired speed), the obstacle brake of :300-345 as the agent's front ray
// left it in aBrake[] (k_agent_front_rays; 0 without sensors, as the reference without a PhysicsWorld), and the on-rails branch
// :434-460.  (lane, s0) are the agent's lane state after the re-acquisition of :264-272, which the caller performs.
// Returns true when the transform was written (Transform::dirty).
// `smooth` = 1 - exp(-2.5 dt) from the host (smoothExp :58-62); the yaw's sin / cos come from the segment table.
__device__ __forceinline__ bool trafficAgentStep(const DeviceState& d, uint32_t i, uint32_t lane, float s0, float dt, float smooth, float mult)
{
  if (d.aMode[i] != kTierOnRails) return false;               // Physics / Kinematic tiers: moved by the physics sync, not here
  const LaneGraphDev& g = d.lanes;
  if (lane == kInvalidLane || lane >= g.segments) return false;
  if (!g.segC[lane].y) return false;
  const LaneStep look = advanceAlongLane(g, lane, s0, d.aLook[i]);   // getLookAheadPoint, sc_traffic_lanes.cpp:281-289
  if (!look.ok) return false;
  const float tx = look.pos[0] - d.px[i], tz = look.pos[2] - d.pz[i];
  if (sqrtf(tx * tx + 0.0f * 0.0f + tz * tz) < 1e-4f) return false;
  float desiredSpeed = g.segB[lane].w * mult;
  desiredSpeed = (0.0f < desiredSpeed) ? desiredSpeed : 0.0f;         // std::max(0.0f, desiredSpeed)
  const float obstacleBrake = d.aBrake ? d.aBrake[i] : 0.0f;          // :300-345
  const float desired = desiredSpeed * (1.0f - obstacleBrake);        // :436
  const float cur = d.aSpeed[i];
  const float speed = cur + (desired - cur) * smooth;
  d.aSpeed[i] = speed;
  const LaneStep st = advanceAlongLane(g, lane, s0, speed * dt);
  if (!st.ok) return false;
  d.aLane[i] = st.lane; d.aS[i] = st.s;
  d.px[i] = st.pos[0]; d.pz[i] = st.pos[2];                    // localPos.y keeps its value (:447)
  const uint4 C = g.segC[st.lane];
  d.rsx[i] = 0.0f; d.rcx[i] = 1.0f; d.rsy[i] = __uint_as_float(C.z); d.rcy[i] = __uint_as_float(C.w); d.rsz[i] = 0.0f; d.rcz[i] = 1.0f;
  return true;
}

// One entity per lane; every lane of the wave must call (the lane re-acquisition is a whole-wave step).
__device__ __forceinline__ bool moverPosition(const DeviceState& d, uint32_t i, uint32_t n, float dt, float smooth, float mult)
{
  const bool in = i < n;
  // everything is requested at once (one round trip); non-movers just drop what they fetched
  const uint32_t kind = in ? d.moverKind[i] : 0u;
  float vx = 0, vz = 0, lox = 0, loz = 0, hix = 0, hiz = 0, x = 0, z = 0;
  if (in) { vx = d.mvx[i]; vz = d.mvz[i]; lox = d.mlox[i]; loz = d.mloz[i]; hix = d.mhix[i]; hiz = d.mhiz[i]; x = d.px[i]; z = d.pz[i]; }
  // An agent without a lane takes the nearest active one first, whatever its tier (sc_traffic_ai.cpp:264-272): rare -- a lane
  // id is lost when its sector's lanes go (sc_traffic_lanes.cpp:227-237) -- so the wave serves such lanes one after the other.
  uint32_t aLane = kInvalidLane; float aS = 0.0f;
  if (d.aLane) {
    const bool agent = kind == kMoverTraffic;
    if (agent) { aLane = d.aLane[i]; aS = d.aS[i]; }
    unsigned long long lost = ballot64(agent && aLane == kInvalidLane);
    while (lost) {
      const int src = __ffsll((long long)lost) - 1;
      lost &= lost - 1ull;
      const uint32_t who = (uint32_t)__shfl((int)i, src, 64);
      float qs;
      const uint32_t q = queryNearestLaneWave(d.lanes, __shfl(x, src, 64), d.py[who], __shfl(z, src, 64), qs);
      if ((int)(threadIdx.x & 63u) == src && q != kInvalidLane) { aLane = q; aS = qs; d.aLane[i] = q; d.aS[i] = qs; }
    }
  }
  if (kind == kMoverTraffic) return trafficAgentStep(d, i, aLane, aS, dt, smooth, mult);      // (behind the loads: they must not wait for `kind`)
  if (kind) {
    x = x + vx * dt; z = z + vz * dt;
    if (kind == 1u) {
      if (x >= hix) x = lox + (x - hix); else if (x < lox) x = hix - (lox - x);
      if (z >= hiz) z = loz + (z - hiz); else if (z < loz) z = hiz - (loz - z);
    } else {
      if (x > hix) { x = hix - (x - hix); vx = -vx; } else if (x < lox) { x = lox + (lox - x); vx = -vx; }
      if (z > hiz) { z = hiz - (z - hiz); vz = -vz; } else if (z < loz) { z = loz + (loz - z); vz = -vz; }
      d.mvx[i] = vx; d.mvz[i] = vz;
    }
    d.px[i] = x; d.pz[i] = z;
  }
  return kind != 0u;
}
__device__ __forceinline__ bool producePosition(const DeviceState& d, const TickParams& p, uint32_t i)
{
  return p.producerKind == 1u ? nudgePosition(d, i, p.n, p.producerParam) : moverPosition(d, i, p.n, p.producerParam, p.trafficSmooth, p.trafficMult);
}

__device__ __forceinline__ uint32_t blockSum(uint32_t v, uint32_t* scratch)
{
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) scratch[wave] = v;
  __syncthreads();
  return scratch[0] + scratch[1] + scratch[2] + scratch[3];
}

// `group`: how many of the fused kernel's spans one compaction workgroup takes (their counts are summed here; fewer,
// longer workgroups keep the whole end-of-tick kernel resident at once).
//
// The role is a handful of small dependent steps, so what it costs is memory round trips, not bytes.  Everything it reads is
// therefore requested up front, before anything is stored (a store to a stream keeps the compiler from hoisting later loads of
// that stream above it): the dirty / unreachable words, the producer's inputs for several tiles at once, the predecessors'
// counts, and ONE visibility word per thread for the whole width (a word = one wave-tile of the fused kernel).  Then: block
// scan of the words' popcounts -> (word, offset) pairs in LDS -> every thread scatters its own entities.  `words` is LDS
// scratch of kCompactLdsWords dwords.
constexpr uint32_t kCompactWordsMax = kTile;                  // visibility words (64 entities each) one workgroup handles the fast way
constexpr uint32_t kCompactLdsWords = kCompactWordsMax * 6u;  // per word: vis (2 dwords), culled (2), offsets (2)

// Draw emission inside the compaction (TickParams::emitMode): the entity placed at position `off` of the ordered visible list is
// draw item `off` -- DrawItem{entity, mesh, material, worldMatrix}, the first `budget` of them (sc_world_partition.cpp:1306-1329)
// -- and, with the frame read-back on, entry `off` of the block's copy of the list.  Same items as k_emit_draws /
// k_emit_draws_staged write from the finished list; no launch of their own, no second pass over the list.
__device__ __forceinline__ void emitVisible(const DeviceState& d, const TickParams& p, uint32_t off, uint32_t j)
{
  float4* items = reinterpret_cast<float4*>(p.emitTarget);
  if (p.emitMode == 2u) {
    if (off < p.emitMaxVisible) p.emitTarget[kFrameHeaderWords + off] = j;
    items = reinterpret_cast<float4*>(p.emitTarget + kFrameHeaderWords + p.emitMaxVisible);      // (emitMaxVisible is a multiple of 4: 16-byte aligned)
  }
  if (p.emitBudget == 0u || off < p.emitBudget) {
    const float4 a = ldRow(d, 0, j), b = ldRow(d, 1, j), c = ldRow(d, 2, j);
    float4* o = items + 5u * (size_t)off;
    o[0] = make_float4(__uint_as_float(j), __uint_as_float(d.meshId[j]), __uint_as_float(d.materialId[j]), 0.0f);
    o[1] = make_float4(a.x, b.x, c.x, 0.0f);     // column 0
    o[2] = make_float4(a.y, b.y, c.y, 0.0f);
    o[3] = make_float4(a.z, b.z, c.z, 0.0f);
    o[4] = make_float4(a.w, b.w, c.w, 1.0f);     // translation column
  }
}

// (kEmit: an instance of its own -- with the emission code merely present the role ran 1.2 us longer on ticks that emit nothing)
template <bool kEmit>
__device__ __forceinline__ void compactBody(const DeviceState& d, const TickParams& p, uint32_t bid, uint32_t nblocks, uint32_t group,
                                            uint32_t* scratch, uint32_t* moved, uint32_t* words)
{
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t wave = threadIdx.x >> 6;
  const uint32_t width = p.span * group;
  const uint32_t begin = bid * width;
  const uint32_t end = (begin + width < p.n) ? begin + width : p.n;
  const bool doCull = (p.flags & SC_TICK_CULL) != 0;
  const bool doCulled = (p.flags & SC_TICK_CULLED_LIST) != 0;

  // ---- requests first
  const uint32_t wBegin = begin >> 5, wEnd = (end + 31u) >> 5;
  uint32_t dirtyKeep = 0; const uint32_t dirtyWord = wBegin + threadIdx.x;
  const bool clearDirty = (p.flags & SC_TICK_XFORM) && dirtyWord < wEnd;
  if (clearDirty) dirtyKeep = d.dirty[dirtyWord] & d.unreach[dirtyWord];

  const uint32_t vBegin = begin >> 6, vEnd = (end + 63u) >> 6, vWords = vEnd - vBegin;
  const bool fast = doCull && vWords <= kCompactWordsMax;
  unsigned long long myVis = 0ull, myCul = 0ull;
  if (fast && threadIdx.x < vWords) {
    myVis = d.vis[vBegin + threadIdx.x];
    if (doCulled) myCul = d.cand[vBegin + threadIdx.x] & ~myVis;
  }
  uint32_t pv = 0, pc = 0;
  if (doCull) for (uint32_t j = threadIdx.x; j < bid * group; j += kTile) { pv += d.blockVis[j]; pc += d.blockCand[j]; }

  // SC_TICK_PRODUCE_NEXT: this tick is over for the span, so the NEXT frame's producer runs here instead of as a
  // launch of its own (nothing else in this kernel reads positions); the "moved" ballots wait in LDS and are OR-ed into
  // the cleared dirty words below.
  const bool produce = (p.flags & SC_TICK_PRODUCE_NEXT) != 0;
  const bool produceEarly = produce && (wEnd - wBegin) <= kMaxSpanWords;
  if (produceEarly) {
    if (p.producerKind == 1u) {
      // root nudge: link word and x of four tiles are requested together, then written
      constexpr uint32_t kBatch = 6;                      // (a compaction workgroup of two 768-entity spans: all of it in one round trip)
      for (uint32_t base = begin; base < end; base += kBatch * kTile) {
        uint32_t lk[kBatch]; float x[kBatch];
#pragma unroll
        for (uint32_t u = 0; u < kBatch; ++u) {
          const uint32_t i = base + u * kTile + threadIdx.x;
          const bool in = i < end;
          lk[u] = in ? d.link[i] : ((kUnreachable << kDepthShift) | 1u);
          x[u] = in ? d.px[i] : 0.0f;
        }
#pragma unroll
        for (uint32_t u = 0; u < kBatch; ++u) {
          const uint32_t tileBase = base + u * kTile, i = tileBase + threadIdx.x;
          const bool root = i < end && (lk[u] & kParentMask) == kNoParent && linkDepth(lk[u]) != kUnreachable;
          if (root) d.px[i] = x[u] + p.producerParam;
          const unsigned long long m = ballot64(root);
          if (lane == 0 && tileBase < end) { moved[((tileBase >> 5) - wBegin) + 2u * wave] = (uint32_t)m; moved[((tileBase >> 5) - wBegin) + 2u * wave + 1u] = (uint32_t)(m >> 32); }
        }
      }
    } else {
      for (uint32_t base = begin; base < end; base += kTile) {
        const unsigned long long m = ballot64(producePosition(d, p, base + threadIdx.x));
        if (lane == 0) { moved[((base >> 5) - wBegin) + 2u * wave] = (uint32_t)m; moved[((base >> 5) - wBegin) + 2u * wave + 1u] = (uint32_t)(m >> 32); }
      }
    }
    __syncthreads();
  }

  if (doCull) {
    uint32_t visBase = blockSum(pv, scratch);
    uint32_t culBase = blockSum(pc, scratch) - visBase;
    if (bid == nblocks - 1 && threadIdx.x == 0) {
      uint32_t tv = visBase, tc = culBase + visBase;
      const uint32_t spans = (p.n + p.span - 1u) / p.span;
      for (uint32_t j = bid * group; j < spans; ++j) { tv += d.blockVis[j]; tc += d.blockCand[j]; }
      d.counters[0] = tv;            // CullingStats::visible
      d.counters[1] = tc - tv;       // CullingStats::culled
      d.counters[6] = tc;            // renderablesTotal
      if (kEmit && p.emitMode) {     // RenderPrepStreaming's counts, and the read-back block's header (k_emit_draws_staged's)
        const uint32_t emitted = (p.emitBudget > 0u && tv > p.emitBudget) ? p.emitBudget : tv;
        d.counters[4] = emitted; d.counters[5] = tv - emitted;
        if (p.emitMode == 2u) {
          const uint32_t nv = tv < p.emitMaxVisible ? tv : p.emitMaxVisible;
          const uint32_t h[kFrameHeaderWords] = { tv, tc - tv, tc, emitted, tv - emitted, 0u, p.emitTickLo, p.emitTickHi, nv, emitted, 0, 0, 0, 0, 0, 0 };
          for (uint32_t k = 0; k < kFrameHeaderWords; ++k) p.emitTarget[k] = h[k];
        }
      }
    }
    const unsigned long long below = (1ull << lane) - 1ull;
    if (fast) {
      // exclusive scan of the words' popcounts over the workgroup (one word per thread)
      uint32_t cv = (uint32_t)__popcll(myVis), cc = (uint32_t)__popcll(myCul);
      uint32_t iv = cv, ic = cc;
#pragma unroll
      for (uint32_t o = 1; o < 64u; o <<= 1) {
        const uint32_t uv = __shfl_up(iv, o, 64), uc = __shfl_up(ic, o, 64);
        if (lane >= o) { iv += uv; ic += uc; }
      }
      __syncthreads();                                   // (blockSum's readers are done with `scratch`)
      if (lane == 63u) scratch[wave] = iv | (ic << 16);    // a wave holds at most 64 x 64 = 4096 of either: 16 bits each
      __syncthreads();
      uint32_t bv = visBase, bc = culBase;
      for (uint32_t k = 0; k < wave; ++k) { bv += scratch[k] & 0xFFFFu; bc += scratch[k] >> 16; }
      unsigned long long* sVis = reinterpret_cast<unsigned long long*>(words);
      unsigned long long* sCul = sVis + kCompactWordsMax;
      uint32_t* sOffV = words + 4u * kCompactWordsMax;
      uint32_t* sOffC = sOffV + kCompactWordsMax;
      if (threadIdx.x < vWords) {
        sVis[threadIdx.x] = myVis; sOffV[threadIdx.x] = bv + iv - cv;
        if (doCulled) { sCul[threadIdx.x] = myCul; sOffC[threadIdx.x] = bc + ic - cc; }
      }
      __syncthreads();
      for (uint32_t base = begin; base < end; base += kTile) {
        const uint32_t i = base + threadIdx.x;
        const uint32_t wi = (i >> 6) - vBegin;           // this thread's wave-tile
        if (wi < vWords) {
          const unsigned long long m = sVis[wi];
          if ((m >> lane) & 1ull) {
            const uint32_t off = sOffV[wi] + (uint32_t)__popcll(m & below);
            d.visibleIdx[off] = i;
            if (kEmit) emitVisible(d, p, off, i);
          }
          if (doCulled) {
            const unsigned long long c = sCul[wi];
            if ((c >> lane) & 1ull) d.culledIdx[sOffC[wi] + (uint32_t)__popcll(c & below)] = i;
          }
        }
      }
    } else {
      const uint32_t nWords = (p.n + 63u) >> 6;
      for (uint32_t base = begin; base < end; base += kTile) {
        const uint32_t w0 = base >> 6;
        unsigned long long m[4], c[4];
#pragma unroll
        for (uint32_t k = 0; k < 4; ++k) {
          const bool ok = (w0 + k) < nWords;
          m[k] = ok ? d.vis[w0 + k] : 0ull;
          c[k] = (doCulled && ok) ? (d.cand[w0 + k] & ~m[k]) : 0ull;
        }
        uint32_t off = visBase, coff = culBase;
#pragma unroll
        for (uint32_t k = 0; k < 4; ++k) {
          if (k < wave) { off += (uint32_t)__popcll(m[k]); coff += (uint32_t)__popcll(c[k]); }
        }
        const unsigned long long mine = (wave == 0) ? m[0] : (wave == 1) ? m[1] : (wave == 2) ? m[2] : m[3];
        const uint32_t i = base + threadIdx.x;
        if ((mine >> lane) & 1ull) {
          const uint32_t at = off + (uint32_t)__popcll(mine & below);
          d.visibleIdx[at] = i;
          if (kEmit) emitVisible(d, p, at, i);
        }
        if (doCulled) {
          const unsigned long long cm = (wave == 0) ? c[0] : (wave == 1) ? c[1] : (wave == 2) ? c[2] : c[3];
          if ((cm >> lane) & 1ull) d.culledIdx[coff + (uint32_t)__popcll(cm & below)] = i;
        }
#pragma unroll
        for (uint32_t k = 0; k < 4; ++k) { visBase += (uint32_t)__popcll(m[k]); culBase += (uint32_t)__popcll(c[k]); }
      }
    }
  }

  // Transform::dirty = false for every visited entity; entities in a cycle keep theirs (sc_ecs.cpp:201)
  if (clearDirty) d.dirty[dirtyWord] = dirtyKeep | (produceEarly ? moved[dirtyWord - wBegin] : 0u);
  if (p.flags & SC_TICK_XFORM) {
    for (uint32_t w = wBegin + kTile + threadIdx.x; w < wEnd; w += kTile) d.dirty[w] &= d.unreach[w];   // spans wider than 8192 entities
  }
  // spans too wide for the LDS ballots: the producer runs behind the dirty clear, one more pass over the span
  if (produce && !produceEarly) {
    __syncthreads();
    for (uint32_t base = begin; base < end; base += kTile) {
      const uint32_t i = base + threadIdx.x;
      markDirtyWave(d, i, p.n, producePosition(d, p, i));
    }
  }
}

__global__ __launch_bounds__(kTile) void k_compact(const DeviceState d, const TickParams p, uint32_t group)
{
  __shared__ uint32_t scratch[kTile / 64];
  __shared__ uint32_t moved[kMaxSpanWords];
  __shared__ uint32_t words[kCompactLdsWords];
  compactBody<false>(d, p, blockIdx.x, gridDim.x, group, scratch, moved, words);
}

// ------------------------------------------------------------------------------------------
// K3: pair search.  One wave per sector bin (grid-stride): the bin's <= 64 records become an LDS
// tile (lane i stages record i), the n(n-1)/2 record pairs are spread over the 64 lanes, each lane
// reads its two records from LDS and applies: not the same box, closed-interval overlap on three
// axes, Bullet's group/mask filter, and "this sector holds the low corner of the intersection".
// Big boxes are tested against the bin's primary records here too, and against each other.
// The bin counter is zeroed by the wave that consumed it (self-cleaning for the next tick).
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ bool boxesOverlap(const float4& amin, const float4& amax, const float4& bmin, const float4& bmax)
{
  return amin.x <= bmax.x && bmin.x <= amax.x && amin.y <= bmax.y && bmin.y <= amax.y && amin.z <= bmax.z && bmin.z <= amax.z;
}
__device__ __forceinline__ bool filterPass(uint32_t la, uint32_t lb)
{
  // (a.group & b.mask) && (b.group & a.mask); layers = group | mask << 16
  return ((la & 0xFFFFu) & (lb >> 16)) != 0u && ((lb & 0xFFFFu) & (la >> 16)) != 0u;
}

// Per-wave pair sink: hits go to an LDS buffer (no atomics); a flush reserves a range in the segment of
// the workgroup's shard with ONE atomic and writes it out coalesced.  Every lane of the wave must call these.
struct PairSink { uint2* buf; uint32_t count; uint32_t shard; };

__device__ __forceinline__ void sinkFlush(const DeviceState& d, const TickParams& p, PairSink& k)
{
  if (!k.count) return;
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t shardCap = p.maxPairs / kPairShards;
  uint32_t done = 0;
  // reserve in the current shard; what does not fit is handed back and tried in the next shard, so the
  // list only truncates when every segment is full.  Shards rotate per flush: load stays even when a
  // few workgroups produce most of the pairs.
  for (uint32_t attempt = 0; attempt < kPairShards && done < k.count; ++attempt) {
    const uint32_t shard = (k.shard + attempt) % kPairShards;
    uint32_t* ctr = &d.pairShardCount[(p.parity * kPairShards + shard) * kShardStride];
    const uint32_t want = k.count - done;
    uint32_t base = 0;
    if (lane == 0) {
      base = atomicAdd(ctr, want);
      const uint32_t room = base < shardCap ? shardCap - base : 0u;
      if (room < want) atomicSub(ctr, want - room);               // give back what this segment cannot hold
    }
    base = __shfl(base, 0, 64);
    const uint32_t room = base < shardCap ? shardCap - base : 0u;
    const uint32_t take = room < want ? room : want;
    __builtin_amdgcn_wave_barrier();
    for (uint32_t q = lane; q < take; q += 64u) d.pairs[(size_t)shard * shardCap + base + q] = k.buf[done + q];
    done += take;
  }
  if (done < k.count && lane == 0) atomicAdd(&d.counters[kCtrPar + 8u * p.parity + kCtrPairs], k.count - done);   // dropped: every segment full
  __builtin_amdgcn_wave_barrier();
  k.shard = (k.shard + 1u) % kPairShards;
  k.count = 0;
}
__device__ __forceinline__ void sinkPush(const DeviceState& d, const TickParams& p, PairSink& k, bool hit, uint32_t ia, uint32_t ib)
{
  const unsigned long long m = ballot64(hit);
  if (!m) return;
  const uint32_t lane = threadIdx.x & 63u;
  if (hit) k.buf[k.count + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = make_uint2(ia < ib ? ia : ib, ia < ib ? ib : ia);
  k.count += (uint32_t)__popcll(m);
  if (k.count > kWavePairBuf - 64u) sinkFlush(d, p, k);        // keep room for a full wave of hits
}

// The state of the OTHER tick parity -- counter set, pair shard counters -- back to zero.
__device__ __forceinline__ void resetOtherParity(const DeviceState& d, const TickParams& p, uint32_t bid, uint32_t nblocks)
{
  if (bid == 0 && threadIdx.x < 8) d.counters[kCtrPar + 8u * (p.parity ^ 1u) + threadIdx.x] = 0u;
  if (bid == 0 && threadIdx.x < kPairShards) d.pairShardCount[((p.parity ^ 1u) * kPairShards + threadIdx.x) * kShardStride] = 0u;
  if (bid == 0 && threadIdx.x == 0) d.lazyCtl[1u + kMaxParity + (p.parity ^ 1u)] = 0u;
  (void)nblocks;
}

// Pipelined tiles: a parity's counter set and pair shard counters are cleared on the TICK stream, by the end-of-tick
// kernel of the tick BEFORE the one that fills them again (TickParams::resetParity; the host has made sure that parity's
// pair half is over).  Until then they hold that pair half's results for the host to read -- no snapshot, no kernel behind
// the pair search.  (Clearing in the pair kernel's last workgroup would need a ticket per workgroup: a thousand device-scope
// atomics on one word; a kernel of its own behind the pair search cost 4 us + a dispatch gap on the pairs stream.)
__device__ __forceinline__ void resetParity(const DeviceState& d, uint32_t q)
{
  if (threadIdx.x < 8) d.counters[kCtrPar + 8u * q + threadIdx.x] = 0u;
  if (threadIdx.x < kPairShards) d.pairShardCount[(q * kPairShards + threadIdx.x) * kShardStride] = 0u;
  if (threadIdx.x == 0) d.lazyCtl[1u + kMaxParity + q] = 0u;
}
__global__ __launch_bounds__(kTile) void k_reset_parity(const DeviceState d, uint32_t q)
{
  if (blockIdx.x == 0) resetParity(d, q);
}

// Does this tile own sector (gx, gz) of its bin grid (coordinates may lie outside the grid)?  A sector belongs to
// the tile nearest to it, so outside the core [1, binS-2] it is ours only on sides where no tile exists.
__device__ __forceinline__ bool ownsSector(const TickParams& p, float gx, float gz)
{
  const int dx = gx < 1.0f ? -1 : (gx > (float)(p.binSX - 2u) ? 1 : 0);
  const int dz = gz < 1.0f ? -1 : (gz > (float)(p.binSZ - 2u) ? 1 : 0);
  return !((dx != 0 && hasNb(p, dx, 0)) || (dz != 0 && hasNb(p, 0, dz)));
}

constexpr uint32_t kPairTabSize = kBinCap * (kBinCap - 1) / 2;
constexpr uint32_t kFineThreshold = 24;      // bins with more records than this use the 4x4 cell grid
constexpr uint32_t kCellWords = 16u * (1u + kOvfPerSector / 64u);     // 4x4 cell masks for the bin tile and every overflow tile of a sector

// crowded sectors (more records than the bin holds) wait in a queue of the whole launch; a full queue leaves a sector to the wave that met it

constexpr uint32_t kTileSlots = kBinCap + 1u; // a wave's LDS tile: the bin's 64 records + one slot that always holds a null record (fast sectors)
constexpr uint32_t kCandCap = 512;           // fast sectors: (slot, slot) candidates a wave collects before it resolves them (a multiple of 64)
#ifndef SC_BROADCAST_MAX
#define SC_BROADCAST_MAX 4
#endif
constexpr uint32_t kBroadcastMax = SC_BROADCAST_MAX;   // fast sectors: up to this many sweepers are broadcast from registers, nothing staged in LDS

// pair predicate shared by both search paths: group/mask filter, closed-interval overlap, and "this sector
// holds the low corner of the intersection" (so the pair is reported from exactly one bin)
__device__ __forceinline__ bool pairHit(const TickParams& p, const float4& amin, const float4& amax, const float4& bmin, const float4& bmax,
                                        float secX, float secZ, uint32_t& ia, uint32_t& ib)
{
  if (!filterPass(__float_as_uint(amin.w), __float_as_uint(bmin.w))) return false;
  if (!boxesOverlap(amin, amax, bmin, bmax)) return false;
  ia = __float_as_uint(amax.w) & ~kPrimary; ib = __float_as_uint(bmax.w) & ~kPrimary;
  const float lx = amin.x > bmin.x ? amin.x : bmin.x, lz = amin.z > bmin.z ? amin.z : bmin.z;
  return ia != ib && (floorf(lx * p.invSector) - p.binOx) == secX && (floorf(lz * p.invSector) - p.binOz) == secZ;
}

// Lazy records: the record a remembered slot of bin `sector` WOULD hold this tick, rebuilt from its owner's world matrix --
// binEntityWave's arithmetic and its rules (the slot carries a record only while the owner's box keeps the primary sector
// the slot was reserved from and still has this copy; a null record otherwise).  The slot itself holds whatever its owner
// wrote last -- always under the owner's id.
//
// A bin that had to be rebuilt once is likely to be needed again (a vehicle that drove into a street of props stays a while):
// its owners are told to write it from the next tick on (kSlotAlways in their homeB byte for this copy; kHomeHot in the bin's
// homeCount), until the next learn tick sorts the bins afresh.
__device__ __forceinline__ void rebuildHomeRecord(const DeviceState& d, const TickParams& p, uint32_t sector, float4& rmin, float4& rmax)
{
  const uint32_t i = __float_as_uint(rmax.w) & 0x00FFFFFFu;
  float4 lo, hi; nullRecord(lo, hi, i | p.rankBits);
  if (i < p.n && (ldU(d, kLINK, i) & kHasBounds)) {
    const uint32_t hA = d.homeA[i];
    {
      const uint32_t offH = sector - hA;                  // which of the owner's copies this slot is
      const uint32_t k = offH == 0u ? 0u : (offH == 1u ? 1u : (offH == p.binSX ? 2u : 3u));
      atomicOr(&d.homeB[i], kSlotAlways << (8u * k));
    }
    const Aff M = loadRows(d, i);
    const BoundsCE b = loadBounds(d, i);
    float mn[3], mx[3];
    worldAabb(M, b, mn, mx);
    const BinPlan plan = planBins(p, mn, mx);
    if (plan.collide && !plan.big) {
      const uint32_t s0 = (uint32_t)plan.z0 * p.binSX + (uint32_t)plan.x0;
      const bool c1 = plan.nx > 1u, c2 = plan.nz > 1u;
      const uint32_t off = sector - s0;
      const bool here = hA == s0 && (off == 0u || (c1 && off == 1u) || (c2 && off == p.binSX) || (c1 && c2 && off == p.binSX + 1u));
      if (here) {
        lo = make_float4(mn[0], mn[1], mn[2], __uint_as_float(ldU(d, kLAYERS, i)));
        hi = make_float4(mx[0], mx[1], mx[2], __uint_as_float(i | p.rankBits | (off == 0u ? kPrimary : 0u)));
      }
    }
  }
  rmin = lo; rmax = hi;
}

#ifdef SC_DIAG_WAVETIME
// timing experiment (never in the product build): how long the sweep of every pair-role wave takes, against the slowest one
__device__ unsigned long long g_waveDiag[40];
__device__ uint32_t g_waveRows[8192 * 8];      // per wave: duration, fast sectors, sum S*n of them, general sectors, sum n of them, rounds, t0 - (min t0) low bits, wave id
#endif
// (kVocab: the vocabulary form of the lazy records, TickParams::lazy 2 -- an instance of its own, only ever the pairs-stream
//  kernel's: with its branches merely present the in-order end-of-tick kernel ran 0.6 us (config 3) to 2.4 us (config 5) longer)
template <bool kVocab>
__device__ __forceinline__ void pairsBody(const DeviceState& d, const TickParams& p, uint32_t bid, uint32_t nblocks,
                                          float4 (*tile)[2 * kTileSlots], uint16_t* pairTab, uint2 (*pairBuf)[kWavePairBuf],
                                          unsigned long long (*cellMembers)[kCellWords])
{
#ifdef SC_DIAG_WAVETIME
  const unsigned long long diagT0 = wall_clock64();
  uint32_t diagFast = 0, diagFastWork = 0, diagGen = 0, diagGenN = 0, diagRounds = 0;
#endif
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t wave = threadIdx.x >> 6;
  const uint32_t waveGlobal = bid * (kTile / 64) + wave;
  const uint32_t totalWaves = nblocks * (kTile / 64);
  const uint32_t sectors = p.binSX * p.binSZ;
  const uint32_t ctr = kCtrPar + 8u * p.parity;
  const uint32_t nbig = min(d.counters[ctr + kCtrBig], p.bigCap);      // (bounded whatever the counter holds)
  const uint32_t novf = min(d.counters[ctr + kCtrSpill], p.ovfCap);     // sector overflow list: own boxes and neighbours' border records
  // lazy records: did this tick's fused kernel write every remembered slot?  If not, the slots of a bin whose own records admit
  // no pair hold old records, and a wave that needs such a bin rebuilds them (rebuildHomeRecord)
  // (p.lazy: 0 = the fused kernel wrote everything; 1 = it says what it did -- big boxes make it write everything; 2 = it left
  //  unwritten what nothing in the world can meet, see k_home_flags)
  //  (the flag is read unconditionally, up front with the other scalars: behind a test of p.lazy it cost every workgroup a stall)
  const bool freshAll = kVocab ? false : (!p.homeReset || d.lazyCtl[1u + p.parity] != 0u);
  // (Tried and dropped, profiles/r04/ab_quiet_tick_skip_lost.log.  (1) Skipping the sweep on a QUIET tick -- no bin whose reserved
  //  records can form a pair, nothing reserved since the tick began: an all-static city -- behind two flags read here: config 3's end of
  //  tick 9.36 us against 9.2, its waves do one round each and the flags cost what the round's own loads cost; the others within noise.
  //  (2) The cast count folded into the homeCount word, one load fewer per sector: config3dyn 29.1 us against 27.1, config 5 43.2
  //  against 40.6 in two sessions -- the word is also WRITTEN by the sweep (the hot mark), and the build that read one word for both
  //  lost what the fast sectors had won; not understood further, the separate homeCast word stays.)
  constexpr bool vocabMode = kVocab;                   // (the launcher picks the instance by p.lazy)
  if (bid == 0u && threadIdx.x == 0u) d.lazyCtl[0] = nbig;             // (what the next fused kernel goes by)
  float4* T = tile[wave];
  PairSink sink = { pairBuf[wave], 0u, bid % kPairShards };

  // the counters of the wave's FIRST 64 sectors are requested here, ahead of the workgroup's prologue (pair table, barrier): their round
  // trip runs behind it (most waves do one round; the loop below starts from these values)
  const uint32_t runLen = p.pairRun, runsPerRound = 64u / runLen;
  const uint32_t laneRun = lane / runLen, laneOff = (lane - laneRun * runLen + waveGlobal) % runLen;      // once per wave
  const bool laneInRound = laneRun < runsPerRound;                                                           // (64 need not be a multiple of the run)
  const uint32_t sector0 = laneInRound ? (laneRun * totalWaves + waveGlobal) * runLen + laneOff : 0xFFFFFFFFu;
  uint32_t preCount = 0, preLay = 0, preHc = 0, preHl = 0, preCast = 0;
  if (sector0 < sectors) {
    preCount = d.binCount[sector0]; preLay = d.binLayers[sector0];
    if (p.homeReset) { preHc = d.homeCount[sector0]; preHl = d.homeLayers[sector0]; }
    if (p.fastPairs) preCast = d.homeCast[sector0];
  }

  // triangular pair table (row i starts at i(i-1)/2; entry = i << 8 | j): the same 4 KB for every workgroup of every launch, so it is
  // COPIED from a constant block the context made once (DeviceState::pairConst) -- one 16-byte load and store per thread -- instead of
  // being built by every workgroup (16 rows per wave: ~130 wave-instructions of prologue, which is what a world with little to search pays for)
  static_assert(kPairTabSize * sizeof(uint16_t) == 252u * 16u, "the copy below moves the table as 252 x 16 bytes");
  if (threadIdx.x < 252u) reinterpret_cast<uint4*>(pairTab)[threadIdx.x] = reinterpret_cast<const uint4*>(d.pairConst)[threadIdx.x];

  // broadcast path: with D records to cast, lane l plays (record l / G, partner phase l % G), G = 64 / D; the division by
  // the wave-uniform G is a multiplication by ceil(2^16 / G) (exact for l < 64)
  __shared__ uint32_t castTab[kBinCap + 1u];
  // crowded sectors (more records than the bin holds) are not searched by the wave that meets them: they go to a queue of
  // the whole launch, and every workgroup, done with its sweep, takes sectors from it, four waves to a sector (see
  // "crowded sectors" below)
  __shared__ uint32_t crowdGathered, crowdItem, crowdLayers;
  if (threadIdx.x == 0) { crowdGathered = 0u; crowdLayers = 0u; }
  if (threadIdx.x <= kBinCap) castTab[threadIdx.x] = d.pairConst[kPairConstCast + threadIdx.x];      // (G | ceil(2^16 / G) << 8 for G = 64 / D: the same block)

  // next tick's counter set starts clean (pipelined tiles do this in the end-of-tick kernel on the tick stream instead:
  // there the next tick's fused kernel may already be filling it while this pair search runs)
  if (!(p.flags & kFlagDeferredReset)) resetOtherParity(d, p, bid, nblocks);
  __syncthreads();

  // a wave visits sectors waveGlobal, +totalWaves, ...; their counts are fetched 64 at a time (lane k
  // holds the k-th) and zeroed at once: this wave is the only consumer of those bins this tick
  // up to 64 sectors per wave and round, as runs of R = pairRun consecutive sectors (pairGeometry()): a wave's
  // counter loads and stores touch 64/R segments instead of 64 separate cache lines.  The runs of one wave are totalWaves
  // runs apart, so a dense district of the world is still spread over many waves; inside a run the lanes are rotated by the
  // wave's index, so waves that start together do not all read bins at the same offset of a 2^k-byte stride.  The launcher
  // sizes the grid so that every wave gets the same number of runs (within one): the slowest wave ends the kernel.
  for (uint32_t round = 0; ((round * runsPerRound) * totalWaves + waveGlobal) * runLen < sectors; ++round) {
    const uint32_t mySector = laneInRound ? ((round * runsPerRound + laneRun) * totalWaves + waveGlobal) * runLen + laneOff : 0xFFFFFFFFu;
    uint32_t myCount = 0, myLay = 0, myHome = 0, myCast = 0;
    bool myStale = false;
    const uint32_t myGx = mySector % p.binSX, myGz = mySector / p.binSX;     // once per 64 sectors, not once per sector
    if (mySector < sectors) {
      // (requested together: one round trip, not five; round 0's are in flight since before the prologue)
      uint32_t lay, hc = 0u, hl = 0u;
      if (round == 0u) { myCount = preCount; lay = preLay; hc = preHc; hl = preHl; myCast = preCast; }
      else {
        myCount = d.binCount[mySector]; lay = d.binLayers[mySector];
        if (p.homeReset) { hc = d.homeCount[mySector]; hl = d.homeLayers[mySector]; }
        if (p.fastPairs) myCast = d.homeCast[mySector];
      }
      myLay = lay;
      // the bin's counters go back to where the next tick starts from: zero, or -- with remembered slots -- the slots that are
      // reserved and the layer summary of their records (binEntityWave, "home slots")
      const bool hot = (hc & kHomeHot) != 0u; hc &= ~kHomeHot;
      if (myCount != hc || lay != hl) { d.binCount[mySector] = hc; d.binLayers[mySector] = hl; }
      myHome = hc;
      const uint32_t hlTest = vocabMode ? layersThatCanMeet(hl, p.vocab) : hl;
      myStale = !freshAll && !hot && hc != 0u && ((hlTest & 0xFFFFu) & (hlTest >> 16)) == 0u &&
                !(myGx == 0u || myGz == 0u || myGx + 1u == p.binSX || myGz + 1u == p.binSZ);      // (!binWrittenEveryTick)
      // vocabulary mode: the bin's own records meet nothing that exists -- without records from elsewhere there is nothing to read
      if (myStale && vocabMode && myCount <= hc) myCount = 0u;
      if (myCount) {
        // no record of this bin can pass the group/mask filter against another one: nothing to read
        if (nbig == 0u && ((lay & 0xFFFFu) & (lay >> 16)) == 0u) myCount = 0u;      // (a crowded sector's overflow slice is reset by the workgroup that takes it off the queue)
      }
    }
    const bool myOver = myCount > kBinCap;             // the sector holds more records than its bin: the rest is in the overflow list
    if (myOver) myCount = kBinCap;
    if (!ballot64(myCount != 0u)) continue;
#ifdef SC_DIAG_WAVETIME
    diagRounds += 1u;
#endif

    // a ring sector on a side where a neighbour tile exists belongs to that neighbour: it reports the pairs whose low
    // corner lies there (it received these boxes through the border exchange).  Decided here, once per 64 sectors.
    bool mine = true;
    {
      const int dx = myGx == 0 ? -1 : (myGx == p.binSX - 1u ? 1 : 0), dz = myGz == 0 ? -1 : (myGz == p.binSZ - 1u ? 1 : 0);
      if ((dx != 0 && hasNb(p, dx, 0)) || (dz != 0 && hasNb(p, 0, dz))) mine = false;    // nearest tile is not this one
    }
    const unsigned long long oursMask = ballot64(mine);
    const uint32_t myGxz = myGx | (myGz << 16);

    // ---- fast sectors (round 4).  The reserved records of a bin were put in order at the learn tick (k_order_home): the cast records --
    // those that pass the filter against their own kind, the dynamic bodies -- are slots [0, D), the rest [D, hc), and no two of the
    // rest can meet (kCastFast).  Whatever lies behind them, slots [hc, n), are VISITORS: boxes that entered the sector since the learn
    // tick, border records, the level kernels' entities -- unclassified, and few.  So nothing is classified here: the visitors simply
    // join the sweeping set -- in LDS the order becomes cast records, visitors, rest -- and every pair with at least one sweeper in it
    // is tested exactly once, the filter deciding at the end as it always does.  The records go to LDS as they lie (a lane past the
    // count loads and stages a null record, slot 64 always holds one), lane (c, g) tests sweeper c against the slots S + g, S + g + G,
    // ... (S sweepers, G = 64 / S), the sweepers meet each other through the triangular table (its first S (S - 1) / 2 entries are
    // exactly the pairs among slots [0, S)), and a round is ONE 16-byte LDS read and four compares on the xz rectangles, AND-ed as
    // wave masks.  A touching pair is not followed up in the round -- one lane's hit used to make the whole wave walk the filter, the
    // low-corner rule and the sink, ~60 instructions, two or three times per sector -- its two slot numbers go to a short list in LDS
    // and the list is resolved behind the rounds, 64 candidates at a time with every lane busy: the full box test (y included),
    // filter, ids, low corner (pairHit).  Same predicate, same pair set.
    // (Tried on top and dropped, profiles/r04/ab_sweeper_lanes_lost.log: LANE = SWEEPER across all fast sectors of the round -- a prefix
    //  sum numbers the sweepers, 64 are taken at a time, each lane walks its sector's slots straight from the bins, nothing staged, no
    //  per-sector set-up at all.  Correct -- 103 parity tests green -- and slow: config 5 90 us against 40, config3dyn 30 against 27: a
    //  slot step is eight dword loads per lane through the vector memory pipe, and that pipe, not instruction issue, then sets the pace.)
    const bool myFast = myCount != 0u && mine && (myCast & kCastFast) != 0u && !myOver && !myStale;
    const unsigned long long fastMask = ballot64(myFast);
    if (fastMask) {
      uint16_t* cand = reinterpret_cast<uint16_t*>(cellMembers[wave]);      // (the wave's cell masks: only the general paths use them)
      // slot 64 of the tile in the fast sectors' layout holds a null record (the general paths use T[0 .. 127] their own way).
      // Null records are READ, from d.nullRec, not built: eight constants in vector registers across these loops were what
      // pushed the kernel into scratch -- a lane past a bin's count simply loads the null record instead of a bin slot.
      if (lane == 0u) {
        const float4 zl = d.nullRec[0], zh = d.nullRec[1];
        T[kBinCap] = make_float4(zl.x, zl.z, zh.x, zh.z); T[kTileSlots + kBinCap] = make_float4(zl.y, zh.y, zl.w, zh.w);
      }
      unsigned long long restF = fastMask;
      // One sector's records are in flight while the one before it is swept: `lo` / `hi` hold lane i's bin slot i of the sector
      // `info` describes (wave-uniform, in scalar registers: n | hc << 8 | D << 16, 0 = no sector left; its grid coordinates).
      // (Deeper prefetch -- three register sets, loads and waits written by hand because hipcc waits with vmcnt(0) -- was built and
      //  measured: no gain, the role is held by instruction issue, not by the records' latency; profiles/r04/ab_fast_sectors.log.)
      float4 lo, hi; uint32_t info, gxzF;
      // (a sector's three numbers are packed in its lane once per round, so a fetch is three cross-lane reads and two scalar selects:
      //  with the packing and the "no sector left" case inside, it was five reads, a handful of scalar shifts and three branches per sector)
      const uint32_t myInfo = myCount | (myHome << 8) | ((myCast & 0xFFu) << 16);
      auto fetch = [&]() __attribute__((always_inline)) {
        const bool more = restF != 0ull;
        const int itF = more ? __ffsll((long long)restF) - 1 : 0;
        restF &= restF - 1ull;
        const uint32_t pk = __builtin_amdgcn_readlane(myInfo, itF), sF = __builtin_amdgcn_readlane(mySector, itF);
        gxzF = __builtin_amdgcn_readlane(myGxz, itF);
        info = more ? pk : 0u;
        const uint32_t nF = info & 0xFFu;
        const float4* r = lane < nF ? d.bins + 2u * ((size_t)sF * kBinCap + lane) : d.nullRec;
        lo = r[0]; hi = r[1];
      };
      fetch();
      while (info) {
        {
          const uint32_t n = info & 0xFFu, hc = (info >> 8) & 0xFFu, D = info >> 16, gxz = gxzF;
          const uint32_t V = n - hc, S = D + V;                      // visitors; sweepers
#ifdef SC_DIAG_WAVETIME
          diagFast += 1u; diagFastWork += S * n;
#endif
          const float secX = (float)(gxz & 0xFFFFu), secZ = (float)(gxz >> 16);
          // (opaque here: the re-arrangement for the tile must not be moved up to the loads -- hipcc did, and waited for every load
          //  right behind its issue to shuffle the components, which left nothing in flight under the sweeps)
          asm volatile("" : "+v"(lo.x), "+v"(lo.y), "+v"(lo.z), "+v"(lo.w), "+v"(hi.x), "+v"(hi.y), "+v"(hi.z), "+v"(hi.w));
          const bool broadcast = S <= kBroadcastMax;
          const float4 clo = lo, chi = hi;                           // this sector's records; `lo` / `hi` go to the next fetch
          if (!broadcast) {
            // the tile in the sweeps' own layout: slot k = (min.x, min.z, max.x, max.z) at T[k], (min.y, max.y, layers, id) at T[65 + k]: a round
            // reads ONE 16-byte word per lane and tests the xz rectangles (four compares); the y interval waits for the resolve -- boxes
            // of a city stand on the ground, their y intervals nearly always overlap
            const uint32_t at = lane + (lane < D ? 0u : (lane < hc ? V : (lane < n ? D - hc : 0u)));      // cast records, visitors, rest
            T[at] = make_float4(clo.x, clo.z, chi.x, chi.z);
            T[kTileSlots + at] = make_float4(clo.y, chi.y, clo.w, chi.w);
          }
          fetch();                                                   // (the next sector's records are under way during the tests below)
#ifdef SC_DIAG_NOSWEEP
          if (false)                                                 // diagnostic build: what fetching and staging the fast sectors' records alone costs
#endif
          if (broadcast) {
            // A handful of sweepers (a lone vehicle among props; one visitor): nothing is staged at all.  Every lane keeps its bin slot's
            // record in registers, each sweeper's xz rectangle is read into scalar registers (v_readlane) and tested by all lanes at
            // once -- four compares with a scalar operand, AND-ed as wave masks with the lanes the sweeper has to meet: a cast record
            // the slots behind it, a visitor the rest and the visitors behind it.  A touch (rare) is followed up on the spot, from registers.
#pragma clang loop unroll(disable)
            for (uint32_t k = 0; k < S; ++k) {
              const uint32_t sl = k < D ? k : hc + (k - D);
              unsigned long long meet = ~((2ull << sl) - 1ull);
              if (k >= D) meet |= ((1ull << hc) - 1ull) & ~((1ull << D) - 1ull);
              const float x0 = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(clo.x), sl)), z0 = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(clo.z), sl));
              const float x1 = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(chi.x), sl)), z1 = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(chi.z), sl));
              const unsigned long long m = ballot64(x0 <= chi.x) & ballot64(clo.x <= x1) & ballot64(z0 <= chi.z) & ballot64(clo.z <= z1) & meet;
              if (!m) continue;
              const float4 smin = make_float4(x0, __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(clo.y), sl)), z0, __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(clo.w), sl)));
              const float4 smax = make_float4(x1, __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(chi.y), sl)), z1, __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(chi.w), sl)));
              uint32_t ia = 0, ib = 0;
              const bool hit = ((m >> lane) & 1ull) && pairHit(p, clo, chi, smin, smax, secX, secZ, ia, ib);
              sinkPush(d, p, sink, hit, ia, ib);
            }
            if (nbig) {
              const bool prim = lane < n && (__float_as_uint(chi.w) & kPrimary);
              const uint32_t myId = __float_as_uint(chi.w) & ~kPrimary;
              for (uint32_t b = 0; b < nbig; ++b) {
                const float4 gmin = d.bigList[2u * (size_t)b], gmax = d.bigList[2u * (size_t)b + 1u];
                const bool hit = prim && boxesOverlap(clo, chi, gmin, gmax) && filterPass(__float_as_uint(clo.w), __float_as_uint(gmin.w));
                sinkPush(d, p, sink, hit, myId, __float_as_uint(gmax.w));
              }
            }
          }
#ifdef SC_DIAG_NOSWEEP
          if (false)
#endif
          if (!broadcast) {
            const uint32_t gi = __builtin_amdgcn_readfirstlane(castTab[S]), G = gi & 0xFFu;
            const uint32_t c = (lane * (gi >> 8)) >> 16;
            const uint32_t cIdx = c < S ? c : kBinCap;               // (idle lanes hold the null record: they touch nothing)
            __builtin_amdgcn_wave_barrier();
            const float4 cxz = T[cIdx];
            // rounds: RA of sweepers against the slots behind them (G slots per sweeper and round), RB of sweepers against each
            // other (64 table entries per round); taken in windows of as many rounds as the candidate list can hold at worst
            // (every lane touching in every round) -- one window for a bin of the usual size, and no test of the list's fill
            // inside the loops
            const uint32_t npairs = S * (S - 1u) / 2u;
            const uint32_t RA = ((n - S + G - 1u) * (gi >> 8)) >> 16, RB = (npairs + 63u) >> 6;
            constexpr uint32_t W = kCandCap / 64u;
            uint32_t j = S + (lane - c * G);
            for (uint32_t base = 0; base < RA + RB; base += W) {
              uint32_t cnt = 0;
              const uint32_t top = base + W < RA + RB ? base + W : RA + RB;
              for (uint32_t r = base; r < (top < RA ? top : RA); ++r, j += G) {
                const uint32_t jj = j < kBinCap ? j : kBinCap;
                const float4 txz = T[jj];
                const unsigned long long m = ballot64(cxz.x <= txz.z) & ballot64(txz.x <= cxz.z) & ballot64(cxz.y <= txz.w) & ballot64(txz.y <= cxz.w);
                if (!m) continue;
                const uint32_t pos = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, cnt));
                if ((m >> lane) & 1ull) cand[pos] = (uint16_t)(cIdx | (jj << 8));
                cnt += (uint32_t)__popcll(m);
              }
              for (uint32_t r = (base > RA ? base : RA); r < top; ++r) {
                const uint32_t q = ((r - RA) << 6) + lane;
                const uint32_t ij = pairTab[q < npairs ? q : 0u];
                const uint32_t a = ij >> 8, b = ij & 255u;
                const float4 axz = T[a], bxz = T[b];
                const unsigned long long m = ballot64(q < npairs) & ballot64(axz.x <= bxz.z) & ballot64(bxz.x <= axz.z) & ballot64(axz.y <= bxz.w) & ballot64(bxz.y <= axz.w);
                if (!m) continue;
                const uint32_t pos = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, cnt));
                if ((m >> lane) & 1ull) cand[pos] = (uint16_t)(a | (b << 8));
                cnt += (uint32_t)__popcll(m);
              }
              // the touching pairs of the window: the full box test, filter, ids, low corner of the intersection in this sector -- every lane busy
              // (Resolving ACROSS sectors -- a sector only notes its three to five touching pairs, one resolve per wave from the records in the
              //  bins -- was built and measured: same pairs, no gain; that resolve's global loads stand exposed at the end of the wave's run.
              //  profiles/r04/ab_deferred_resolve_lost.log)
              __builtin_amdgcn_wave_barrier();
#ifdef SC_DIAG_NORESOLVE
              cnt = 0;                                            // diagnostic build: what the sweeps alone cost (no pairs reported)
#endif
              for (uint32_t q0 = 0; q0 < cnt; q0 += 64u) {
                const uint32_t q = q0 + lane;
                const uint32_t e = q < cnt ? cand[q] : 0u;
                const uint32_t a = e & 255u, b = e >> 8;
                const float4 axz = T[a], ay = T[kTileSlots + a], bxz = T[b], by = T[kTileSlots + b];
                uint32_t ia = 0, ib = 0;
                const bool hit = q < cnt && pairHit(p, make_float4(axz.x, ay.x, axz.y, ay.z), make_float4(axz.z, ay.y, axz.w, ay.w),
                                                    make_float4(bxz.x, by.x, bxz.y, by.z), make_float4(bxz.z, by.y, bxz.w, by.w), secX, secZ, ia, ib);
                sinkPush(d, p, sink, hit, ia, ib);
              }
              __builtin_amdgcn_wave_barrier();
            }
          }
          // big boxes against this bin's primary records (each binned box has exactly one primary copy), as in the general path
          if (nbig && !broadcast) {
            const uint32_t at = lane + (lane < D ? 0u : (lane < hc ? V : (lane < n ? D - hc : 0u)));
            const float4 oxz = T[at], oy = T[kTileSlots + at];
            const bool prim = lane < n && (__float_as_uint(oy.w) & kPrimary);
            const uint32_t myId = __float_as_uint(oy.w) & ~kPrimary;
            for (uint32_t b = 0; b < nbig; ++b) {
              const float4 gmin = d.bigList[2u * (size_t)b], gmax = d.bigList[2u * (size_t)b + 1u];
              const bool hit = prim && oxz.x <= gmax.x && gmin.x <= oxz.z && oy.x <= gmax.y && gmin.y <= oy.y && oxz.y <= gmax.z && gmin.z <= oxz.w &&
                               filterPass(__float_as_uint(oy.z), __float_as_uint(gmin.w));
              sinkPush(d, p, sink, hit, myId, __float_as_uint(gmax.w));
            }
          }
          __builtin_amdgcn_wave_barrier();
        }
      }
    }
    const unsigned long long work = ballot64(myCount != 0u && !myFast);
    if (!work) continue;
    // lazy records: bins of unwritten records that have to be read after all (a record from elsewhere made them admissible; a
    // big box is about) are rebuilt in place first -- rare, and outside the loop below
    const unsigned long long staleMask = ballot64(myStale && myCount != 0u);
    if (staleMask) {
      for (unsigned long long st = staleMask; st; st &= st - 1ull) {
        const int k = __ffsll((long long)st) - 1;
        const uint32_t sK = __shfl(mySector, k, 64), hcK = __shfl(myHome, k, 64);
        if (lane < hcK) {
          float4* r = d.bins + 2u * ((size_t)sK * kBinCap + lane);
          float4 lo, hi;
          if (vocabMode) nullRecord(lo, hi);               // (the owners' records meet nothing: records from elsewhere are searched among themselves)
          else { lo = r[0]; hi = r[1]; rebuildHomeRecord(d, p, sK, lo, hi); }
          r[0] = lo; r[1] = hi;
        }
      }
      __threadfence();
      // (marked behind the fence: whoever sees the mark -- the workgroup that takes a crowded sector off the queue -- sees the records)
      if (!vocabMode && myStale && myCount != 0u) d.homeCount[mySector] = myHome | kHomeHot;
    }
    // what a sector's turn needs from the lane that holds it: index, count, layer summary, grid coordinates -- four
    // cross-lane reads.  (ds_bpermute although the picked lane is wave-uniform: v_readlane measured slower, 27.1 against
    // 26.3 us on config3dyn -- a VALU slot each plus SGPR-hazard waits; and recomputing the index from the lane number
    // made the compiler re-load a kernel argument inside this loop: the scalar registers are all taken.)
    // software pipeline: records of the next non-empty sector are in flight while this one is tested
    int it = __ffsll((long long)work) - 1;
    uint32_t n = __shfl(myCount, it, 64), gxz = __shfl(myGxz, it, 64), binLay = __shfl(myLay, it, 64);
    uint32_t s = __shfl(mySector, it, 64);            // (cross-lane reads stay outside divergent code: every lane takes part)
    float4 rmin = make_float4(0, 0, 0, 0), rmax = make_float4(0, 0, 0, 0);
    if (lane < n) {
      const float4* r = d.bins + 2u * ((size_t)s * kBinCap + lane);
      rmin = r[0]; rmax = r[1];
    }
    unsigned long long rest = work & ~(1ull << it);
    while (it >= 0) {
      const int itNext = rest ? __ffsll((long long)rest) - 1 : -1;
      uint32_t nNext = 0, gxzNext = 0, binLayNext = 0, sNext = 0;
      float4 nmin = make_float4(0, 0, 0, 0), nmax = make_float4(0, 0, 0, 0);
      if (itNext >= 0) {
        rest &= ~(1ull << itNext);
        nNext = __shfl(myCount, itNext, 64); gxzNext = __shfl(myGxz, itNext, 64); binLayNext = __shfl(myLay, itNext, 64);
        sNext = __shfl(mySector, itNext, 64);
        if (lane < nNext) {
          const float4* r = d.bins + 2u * ((size_t)sNext * kBinCap + lane);
          nmin = r[0]; nmax = r[1];
        }
      }

#ifdef SC_DIAG_WAVETIME
      diagGen += 1u; diagGenN += n;
#endif
      const uint32_t gx = gxz & 0xFFFFu, gz = gxz >> 16;
      const bool ours = (oursMask >> it) & 1ull;
      const bool valid = lane < n && ours;
      // Filter first (integer, cheap): like btDbvtBroadphase, which keeps static bodies in a separate set
      // and never tests fixed-vs-fixed, a record that cannot pass (a.group & b.mask) && (b.group & a.mask)
      // against ANY other record of this bin is dropped before the box tests, and a bin without a single
      // admissible record (e.g. only static props: group 2 / mask 1) is skipped altogether.  The reported
      // pair set is unchanged -- the filter is part of the pair predicate either way.
      // (the OR of the bin's layer words comes with the bin: the summary the fused kernel keeps -- it may hold a record more
      //  than the tile does (overflow), which only makes this pre-filter a little weaker; pairHit applies the filter itself)
      const uint32_t lay = valid ? __float_as_uint(rmin.w) : 0u;
      const uint32_t all = binLay;
      const bool admissible = valid && ((lay & 0xFFFFu) & (all >> 16)) != 0u && ((all & 0xFFFFu) & (lay >> 16)) != 0u;
      const unsigned long long validMask = ballot64(admissible);
      const bool anyPairs = __popcll(validMask) >= 2;
      const float secX = (float)gx, secZ = (float)gz;
      // (Round 2's broadcast paths over the records that can collide at all -- dynamic bodies against the rest, lane = (cast record,
      //  partner phase) -- moved to the fast sectors in round 4, where the bins arrive sorted that way and nothing is classified per
      //  tick.  What still comes here -- bins whose reserved records can meet across kinds (say 4/8 against 8/4), unwritten bins that
      //  had to be rebuilt, worlds without remembered slots -- takes the general forms: the triangular table or the 4x4 cell grid.)
      if (anyPairs) {
        T[2u * lane] = rmin; T[2u * lane + 1u] = rmax;
        __builtin_amdgcn_wave_barrier();
        if (n <= kFineThreshold) {
          // sparse bin: all n(n-1)/2 record pairs spread over the lanes through the triangular table
          const uint32_t npairs = n * (n - 1u) / 2u;
          for (uint32_t q0 = 0; q0 < npairs; q0 += 64u) {
            const uint32_t q = q0 + lane;
            bool hit = false;
            uint32_t ia = 0, ib = 0;
            if (q < npairs) {
              const uint32_t ij = pairTab[q];
              const uint32_t i = ij >> 8, j = ij & 255u;
              if ((validMask >> i) & (validMask >> j) & 1ull)
                hit = pairHit(p, T[2u * i], T[2u * i + 1u], T[2u * j], T[2u * j + 1u], secX, secZ, ia, ib);
            }
            sinkPush(d, p, sink, hit, ia, ib);
          }
        } else {
          // dense bin: LDS grid of 4x4 cells (16 m at the default sector size) inside the tile.  A box covering
          // at most 2x2 cells is "small": it registers in its cells with LDS atomics and only meets the small
          // boxes sharing a cell (any monotone cell function keeps every overlapping pair, clamping included);
          // the few wider boxes (ground slabs span the sector) are broadcast partners for everybody.
          unsigned long long* cells = cellMembers[wave];
          if (lane < 16u) cells[lane] = 0ull;
          const float ox = (secX + p.binOx) * 4.0f, oz = (secZ + p.binOz) * 4.0f, inv4 = p.invSector * 4.0f;
          const int cx0 = min(3, max(0, (int)floorf(rmin.x * inv4 - ox))), cx1 = min(3, max(0, (int)floorf(rmax.x * inv4 - ox)));
          const int cz0 = min(3, max(0, (int)floorf(rmin.z * inv4 - oz))), cz1 = min(3, max(0, (int)floorf(rmax.z * inv4 - oz)));
          const bool small = admissible && (cx1 - cx0) <= 1 && (cz1 - cz0) <= 1;
          const bool wide = admissible && !small;
          __builtin_amdgcn_wave_barrier();
          const uint32_t c00 = (uint32_t)(cz0 * 4 + cx0), c01 = (uint32_t)(cz0 * 4 + cx1), c10 = (uint32_t)(cz1 * 4 + cx0), c11 = (uint32_t)(cz1 * 4 + cx1);
          if (small) {
            const unsigned long long bit = 1ull << lane;
            atomicOr(&cells[c00], bit);
            if (c01 != c00) atomicOr(&cells[c01], bit);
            if (c10 != c00) atomicOr(&cells[c10], bit);
            if (c11 != c01 && c11 != c10) atomicOr(&cells[c11], bit);
          }
          __builtin_amdgcn_wave_barrier();
          unsigned long long cand = 0ull;
          if (small) cand = (cells[c00] | cells[c01] | cells[c10] | cells[c11]) & ((1ull << lane) - 1ull);   // partners j < i
          while (ballot64(cand != 0ull)) {
            bool hit = false; uint32_t ia = 0, ib = 0;
            if (cand) {
              const uint32_t j = (uint32_t)__ffsll((long long)cand) - 1u;
              cand &= cand - 1ull;
              hit = pairHit(p, rmin, rmax, T[2u * j], T[2u * j + 1u], secX, secZ, ia, ib);
            }
            sinkPush(d, p, sink, hit, ia, ib);
          }
          unsigned long long wides = ballot64(wide);
          while (wides) {
            const uint32_t wI = (uint32_t)__ffsll((long long)wides) - 1u;
            wides &= wides - 1ull;
            bool hit = false; uint32_t ia = 0, ib = 0;
            // every small box meets every wide one; two wide boxes meet once (lower lane tests against the higher)
            if (admissible && lane != wI && (small || lane < wI)) hit = pairHit(p, rmin, rmax, T[2u * wI], T[2u * wI + 1u], secX, secZ, ia, ib);
            sinkPush(d, p, sink, hit, ia, ib);
          }
        }
      }

      // big boxes against this bin's primary records (each binned box has exactly one primary copy)
      if (nbig) {
        const bool mine = valid && (__float_as_uint(rmax.w) & kPrimary);
        const uint32_t myId = __float_as_uint(rmax.w) & ~kPrimary;
        for (uint32_t b = 0; b < nbig; ++b) {
          const float4 gmin = d.bigList[2u * (size_t)b], gmax = d.bigList[2u * (size_t)b + 1u];
          const bool hit = mine && boxesOverlap(rmin, rmax, gmin, gmax) && filterPass(__float_as_uint(rmin.w), __float_as_uint(gmin.w));
          sinkPush(d, p, sink, hit, myId, __float_as_uint(gmax.w));
        }
      }
      // Sector overflow: records that found this sector's bin full -- this tile's own boxes and border records that
      // arrived from a neighbour alike -- sit in the overflow list, tagged with the sector.  They belong to the sector as much
      // as the records in the bin: each meets the bin's records and the sector's other overflow records under the same rule
      // (low corner of the intersection in this sector), and -- if it is its box's primary copy -- the big boxes.
      // That part of a crowded sector's search is not this wave's: the sector is on the tick's queue of crowded sectors (put
      // there by whoever was handed slot 64 of its bin) and a whole workgroup takes it behind the sweep -- "crowded sectors" below.
      __builtin_amdgcn_wave_barrier();
      it = itNext; n = nNext; s = sNext; gxz = gxzNext; binLay = binLayNext; rmin = nmin; rmax = nmax;
    }
  }

#ifdef SC_DIAG_WAVETIME
  {
    const unsigned long long diagT1 = wall_clock64();
    if ((threadIdx.x & 63u) == 0u) {
      atomicAdd(&g_waveDiag[0], diagT1 - diagT0); atomicMax(&g_waveDiag[1], diagT1 - diagT0); atomicAdd(&g_waveDiag[2], 1ull);
      atomicMin(&g_waveDiag[3], diagT0); atomicMax(&g_waveDiag[4], diagT1); atomicMax(&g_waveDiag[5], diagT0);
      if (waveGlobal < 8192u) { uint32_t* r = g_waveRows + 8u * waveGlobal; r[0] = (uint32_t)(diagT1 - diagT0); r[1] = diagFast; r[2] = diagFastWork; r[3] = diagGen; r[4] = diagGenN; r[5] = diagRounds; r[6] = (uint32_t)diagT0; r[7] = waveGlobal; }
      { unsigned long long b = (diagT1 - diagT0) >> 7; if (b > 31ull) b = 31ull; atomicAdd(&g_waveDiag[8 + b], 1ull); }      // 1.28 us buckets (100 MHz)
    }
  }
#endif


  // ---- crowded sectors, four waves to a sector (round 3).  A sector at the engine's own budget -- 200 boxes -- is a bin and
  // three tiles of overflow records; searched by the one wave that met it, it was a serial chain of dependent loads (the
  // slice sweep, four registrations, ten tile stagings, each through an index and a record) with nothing else on that SIMD to
  // hide them: 120 us for a district of 512 such sectors while nine tenths of the chip idled (profiles/r02/crowded_sectors.json).
  // Now the workgroup's waves share one parked sector at a time: the sweep of the sector's slice of the overflow list is cut
  // into chunks of 256 tags taken in turn (the matches' list positions go to one scratch row through an LDS counter: their
  // order is immaterial), the tiles register their boxes in the 4x4 cell masks in turn, and the (register tile, staged tile)
  // combinations -- k(k+3)/2 of them for k overflow tiles -- go round the waves; each wave keeps its own staging tile and pair
  // buffer.  Same predicate, same pair set.
  const uint32_t crowded = min(d.counters[ctr + kCtrCrowdTail], sectors);      // complete: the binning kernels are over
  for (; crowded != 0u;) {
    __syncthreads();                                                  // (the sweep, or the last sector, is over for every wave)
    if (threadIdx.x == 0) {
      const uint32_t idx = atomicAdd(&d.counters[ctr + kCtrCrowdHead], 1u);   // (one fetch-and-add per sector: nobody queues during this launch)
      crowdItem = idx < crowded ? d.crowdQueue[p.parity * sectors + idx] : 0xFFFFFFFFu;
    }
    __syncthreads();
    const uint32_t s = crowdItem;
    if (s >= sectors) break;
    const uint32_t n = kBinCap;                                       // (a crowded sector's bin is full)
    const uint32_t gx = s % p.binSX, gz = s / p.binSX;
    const float secX = (float)gx, secZ = (float)gz;
    // a ring sector on a side where a neighbour tile exists belongs to that neighbour: only its slice is reset
    bool ours = true;
    {
      const int dx = gx == 0 ? -1 : (gx == p.binSX - 1u ? 1 : 0), dz = gz == 0 ? -1 : (gz == p.binSZ - 1u ? 1 : 0);
      if ((dx != 0 && hasNb(p, dx, 0)) || (dz != 0 && hasNb(p, 0, dz))) ours = false;
    }
    uint32_t* row = d.ovfIdx + (size_t)(bid * (kTile / 64u) < kOvfWaves ? bid * (kTile / 64u) : kOvfWaves - 1u) * kOvfPerSector;
    uint32_t eLo = d.ovfLo[s], eHi = d.ovfHi[s];
    // lazy records: the bin's remembered slots hold old records -- rebuilt in place (the sweep's wave may have done so already,
    // or be doing it now: same ids, same values)
    const uint32_t hcRaw = freshAll ? kHomeHot : d.homeCount[s];     // (the sweep's wave may have marked the bin hot a moment ago: then it rebuilt it, too)
    if (!(hcRaw & kHomeHot) && !binWrittenEveryTick(vocabMode ? layersThatCanMeet(d.homeLayers[s], p.vocab) : d.homeLayers[s], s, p.binSX, p.binSZ)) {
      if (wave == 0u && lane < hcRaw) {
        float4* r = d.bins + 2u * ((size_t)s * kBinCap + lane);
        float4 lo, hi;
        if (vocabMode) nullRecord(lo, hi);
        else { lo = r[0]; hi = r[1]; rebuildHomeRecord(d, p, s, lo, hi); }
        r[0] = lo; r[1] = hi;
      }
      __threadfence();
    }
    if (eHi > novf) eHi = novf;
    if (!ours) eLo = eHi;
    __syncthreads();                                                  // (everybody has the slice bounds)
    if (threadIdx.x == 0) { d.ovfLo[s] = 0xFFFFFFFFu; d.ovfHi[s] = 0u; }
    for (uint32_t e0 = eLo + wave * 256u; e0 < eHi; e0 += 4u * 256u) {      // four tags per lane in flight, as the one-wave sweep
      uint32_t tag[4];
#pragma unroll
      for (uint32_t u = 0; u < 4u; ++u) { const uint32_t q = e0 + u * 64u + lane; tag[u] = q < eHi ? d.spillSector[q] : 0xFFFFFFFFu; }
#pragma unroll
      for (uint32_t u = 0; u < 4u; ++u) {
        const bool match = tag[u] == s;
        const unsigned long long mm = ballot64(match);
        if (!mm) continue;
        uint32_t base = 0;
        if (lane == 0) base = atomicAdd(&crowdGathered, (uint32_t)__popcll(mm));
        base = __shfl(base, 0, 64);
        const uint32_t at = base + (uint32_t)__popcll(mm & ((1ull << lane) - 1ull));
        if (match && at < kOvfPerSector) row[at] = e0 + u * 64u + lane;
      }
    }
    __threadfence_block();
    __syncthreads();
    uint32_t m = crowdGathered;
    if (m > kOvfPerSector) { if (threadIdx.x == 0) atomicAdd(&d.counters[ctr + kCtrBorderLost], m - kOvfPerSector); m = kOvfPerSector; }
    const uint32_t tiles = (m + 63u) / 64u + 1u;
    unsigned long long* CM = cellMembers[0];                          // one set of cell masks for the workgroup
    for (uint32_t t = threadIdx.x; t < tiles * 16u; t += kTile) CM[t] = 0ull;
    __syncthreads();
    if (threadIdx.x == 0) crowdGathered = 0u;                         // (for the next parked sector: nobody reads it again before the barrier above)
    const float ox = (secX + p.binOx) * 4.0f, oz = (secZ + p.binOz) * 4.0f, inv4 = p.invSector * 4.0f;
    auto loadRec = [&](uint32_t t, float4& lo, float4& hi) -> bool {
      lo = make_float4(0, 0, 0, 0); hi = make_float4(0, 0, 0, 0);
      if (t == 0u) {
        if (lane >= n) return false;
        const float4* rec = d.bins + 2u * ((size_t)s * kBinCap + lane);
        lo = rec[0]; hi = rec[1];
        return true;
      }
      const uint32_t q = (t - 1u) * 64u + lane;
      if (q >= m) return false;
      const uint32_t idx = row[q];
      lo = d.spill[2u * (size_t)idx]; hi = d.spill[2u * (size_t)idx + 1u];
      return true;
    };
    auto cellRange = [&](const float4& lo, const float4& hi, int& cx0, int& cx1, int& cz0, int& cz1) {
      cx0 = min(3, max(0, (int)floorf(lo.x * inv4 - ox))); cx1 = min(3, max(0, (int)floorf(hi.x * inv4 - ox)));
      cz0 = min(3, max(0, (int)floorf(lo.z * inv4 - oz))); cz1 = min(3, max(0, (int)floorf(hi.z * inv4 - oz)));
    };
    for (uint32_t t = wave; t < tiles; t += kTile / 64u) {           // registration, a tile per wave in turn
      float4 lo, hi; int cx0, cx1, cz0, cz1;
      const bool has = loadRec(t, lo, hi);
      cellRange(lo, hi, cx0, cx1, cz0, cz1);
      if (has) for (int cz = cz0; cz <= cz1; ++cz) for (int cx = cx0; cx <= cx1; ++cx) atomicOr(&CM[t * 16u + (uint32_t)(cz * 4 + cx)], 1ull << lane);
      // the sector's layer summary, from the records themselves (the sweep has reset the bin's own by now)
      uint32_t lay = has ? __float_as_uint(lo.w) : 0u;
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) lay |= (uint32_t)__shfl_xor((int)lay, o, 64);
      if (lane == 0 && lay) atomicOr(&crowdLayers, lay);
    }
    __syncthreads();
    const uint32_t sectorLay = crowdLayers;
    __syncthreads();
    if (threadIdx.x == 0) crowdLayers = 0u;
    // no record of this sector can pass the group/mask filter against another one, and no big box is about: nothing to test
    if (nbig == 0u && ((sectorLay & 0xFFFFu) & (sectorLay >> 16)) == 0u) continue;
    uint32_t combo = 0;
    for (uint32_t k = 1; k < tiles; ++k) {                           // (tile 0 against itself was the sweep's work)
      for (uint32_t jt = 0; jt <= k; ++jt, ++combo) {
        if ((combo & (kTile / 64u - 1u)) != wave) continue;
        float4 xmin, xmax; int cx0, cx1, cz0, cz1;
        const bool has = loadRec(k, xmin, xmax);
        cellRange(xmin, xmax, cx0, cx1, cz0, cz1);
        __builtin_amdgcn_wave_barrier();
        if (jt == k) { T[2u * lane] = xmin; T[2u * lane + 1u] = xmax; }
        else { float4 lo, hi; loadRec(jt, lo, hi); T[2u * lane] = lo; T[2u * lane + 1u] = hi; }
        __builtin_amdgcn_wave_barrier();
        unsigned long long cand = 0ull;
        if (has) {
          for (int cz = cz0; cz <= cz1; ++cz) for (int cx = cx0; cx <= cx1; ++cx) cand |= CM[jt * 16u + (uint32_t)(cz * 4 + cx)];
          if (jt == k) cand &= (1ull << lane) - 1ull;          // inside a tile: partners in lower lanes
        }
        while (ballot64(cand != 0ull)) {
          bool hit = false; uint32_t ia = 0, ib = 0;
          if (cand) {
            const uint32_t i = (uint32_t)__ffsll((long long)cand) - 1u;
            cand &= cand - 1ull;
            hit = pairHit(p, xmin, xmax, T[2u * i], T[2u * i + 1u], secX, secZ, ia, ib);
          }
          sinkPush(d, p, sink, hit, ia, ib);
        }
        if (jt == 0u && nbig) {                                  // big boxes against this tile's primary copies, once per tile
          const bool mine = has && (__float_as_uint(xmax.w) & kPrimary);
          const uint32_t xid = __float_as_uint(xmax.w) & ~kPrimary;
          for (uint32_t b2 = 0; b2 < nbig; ++b2) {
            const float4 gmin = d.bigList[2u * (size_t)b2], gmax = d.bigList[2u * (size_t)b2 + 1u];
            const bool hit = mine && boxesOverlap(xmin, xmax, gmin, gmax) && filterPass(__float_as_uint(xmin.w), __float_as_uint(gmin.w));
            sinkPush(d, p, sink, hit, xid, __float_as_uint(gmax.w));
          }
        }
      }
    }
  }

  // big boxes against each other: wave w takes big b = w, w + totalWaves, ...; lanes sweep the partners after b.
  // On a tiled world the list also holds the neighbours' big boxes that reach this tile, every tile that knows both
  // boxes sees the pair, and the one owning the sector with the low corner of the intersection reports it.
  const bool tiled = p.neighbourMask != 0u;
  for (uint32_t b = waveGlobal; b < nbig; b += totalWaves) {
    const float4 gmin = d.bigList[2u * (size_t)b], gmax = d.bigList[2u * (size_t)b + 1u];
    for (uint32_t j0 = b + 1u; j0 < nbig; j0 += 64u) {
      const uint32_t j = j0 + lane;
      bool hit = false; uint32_t jid = 0;
      if (j < nbig) {
        const float4 hmin = d.bigList[2u * (size_t)j], hmax = d.bigList[2u * (size_t)j + 1u];
        jid = __float_as_uint(hmax.w);
        hit = boxesOverlap(gmin, gmax, hmin, hmax) && filterPass(__float_as_uint(gmin.w), __float_as_uint(hmin.w));
        if (hit && tiled) {
          const float lx = gmin.x > hmin.x ? gmin.x : hmin.x, lz = gmin.z > hmin.z ? gmin.z : hmin.z;
          hit = ownsSector(p, floorf(lx * p.invSector) - p.binOx, floorf(lz * p.invSector) - p.binOz);
        }
      }
      sinkPush(d, p, sink, hit, __float_as_uint(gmax.w), jid);
    }
  }
  sinkFlush(d, p, sink);
}

template <bool kVocab>
__global__ __launch_bounds__(kTile) SC_PAIR_OCC void k_pairs(const DeviceState d, const TickParams p)
{
  __shared__ float4 tile[kTile / 64][2 * kTileSlots];     // 2 KiB per wave: the bin as an LDS tile (+ a null record)
  __shared__ __attribute__((aligned(16))) uint16_t pairTab[kPairTabSize];              // q -> (i << 8 | j), 0 <= j < i < 64
  __shared__ uint2 pairBuf[kTile / 64][kWavePairBuf];     // 2 KiB per wave: hits waiting for a flush
  __shared__ unsigned long long cellMembers[kTile / 64][kCellWords];   // per wave: which records touch each of the 4x4 cells, per tile of 64
  pairsBody<kVocab>(d, p, blockIdx.x, gridDim.x, tile, pairTab, pairBuf, cellMembers);
}

// read-back helper: concatenates the shards' segments into one list and writes the total found
__global__ __launch_bounds__(kTile) void k_gather_pairs(const DeviceState d, const TickParams p, uint32_t parity, uint2* __restrict__ dst, uint32_t* __restrict__ total)
{
  const uint32_t shard = blockIdx.x;
  const uint32_t shardCap = p.maxPairs / kPairShards;
  uint32_t before = 0, found = 0;
  for (uint32_t s = 0; s < kPairShards; ++s) {
    const uint32_t c = d.pairShardCount[(parity * kPairShards + s) * kShardStride];
    found += c < shardCap ? c : shardCap;                // (a counter can read above the cap only transiently)
    if (s < shard) before += c < shardCap ? c : shardCap;
  }
  const uint32_t mine = d.pairShardCount[(parity * kPairShards + shard) * kShardStride];
  const uint32_t take = mine < shardCap ? mine : shardCap;
  for (uint32_t q = threadIdx.x; q < take; q += kTile) dst[before + q] = d.pairs[(size_t)shard * shardCap + q];
  if (shard == 0 && threadIdx.x == 0) {
    const uint32_t dropped = d.counters[kCtrPar + 8u * parity + kCtrPairs];   // pairs found after every segment was full
    total[0] = found + dropped; total[1] = dropped ? 1u : 0u;
  }
}

// compaction and pair search both depend only on the fused kernel: one launch, workgroups split by role
template <bool kEmit>
__global__ __launch_bounds__(kTile) SC_PAIR_OCC void k_compact_pairs(const DeviceState d, const TickParams p, uint32_t compactBlocks, uint32_t group)
{
  __shared__ float4 tile[kTile / 64][2 * kTileSlots];
  __shared__ __attribute__((aligned(16))) uint16_t pairTab[kPairTabSize];
  __shared__ uint2 pairBuf[kTile / 64][kWavePairBuf];
  __shared__ unsigned long long cellMembers[kTile / 64][kCellWords];
  __shared__ uint32_t scratch[kTile / 64];
  __shared__ uint32_t moved[kMaxSpanWords];
  // (a workgroup plays one role: the compaction role borrows the pair role's tile area -- 8 KiB >= kCompactLdsWords dwords)
  static_assert(sizeof(tile) >= kCompactLdsWords * sizeof(uint32_t), "compaction scratch does not fit the pair tiles");
  // Which role takes the FIRST workgroup indices matters when the launch does not fit the chip at once (1M entities: ~512 compaction + 1041
  // pair workgroups against 1280 resident ones at five per CU): workgroups start in index order, and the ones that do not fit start when
  // others end.  The pair role is the long one on a world that searches (16 sectors a wave, 20-30 us), the compaction role takes ~8 us: with
  // the compaction role first, a quarter of the pair workgroups started 8 us late and ended the kernel 8 us late (round 4; the average wave
  // lived 32 of the kernel's 40 us on config 5).  Pair role first: every pair workgroup starts at once, the compaction workgroups that do not
  // fit start behind the first compaction workgroups that end -- earlier compaction workgroups are always resident before later ones, which
  // is what their look-back needs.  (Both orders behind a run-time switch doubled the kernel's code and cost every world 2-4 us.)
  const uint32_t pairBlocks = gridDim.x - compactBlocks;
  if (blockIdx.x < pairBlocks) pairsBody<false>(d, p, blockIdx.x, pairBlocks, tile, pairTab, pairBuf, cellMembers);
  else compactBody<kEmit>(d, p, blockIdx.x - pairBlocks, compactBlocks, group, scratch, moved, reinterpret_cast<uint32_t*>(&tile[0][0]));
}

// ------------------------------------------------------------------------------------------
// Border exchange (multi-GPU tiles).  After the fused kernel the ring bins of this tile hold exactly
// the boxes that reach into a neighbour's sectors.  k_border_pack turns each ring side that has a
// neighbour into one fixed-capacity message (per-bin counts + records packed bin after bin);
// k_border_merge appends what the neighbours sent into this tile's core-edge bins.  One workgroup
// per direction; a ring side is at most a few hundred bins.
// ------------------------------------------------------------------------------------------
// Ownership of ring sectors: a sector outside this tile's core belongs to the tile nearest to it
// (world coordinates clamped into the world), so an out-of-world ring cell next to a neighbour's
// column is that neighbour's.  hasNb(dx,dz): a tile exists one step that way.

// ring cell l of side (dx,dz) in this tile's bin grid; *send tells whether this tile hands it to that neighbour
__device__ __forceinline__ uint32_t ringCell(const TickParams& p, int dx, int dz, uint32_t l, bool* send)
{
  const uint32_t coreSX = p.binSX - 2u, coreSZ = p.binSZ - 2u;
  uint32_t gx, gz;
  if (dx != 0 && dz != 0) { gx = dx < 0 ? 0u : coreSX + 1u; gz = dz < 0 ? 0u : coreSZ + 1u; *send = true; }
  else if (dx != 0) {
    gx = dx < 0 ? 0u : coreSX + 1u; gz = l;
    *send = (l >= 1u && l <= coreSZ) || (l == 0u && !hasNb(p, 0, -1)) || (l == coreSZ + 1u && !hasNb(p, 0, 1));
  } else {
    gz = dz < 0 ? 0u : coreSZ + 1u; gx = l;
    *send = (l >= 1u && l <= coreSX) || (l == 0u && !hasNb(p, -1, 0)) || (l == coreSX + 1u && !hasNb(p, 1, 0));
  }
  return gz * p.binSX + gx;
}
// where cell l of the message received from the neighbour in direction (dx,dz) lands in this tile's grid
__device__ __forceinline__ uint32_t landingCell(const TickParams& p, int dx, int dz, uint32_t l)
{
  const uint32_t coreSX = p.binSX - 2u, coreSZ = p.binSZ - 2u;
  uint32_t gx, gz;
  if (dx != 0 && dz != 0) { gx = dx < 0 ? 1u : coreSX; gz = dz < 0 ? 1u : coreSZ; }
  else if (dx != 0) { gx = dx < 0 ? 1u : coreSX; gz = l; }
  else { gz = dz < 0 ? 1u : coreSZ; gx = l; }
  return gz * p.binSX + gx;
}

// exclusive prefix of one value per thread over the 256 threads of a workgroup (+ carry); *total = carry + sum.
// sWave: 4 words of LDS.  Ends with a barrier, so sWave can be reused at once.
__device__ __forceinline__ uint32_t blockScanExclusive(uint32_t v, uint32_t carry, uint32_t* sWave, uint32_t* total)
{
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  uint32_t incl = v;
#pragma unroll
  for (uint32_t o = 1; o < 64u; o <<= 1) { const uint32_t up = __shfl_up(incl, o, 64); if (lane >= o) incl += up; }
  if (lane == 63u) sWave[wave] = incl;
  __syncthreads();
  uint32_t before = carry;
  for (uint32_t w = 0; w < wave; ++w) before += sWave[w];
  *total = carry + sWave[0] + sWave[1] + sWave[2] + sWave[3];
  __syncthreads();
  return before + incl - v;
}

__device__ __forceinline__ void borderPackBody(const DeviceState& d, const TickParams& p, uint32_t dir)
{
  __shared__ uint32_t sWave[4];
  if (!((p.neighbourMask >> dir) & 1u) || !d.borderSend[dir]) return;
  int dx, dz; borderDir(dir, dx, dz);
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t L = borderLen(dir, p.binSX - 2u, p.binSZ - 2u);
  uint32_t* msg = d.borderSend[dir];
  const uint32_t cap = borderRecCap(L, p.borderRecs);
  float4* records = reinterpret_cast<float4*>(msg + kBorderHeader + L);
  // Layout of the bin part (round 4): per-cell counts, then kBorderFixed FIXED record slots per cell, then a shared spill-over area
  // for what a cell holds beyond its fixed slots (packed cell after cell, offsets from a scan over the extras).  The kernel is a
  // chain of dependent memory round trips and nothing else; with fixed slots a cell's first records are requested together with
  // its count and stored where no other cell's count matters -- the scan, and the loads and stores behind it, exist only on
  // ticks when some ring cell holds more than kBorderFixed records (round 3 packed everything behind the scan: count -> scan ->
  // record load -> store; the usual tick is now count+records -> store).  A thread takes one cell (kTile cells in flight per pass).
  // A ring sector that holds more than its bin (round 3): the sector's counter kept counting and the rest of its records sit in
  // the sector overflow list, tagged with the sector.  They cross the border like the bin's records -- the message reserves
  // room for the sector's whole count, the thread copies the bin part, and the WAVE then sweeps the sector's slice of the
  // overflow list for the rest (the pair search gathers a sector's overflow the same way).  What a message cannot hold, and
  // what lies beyond a sector's 64 + 1024 records, is counted in border_lost.
  const uint32_t ctr = kCtrPar + 8u * p.parity;
  const uint32_t nLocal = min(d.counters[ctr + kCtrBig], p.bigCap);          // the merge has not run yet: only this tile's boxes
  const uint32_t novf = min(d.counters[ctr + kCtrSpill], p.ovfCap);
  const uint32_t K = borderFixedSlots(L, p.borderRecs);
  float4* extraRec = records + 2u * ((size_t)L * K);
  const uint32_t capX = cap - L * K;
  // the overflow part of the bins a wave's lanes hold: one bin after the other, all 64 lanes on its slice of the list
  // (off: where the cell's extras start in the spill-over area; its records K..63 come first, the list's behind them)
  auto sweepOverflow = [&](uint32_t cell, uint32_t off, uint32_t take) {
    unsigned long long todo = ballot64(take > kBinCap);
    while (todo) {
      const int src = __ffsll((long long)todo) - 1;
      todo &= todo - 1ull;
      const uint32_t wCell = __shfl(cell, src, 64), wOff = __shfl(off, src, 64), wTake = __shfl(take, src, 64);
      const uint32_t want = wTake - kBinCap;                 // records to find in the list
      uint32_t eLo = d.ovfLo[wCell], eHi = d.ovfHi[wCell];
      if (eHi > novf) eHi = novf;
      float4* dst = extraRec + 2u * ((size_t)wOff + kBinCap - K);
      uint32_t found = 0;
      for (uint32_t e0 = eLo; e0 < eHi && found < want; e0 += 64u) {
        const uint32_t e = e0 + lane;
        const bool match = e < eHi && d.spillSector[e] == wCell;
        const unsigned long long mm = ballot64(match);
        const uint32_t at = found + (uint32_t)__popcll(mm & ((1ull << lane) - 1ull));
        if (match && at < want) { dst[2u * at] = d.spill[2u * (size_t)e]; dst[2u * at + 1u] = d.spill[2u * (size_t)e + 1u]; }
        found += (uint32_t)__popcll(mm);
      }
      // (the counter promised more than the list holds -- only when the list itself ran out: keep the count, pad with nothing)
      for (uint32_t q = (found < want ? found : want) + lane; q < want; q += 64u) { float4 lo, hi; nullRecord(lo, hi); dst[2u * q] = lo; dst[2u * q + 1u] = hi; }
    }
  };
  uint32_t carry = 0;                                         // records in the spill-over area so far
  for (uint32_t base = 0; base < L; base += kTile) {
    const uint32_t l = base + threadIdx.x;
    uint32_t c = 0, cell = 0, over = 0;
    static_assert(kBorderFixed == 4, "the fixed slots are spelled out (arrays of float4 end up in scratch)");
    const float4 zero4 = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    float4 a0 = zero4, b0 = zero4, a1 = zero4, b1 = zero4, a2 = zero4, b2 = zero4, a3 = zero4, b3 = zero4;
    bool send = false;
    if (l < L) cell = ringCell(p, dx, dz, l, &send);
    if (send) {
      c = d.binCount[cell];
      const float4* r = d.bins + 2u * ((size_t)cell * kBinCap);       // (a bin row is there whatever it holds)
      a0 = r[0]; b0 = r[1]; a1 = r[2]; b1 = r[3]; a2 = r[4]; b2 = r[5]; a3 = r[6]; b3 = r[7];
    }
    if (c > kSectorRecMax) { over += c - kSectorRecMax; c = kSectorRecMax; }
    const uint32_t nf = c < K ? c : K;
    {
      float4* dst = records + 2u * ((size_t)l * K);
      if (nf > 0u) { dst[0] = a0; dst[1] = b0; }
      if (nf > 1u) { dst[2] = a1; dst[3] = b1; }
      if (nf > 2u) { dst[4] = a2; dst[5] = b2; }
      if (nf > 3u) { dst[6] = a3; dst[7] = b3; }
    }
    const uint32_t e = c - nf;                                // beyond the fixed slots
    uint32_t take = c;
    if (__syncthreads_or(e ? 1 : 0)) {
      uint32_t total;
      const uint32_t off = blockScanExclusive(e, carry, sWave, &total);
      const uint32_t te = (off + e <= capX) ? e : (off < capX ? capX - off : 0u);
      over += e - te;                                         // the message is full
      take = nf + te;
      const uint32_t inBin = take < kBinCap ? take : kBinCap;
      const float4* src = d.bins + 2u * ((size_t)cell * kBinCap);
      float4* dst = extraRec + 2u * (size_t)off;
      for (uint32_t r = nf; r < inBin; ++r) { dst[2u * (r - nf)] = src[2u * r]; dst[2u * (r - nf) + 1u] = src[2u * r + 1u]; }
      sweepOverflow(cell, off, take);
      carry = total;
    }
    if (over) atomicAdd(&d.counters[ctr + kCtrBorderLost], over);
    if (l < L) msg[kBorderHeader + l] = take;
  }
  if (threadIdx.x == 0) { msg[0] = carry < capX ? carry : capX; msg[1] = carry > capX ? 1u : 0u; }

  // ---- big-box section: this tile's big boxes that reach the neighbour's owned region (its core, unbounded on
  // the sides where the world ends) within kBigReach sectors
  uint32_t* big = msg + borderBinWords(dir, p.binSX - 2u, p.binSZ - 2u, p.borderRecs);
  __shared__ uint32_t bigCount, bigLost;
  if (threadIdx.x == 0) { bigCount = 0u; bigLost = 0u; }
  __syncthreads();
  if (p.tilesX) {
    const float SX = (float)(p.binSX - 2u), SZ = (float)(p.binSZ - 2u);
    const float inf = INFINITY;
    auto reaches = [&](float bx0, float bx1, float bz0, float bz1, int ox, int oz) {
      // tile at offset (ox, oz): core [1 + ox*SX, SX + ox*SX]; outer sides of the world are open
      const int tx = (int)p.tileX + ox, tz = (int)p.tileZ + oz;
      const float x0 = tx == 0 ? -inf : 1.0f + (float)ox * SX - kBigReach, x1 = tx == (int)p.tilesX - 1 ? inf : SX + (float)ox * SX + kBigReach;
      const float z0 = tz == 0 ? -inf : 1.0f + (float)oz * SZ - kBigReach, z1 = tz == (int)p.tilesZ - 1 ? inf : SZ + (float)oz * SZ + kBigReach;
      return bx1 >= x0 && bx0 <= x1 && bz1 >= z0 && bz0 <= z1;
    };
    const bool firstDir = (p.neighbourMask & ((1u << dir) - 1u)) == 0u;      // one workgroup also looks for boxes out of reach
    for (uint32_t b = threadIdx.x; b < nLocal; b += kTile) {
      const float4 lo = d.bigList[2u * (size_t)b], hi = d.bigList[2u * (size_t)b + 1u];
      const float bx0 = floorf(lo.x * p.invSector) - p.binOx, bx1 = floorf(hi.x * p.invSector) - p.binOx;
      const float bz0 = floorf(lo.z * p.invSector) - p.binOz, bz1 = floorf(hi.z * p.invSector) - p.binOz;
      if (reaches(bx0, bx1, bz0, bz1, dx, dz)) {
        const uint32_t slot = atomicAdd(&bigCount, 1u);
        if (slot < kBorderBigCap) { float4* o = reinterpret_cast<float4*>(big + 2) + 2u * slot; o[0] = lo; o[1] = hi; }
        else atomicOr(&bigLost, 1u);
      }
      if (firstDir) {
        bool far = false;
        for (int tz = 0; tz < (int)p.tilesZ; ++tz)
          for (int tx = 0; tx < (int)p.tilesX; ++tx) {
            const int ox = tx - (int)p.tileX, oz = tz - (int)p.tileZ;
            if (ox >= -1 && ox <= 1 && oz >= -1 && oz <= 1) continue;
            far = far || reaches(bx0, bx1, bz0, bz1, ox, oz);
          }
        if (far) atomicAdd(&d.counters[ctr + kCtrBorderLost], 1u);
      }
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    big[0] = bigCount < kBorderBigCap ? bigCount : kBorderBigCap; big[1] = bigLost;
    d.counters[ctr + kCtrBigLocal] = nLocal;
  }
  // ---- halo section (traffic sensors on a tiled world): the records of this tile's CORE-EDGE sectors along this side, cell l in
  // its own kBinCap slots -- no scan, every load independent of every count.  A wave takes a cell at a time (the records of a
  // bin are one coalesced read); what a crowded sector keeps in the overflow list does not travel and is counted.
  if (p.halo) {
    uint32_t* halo = big + kBorderBigWords;
    float4* hrec = reinterpret_cast<float4*>(halo + L);
    const uint32_t wv = threadIdx.x >> 6;
    for (uint32_t l = wv; l < L; l += kTile / 64u) {
      // the core-edge cell next to ring cell l of this side: where a message from that neighbour lands (landingCell); the end
      // cells of a side message lie outside the core
      const bool core = (dx != 0 && dz != 0) || (l >= 1u && l + 2u <= L);
      uint32_t cnt = 0;
      if (core) {
        const uint32_t cell = landingCell(p, dx, dz, l);
        cnt = d.binCount[cell];
        if (cnt > kBinCap) { if (lane == 0u) atomicAdd(&d.counters[ctr + kCtrBorderLost], cnt - kBinCap); cnt = kBinCap; }
        if (lane < cnt) {
          const float4* src = d.bins + 2u * ((size_t)cell * kBinCap + lane);
          hrec[2u * ((size_t)l * kBinCap + lane)] = src[0]; hrec[2u * ((size_t)l * kBinCap + lane) + 1u] = src[1];
        }
      }
      if (lane == 0u) halo[l] = cnt;
    }
  }
}

__global__ __launch_bounds__(kTile) void k_border_pack(const DeviceState d, const TickParams p)
{
  borderPackBody(d, p, blockIdx.x);
}

// split flow: the compaction role and the eight pack workgroups both depend only on the fused kernel -- one launch
__global__ __launch_bounds__(kTile) void k_compact_pack(const DeviceState d, const TickParams p, uint32_t compactBlocks, uint32_t group)
{
  __shared__ uint32_t scratch[kTile / 64];
  __shared__ uint32_t moved[kMaxSpanWords];
  __shared__ uint32_t words[kCompactLdsWords];
  // (the eight pack workgroups take the first indices: theirs is the longer chain of round trips, and the exchange waits for them)
  if (blockIdx.x == 8u && (p.flags & kFlagDeferredReset)) resetParity(d, p.resetParity);      // pipelined tiles: next tick's counters
  if (blockIdx.x < 8u) borderPackBody(d, p, blockIdx.x);
  else compactBody<false>(d, p, blockIdx.x - 8u, compactBlocks, group, scratch, moved, words);
}

__global__ __launch_bounds__(kTile) void k_border_merge(const DeviceState d, const TickParams p)
{
  __shared__ uint32_t sWave[4], sOff[1];
  const uint32_t dir = blockIdx.x;                       // the neighbour in direction dir sent borderRecv[dir]
  if (!((p.neighbourMask >> dir) & 1u) || !d.borderRecv[dir]) return;
  int dx, dz; borderDir(dir, dx, dz);
  const uint32_t L = borderLen(dir, p.binSX - 2u, p.binSZ - 2u);
  const uint32_t* msg = d.borderRecv[dir];
  const uint32_t ctr = kCtrPar + 8u * p.parity;
  if (threadIdx.x == 0 && msg[1]) atomicAdd(&d.counters[ctr + kCtrBorderLost], 1u);   // sender ran out of message space
  const float4* records = reinterpret_cast<const float4*>(msg + kBorderHeader + L);
  // (requested up front, used at the end: the kernel is a chain of dependent round trips)
  const uint32_t* big = msg + borderBinWords(dir, p.binSX - 2u, p.binSZ - 2u, p.borderRecs);
  const uint32_t bigHead0 = big[0], bigHead1 = big[1];
  const uint32_t lane = threadIdx.x & 63u;
  // One record to its place: slot `slot` of the sector's bin, or entry q of the sector overflow list when the bin is full.
  auto place = [&](uint32_t sector, uint32_t slot, uint32_t q, const float4& lo, const float4& hi) __attribute__((always_inline)) {
    if (slot < kBinCap) {
      float4* dst = d.bins + 2u * ((size_t)sector * kBinCap + slot);
      dst[0] = lo; dst[1] = hi;
    } else if (q < p.ovfCap) {
      d.spill[2u * (size_t)q] = lo; d.spill[2u * (size_t)q + 1u] = hi; d.spillSector[q] = sector;
    } else atomicAdd(&d.counters[ctr + kCtrBorderLost], 1u);
  };
  // reserve room for c records of one landing sector: bin slots first, the rest in the sector overflow list (ONE reservation
  // each, and the sector's slice bounds follow: two atomics per sector, not per record)
  auto reserve = [&](uint32_t sector, uint32_t c, uint32_t& slot0, uint32_t& q0) __attribute__((always_inline)) {
    slot0 = atomicAdd(&d.binCount[sector], c);
    if (slot0 <= kBinCap && slot0 + c > kBinCap) queueCrowded(d, p, sector);         // (slot 64 is among the ones this reservation got)
    const uint32_t inBin = slot0 < kBinCap ? (c < kBinCap - slot0 ? c : kBinCap - slot0) : 0u;
    const uint32_t nOver = c - inBin;
    q0 = 0;
    if (nOver) {
      q0 = atomicAdd(&d.counters[ctr + kCtrSpill], nOver);
      if (q0 < p.ovfCap) { atomicMin(&d.ovfLo[sector], q0); atomicMax(&d.ovfHi[sector], (q0 + nOver < p.ovfCap) ? q0 + nOver : p.ovfCap); }
    }
    // (record r goes to slot0 + r while that is below the bin's capacity, else to list entry q0 + (slot0 + r - max(slot0, 64)))
  };
  constexpr uint32_t kSerial = 24;                         // a sector's records up to this many are landed by its own thread (the wave-wide path costs a few round trips per sector: it is for crowded sectors, not for bins with five records)
  const uint32_t recCap = borderRecCap(L, p.borderRecs);
  const uint32_t K = borderFixedSlots(L, p.borderRecs);
  const float4* extraRec = records + 2u * ((size_t)L * K);
  const uint32_t capX = recCap - L * K;
  const uint32_t extras = msg[0];                          // records in the spill-over area: none on the usual tick -> no scan
  // cell l's records: the first K in its fixed slots (already in registers: fa/fb), the rest at `off` of the spill-over area
  auto landBin = [&](uint32_t l, uint32_t off, uint32_t c, const float4& a0, const float4& b0, const float4& a1, const float4& b1,
                     const float4& a2, const float4& b2, const float4& a3, const float4& b3) __attribute__((always_inline)) {
    // the sender's ring cell l on its side (-dx,-dz) is this tile's cell l along its own side (dx,dz)
    const uint32_t sector = landingCell(p, dx, dz, l);
    uint32_t slot0, q0; reserve(sector, c, slot0, q0);
    const uint32_t firstOver = slot0 < kBinCap ? kBinCap : slot0;
    uint32_t lay = 0;
    const uint32_t nf = c < K ? c : K;
    if (nf > 0u) { lay |= __float_as_uint(a0.w); place(sector, slot0, q0 + (slot0 - firstOver), a0, b0); }
    if (nf > 1u) { lay |= __float_as_uint(a1.w); place(sector, slot0 + 1u, q0 + (slot0 + 1u - firstOver), a1, b1); }
    if (nf > 2u) { lay |= __float_as_uint(a2.w); place(sector, slot0 + 2u, q0 + (slot0 + 2u - firstOver), a2, b2); }
    if (nf > 3u) { lay |= __float_as_uint(a3.w); place(sector, slot0 + 3u, q0 + (slot0 + 3u - firstOver), a3, b3); }
    const float4* src = extraRec + 2u * (size_t)off;
    for (uint32_t r = K; r < c; ++r) {
      const float4 lo = src[2u * (r - K)], hi = src[2u * (r - K) + 1u];
      lay |= __float_as_uint(lo.w);
      place(sector, slot0 + r, q0 + (slot0 + r - firstOver), lo, hi);
    }
    if (lay) atomicOr(&d.binLayers[sector], lay);
    // the declared layer vocabulary is a contract between the tiles (scTickSetWorldLayers): what arrives outside it is counted, never silent
    if (p.vocabKnown && (lay & ~p.vocab)) atomicAdd(&d.lazyCtl[1u + kMaxParity + p.parity], 1u);
  };
  // crowded landing sectors: the wave lands them one after the other, a record per lane and round
  auto landCrowded = [&](uint32_t l, uint32_t off, uint32_t c, bool mine) __attribute__((always_inline)) {
    unsigned long long todo = ballot64(mine);
    while (todo) {
      const int srcLane = __ffsll((long long)todo) - 1;
      todo &= todo - 1ull;
      const uint32_t wl = __shfl(l, srcLane, 64), wOff = __shfl(off, srcLane, 64), wc = __shfl(c, srcLane, 64);
      const uint32_t sector = landingCell(p, dx, dz, wl);
      uint32_t slot0 = 0, q0 = 0;
      if (lane == 0) reserve(sector, wc, slot0, q0);
      slot0 = __shfl(slot0, 0, 64); q0 = __shfl(q0, 0, 64);
      const uint32_t firstOver = slot0 < kBinCap ? kBinCap : slot0;
      const float4* fsrc = records + 2u * ((size_t)wl * K);
      const float4* xsrc = extraRec + 2u * (size_t)wOff;
      uint32_t lay = 0;
      for (uint32_t r = lane; r < wc; r += 64u) {
        const float4* s = r < K ? fsrc + 2u * r : xsrc + 2u * (r - K);
        const float4 lo = s[0], hi = s[1];
        lay |= __float_as_uint(lo.w);
        place(sector, slot0 + r, q0 + (slot0 + r - firstOver), lo, hi);
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) lay |= (uint32_t)__shfl_xor((int)lay, o, 64);
      if (lane == 0 && lay) atomicOr(&d.binLayers[sector], lay);
      if (lane == 0 && p.vocabKnown && (lay & ~p.vocab)) atomicAdd(&d.lazyCtl[1u + kMaxParity + p.parity], 1u);
    }
  };
  // Two passes over the message's cells: the first lands the usual ones -- up to kSerial records, each cell by its own thread --,
  // the second, which exists only when some cell holds more, lands those a wave at a time.  (kSerial was 4 at first: ring bins
  // of the usual world hold 0-3 records but now and then five or six, and every such bin then cost its wave a few round trips of
  // the wave-wide path -- the kernel took 23 instead of 10 us, the in-order tile step 89 instead of 73.)
  // The count, the header and the cell's fixed slots are requested together (round 4); the slot reservation follows the count,
  // the stores follow both: three round trips on the usual tick, where round 3 had count -> scan -> record -> store.
  uint32_t carry = 0;
  bool anyCrowded = false;
  for (uint32_t base = 0; base < L; base += kTile) {
    const uint32_t l = base + threadIdx.x;
    // (counts are what a neighbour wrote: held to what a sector can hold whatever arrives)
    uint32_t c = 0;
    static_assert(kBorderFixed == 4, "the fixed slots are spelled out (arrays of float4 end up in scratch)");
    const float4 zero4 = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    float4 a0 = zero4, b0 = zero4, a1 = zero4, b1 = zero4, a2 = zero4, b2 = zero4, a3 = zero4, b3 = zero4;
    if (l < L) {
      c = min(msg[kBorderHeader + l], kSectorRecMax);
      const float4* r = records + 2u * ((size_t)l * K);
      if (K > 0u) { a0 = r[0]; b0 = r[1]; }
      if (K > 1u) { a1 = r[2]; b1 = r[3]; }
      if (K > 2u) { a2 = r[4]; b2 = r[5]; }
      if (K > 3u) { a3 = r[6]; b3 = r[7]; }
    }
    const uint32_t e = c > K ? c - K : 0u;
    uint32_t off = 0;
    if (extras) { uint32_t total; off = blockScanExclusive(e, carry, sWave, &total); carry = total; }
    const bool ok = c && (e == 0u || (extras && off + e <= capX));
    if (ok && c <= kSerial) landBin(l, off, c, a0, b0, a1, b1, a2, b2, a3, b3);
    anyCrowded = anyCrowded || (ok && c > kSerial);
  }
  if (__syncthreads_or(anyCrowded ? 1 : 0)) {
    carry = 0;
    for (uint32_t base = 0; base < L; base += kTile) {
      const uint32_t l = base + threadIdx.x;
      const uint32_t c = l < L ? min(msg[kBorderHeader + l], kSectorRecMax) : 0u;
      const uint32_t e = c > K ? c - K : 0u;
      uint32_t total;
      const uint32_t off = blockScanExclusive(e, carry, sWave, &total);
      const bool ok = c && (e == 0u || off + e <= capX);
      landCrowded(l, off, c, ok && c > kSerial);
      carry = total;
    }
  }
  // the neighbour's big boxes that reach this tile join the big list behind this tile's own
  const uint32_t m = bigHead0 < kBorderBigCap ? bigHead0 : kBorderBigCap;
  if (threadIdx.x == 0) {
    sOff[0] = m ? atomicAdd(&d.counters[kCtrPar + 8u * p.parity + kCtrBig], m) : 0u;
    if (bigHead1) atomicAdd(&d.counters[kCtrPar + 8u * p.parity + kCtrBorderLost], 1u);
  }
  __syncthreads();
  const uint32_t at = sOff[0];
  const float4* src = reinterpret_cast<const float4*>(big + 2);
  for (uint32_t r = threadIdx.x; r < 2u * m; r += kTile) {
    if (at + r / 2u < p.bigCap) d.bigList[2u * (size_t)at + r] = src[r];
    if (p.vocabKnown && !(r & 1u) && (__float_as_uint(src[r].w) & ~p.vocab)) atomicAdd(&d.lazyCtl[1u + kMaxParity + p.parity], 1u);      // (big boxes too)
  }
  // ---- halo section: the neighbour's core-edge records land in THIS tile's ring cells on that side, behind the tile's own boxes that
  // reach out there.  Ring sectors on a side with a neighbour are not this tile's to search (the pair search skips them), the border
  // pack of the next tick runs before anything lands again, and the pair search leaves every bin's counter at its remembered value:
  // the records exist for this tick's ray queries and obstacle rays only.
  if (p.halo) {
    const uint32_t* halo = big + kBorderBigWords;
    const float4* hrec = reinterpret_cast<const float4*>(halo + L);
    const uint32_t wv = threadIdx.x >> 6;
    for (uint32_t l = wv; l < L; l += kTile / 64u) {
      const uint32_t cnt = min(halo[l], kBinCap);
      if (!cnt) continue;
      bool send;
      const uint32_t cell = ringCell(p, dx, dz, l, &send);
      uint32_t slot0 = 0;
      if (lane == 0u) slot0 = atomicAdd(&d.binCount[cell], cnt);
      slot0 = __shfl(slot0, 0, 64);
      if (lane < cnt) {
        if (slot0 + lane < kBinCap) {
          float4* dst = d.bins + 2u * ((size_t)cell * kBinCap + slot0 + lane);
          dst[0] = hrec[2u * ((size_t)l * kBinCap + lane)]; dst[1] = hrec[2u * ((size_t)l * kBinCap + lane) + 1u];
        }
      }
      if (lane == 0u && slot0 + cnt > kBinCap) atomicAdd(&d.counters[ctr + kCtrBorderLost], slot0 + cnt - (slot0 < kBinCap ? kBinCap : slot0));
    }
  }
}

// ------------------------------------------------------------------------------------------
// Small producers / accessors
// ------------------------------------------------------------------------------------------
// The frame's frustum lives in device memory (the kernels read it with scalar loads at the point of use); the host
// hands new planes over as kernel arguments, in stream order with the ticks.
__global__ void k_set_frustum(float* __restrict__ dst, const Frustum6 fr)
{
  if (threadIdx.x < 24u) dst[threadIdx.x] = fr.p[threadIdx.x >> 2][threadIdx.x & 3u];
}

// Four consecutive entities per lane (16-byte loads and stores): 8 lanes cover one dirty word.
__global__ __launch_bounds__(kTile) void k_nudge_roots_x(const DeviceState d, uint32_t n, float dx)
{
  const uint32_t t = blockIdx.x * kTile + threadIdx.x;
  const uint32_t i0 = 4u * t;                       // capacity is padded to kTile entities: whole quads are addressable
  const uint32_t lane = threadIdx.x & 63u;
  const bool in = i0 < n;
  // one memory round trip: links, positions and the dirty word are requested together
  uint4 lk = make_uint4(0, 0, 0, 0); float4 x = make_float4(0, 0, 0, 0);
  if (in) { lk = *reinterpret_cast<const uint4*>(d.link + i0); x = *reinterpret_cast<const float4*>(d.px + i0); }
  uint32_t dw = 0;
  if (in && (lane & 7u) == 0u) dw = d.dirty[i0 >> 5];
  auto isRoot = [&](uint32_t l, uint32_t k) { return i0 + k < n && (l & kParentMask) == kNoParent && linkDepth(l) != kUnreachable; };
  const bool r0 = isRoot(lk.x, 0), r1 = isRoot(lk.y, 1), r2 = isRoot(lk.z, 2), r3 = isRoot(lk.w, 3);
  if (r0) x.x = x.x + dx;
  if (r1) x.y = x.y + dx;
  if (r2) x.z = x.z + dx;
  if (r3) x.w = x.w + dx;
  if (r0 | r1 | r2 | r3) *reinterpret_cast<float4*>(d.px + i0) = x;
  // the eight lanes of a dirty word combine their four bits
  uint32_t bits = ((uint32_t)r0 | (uint32_t)r1 << 1 | (uint32_t)r2 << 2 | (uint32_t)r3 << 3) << (4u * (lane & 7u));
  bits |= (uint32_t)__shfl_xor((int)bits, 1, 64);
  bits |= (uint32_t)__shfl_xor((int)bits, 2, 64);
  bits |= (uint32_t)__shfl_xor((int)bits, 4, 64);
  if (in && (lane & 7u) == 0u && bits) d.dirty[i0 >> 5] = dw | bits;
}

// Read-back / debug only (SC_TICK_DENSE_AABBS): per-entity world AABBs in dense order, from the stored
// matrices and bounds -- the same worldAabb() the binning uses, kept out of the hot kernel.
__global__ __launch_bounds__(kTile) void k_dense_aabbs(const DeviceState d, uint32_t n)
{
  const uint32_t i = blockIdx.x * kTile + threadIdx.x;
  if (i >= n) return;
  const uint32_t lk = ldU(d, kLINK, i);
  float mn[3] = { INFINITY, INFINITY, INFINITY }, mx[3] = { -INFINITY, -INFINITY, -INFINITY };
  if (lk & kHasBounds) { const Aff M = loadRows(d, i); const BoundsCE b = loadBounds(d, i); worldAabb(M, b, mn, mx); }
  d.aabbMin[i] = make_float4(mn[0], mn[1], mn[2], 0.0f);
  d.aabbMax[i] = make_float4(mx[0], mx[1], mx[2], 0.0f);
}

__global__ __launch_bounds__(kTile) void k_advance_movers(const DeviceState d, uint32_t n, float dt, float smooth, float mult)
{
  const uint32_t i = blockIdx.x * kTile + threadIdx.x;
  markDirtyWave(d, i, n, moverPosition(d, i, n, dt, smooth, mult));
}

// TrafficLODSystem's tier selection, per vehicle (src/engine/traffic/sc_traffic_lod.cpp:303-307 distance to the player in
// the xz plane, :323-353 the hysteresis between the tiers): writes the desired tier, counts the tiers, and lists the few
// agents that want the Physics or Kinematic tier (the host applies the caps :355-417 to that list).
__global__ __launch_bounds__(kTile) void k_traffic_tiers(const DeviceState d, uint32_t n, const TierParams tp)
{
  const uint32_t i = blockIdx.x * kTile + threadIdx.x;
  const bool agent = i < n && d.moverKind[i] == kMoverTraffic;
  uint32_t want = kTierOnRails; float dist = 0.0f;
  if (agent) {
    const float dx = d.px[i] - tp.px, dz = d.pz[i] - tp.pz;
    dist = sqrtf(dx * dx + dz * dz);
    const uint32_t cur = d.aMode[i];
    if (cur == kTierPhysics) want = (dist > tp.aExit) ? ((dist < tp.bEnter) ? kTierKinematic : kTierOnRails) : kTierPhysics;
    else if (cur == kTierKinematic) want = (dist < tp.aEnter) ? kTierPhysics : ((dist > tp.bExit) ? kTierOnRails : kTierKinematic);
    else want = (dist < tp.aEnter) ? kTierPhysics : ((dist < tp.bEnter) ? kTierKinematic : kTierOnRails);
    d.aDesired[i] = want;
  }
  const unsigned long long mp = ballot64(agent && want == kTierPhysics), mk = ballot64(agent && want == kTierKinematic), mr = ballot64(agent && want == kTierOnRails);
  if ((threadIdx.x & 63u) == 0u) {
    if (mp) atomicAdd(&d.tierCounts[0], (uint32_t)__popcll(mp));
    if (mk) atomicAdd(&d.tierCounts[1], (uint32_t)__popcll(mk));
    if (mr) atomicAdd(&d.tierCounts[2], (uint32_t)__popcll(mr));
  }
  if (agent && want != kTierOnRails) {
    const uint32_t slot = atomicAdd(&d.tierCounts[3], 1u);
    // (a distance is never negative: its sign bit carries which of the two tiers is wanted)
    if (slot < kTierNearCap) d.tierNear[slot] = make_uint2(i, __float_as_uint(dist) | (want == kTierKinematic ? 0x80000000u : 0u));
  }
}

// The total cap of TrafficLODSystem (sc_traffic_lod.cpp:419-465): when more vehicles exist than maxTrafficVehiclesTotal the surplus
// is despawned -- OnRails vehicles first, then Kinematic, then Physics, the farthest first inside a tier.  The device lists every
// agent with a key that orders exactly so (tier rank in the high word, the bits of its xz distance to the player -- never
// negative, so they order like the value -- in the low one); the host takes the top of that list (scTickSelectTrafficDespawns).
__global__ __launch_bounds__(kTile) void k_traffic_despawn_keys(const DeviceState d, uint32_t n, float px, float pz, uint32_t* __restrict__ count,
                                                                uint32_t* __restrict__ outIdx, unsigned long long* __restrict__ outKey)
{
  const uint32_t i = blockIdx.x * kTile + threadIdx.x;
  const bool agent = i < n && d.moverKind[i] == kMoverTraffic;
  const unsigned long long m = ballot64(agent);
  if (!m) return;
  const uint32_t lane = threadIdx.x & 63u;
  uint32_t base = 0;
  if (lane == 0) base = atomicAdd(count, (uint32_t)__popcll(m));
  base = __shfl(base, 0, 64);
  if (agent) {
    const float dx = d.px[i] - px, dz = d.pz[i] - pz;
    const float dist = sqrtf(dx * dx + dz * dz);                                   // :303-307, as the tier selection
    const uint32_t mode = d.aMode[i];
    const uint32_t rank = mode == kTierOnRails ? 2u : (mode == kTierKinematic ? 1u : 0u);
    const uint32_t at = base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
    outIdx[at] = i;
    outKey[at] = ((unsigned long long)rank << 32) | __float_as_uint(dist);
  }
}

// TrafficVehicle::mode = the desired tier (applyMode, sc_traffic_lod.cpp:486-487), then the handful the caps changed
__global__ __launch_bounds__(kTile) void k_apply_tiers(const DeviceState d, uint32_t n)
{
  const uint32_t i = blockIdx.x * kTile + threadIdx.x;
  if (i < n && d.moverKind[i] == kMoverTraffic) d.aMode[i] = d.aDesired[i];
}
__global__ __launch_bounds__(kTile) void k_patch_tiers(const DeviceState d, const uint2* __restrict__ patches, uint32_t count)
{
  const uint32_t t = blockIdx.x * kTile + threadIdx.x;
  if (t < count) d.aMode[patches[t].x] = patches[t].y;
}

__global__ __launch_bounds__(kTile) void k_set_dirty_range(const DeviceState d, uint32_t first, uint32_t count)
{
  const uint32_t w = (first >> 5) + blockIdx.x * kTile + threadIdx.x;
  const uint32_t last = first + count;            // exclusive
  if ((w << 5) >= last) return;
  const uint32_t lo = w << 5;
  uint32_t mask = 0xFFFFFFFFu;
  if (first > lo) mask &= 0xFFFFFFFFu << (first - lo);
  if (last < lo + 32u) mask &= 0xFFFFFFFFu >> (lo + 32u - last);
  d.dirty[w] |= mask;
}

__global__ __launch_bounds__(kTile) void k_set_dirty_indices(const DeviceState d, const uint32_t* __restrict__ idx, uint32_t count)
{
  const uint32_t t = blockIdx.x * kTile + threadIdx.x;
  if (t < count) atomicOr(&d.dirty[idx[t] >> 5], 1u << (idx[t] & 31u));
}

__global__ __launch_bounds__(kTile) void k_gather_rows(const DeviceState d, const uint32_t* __restrict__ idx, uint32_t count, float* __restrict__ out12)
{
  const uint32_t t = blockIdx.x * kTile + threadIdx.x;
  if (t >= count) return;
  const uint32_t j = idx[t];
  float4* o = reinterpret_cast<float4*>(out12) + 3u * (size_t)t;
  o[0] = d.w0[j]; o[1] = d.w1[j]; o[2] = d.w2[j];
}

// Swap-remove relocations (ComponentPool::remove, sc_ecs.h:240-262, applied to every per-entity array at
// once): entity src[k] moves to slot dst[k].  The host guarantees every src lies at or beyond the new
// entity count and every dst below it, so no slot is both read and written.  One thread per (move, array).
constexpr uint32_t kMoveSlots = 40;     // 22 streams, 3 matrix rows, the dirty bit, 7 mover arrays, 5 traffic-agent arrays, 2 sensor arrays
__global__ __launch_bounds__(kTile) void k_move_entities(const DeviceState d, const uint32_t* __restrict__ src,
                                                         const uint32_t* __restrict__ dst, uint32_t moves)
{
  const uint32_t t = blockIdx.x * kTile + threadIdx.x;
  const uint32_t k = t / kMoveSlots, slot = t % kMoveSlots;
  if (k >= moves) return;
  const uint32_t from = src[k], to = dst[k];
  if (slot < kStreamCount) {
    uint32_t* base = reinterpret_cast<uint32_t*>(const_cast<char*>(d.fslab) + (size_t)slot * d.capBytes);
    base[to] = base[from];
  } else if (slot < kStreamCount + 3u) {
    float4* rows = reinterpret_cast<float4*>(d.rslab + (size_t)(slot - kStreamCount) * d.capBytes16);
    rows[to] = rows[from];
  } else if (slot == kStreamCount + 3u) {
    const bool isDirty = (d.dirty[from >> 5] >> (from & 31u)) & 1u;
    if (isDirty) atomicOr(&d.dirty[to >> 5], 1u << (to & 31u));
    else atomicAnd(&d.dirty[to >> 5], ~(1u << (to & 31u)));
  } else if (d.moverKind && slot < kStreamCount + 11u) {
    uint32_t* arrays[7] = { d.moverKind, reinterpret_cast<uint32_t*>(d.mvx), reinterpret_cast<uint32_t*>(d.mvz),
                            reinterpret_cast<uint32_t*>(d.mlox), reinterpret_cast<uint32_t*>(d.mloz),
                            reinterpret_cast<uint32_t*>(d.mhix), reinterpret_cast<uint32_t*>(d.mhiz) };
    uint32_t* a = arrays[slot - kStreamCount - 4u];
    a[to] = a[from];
  } else if (d.aLane && slot >= kStreamCount + 11u && slot < kStreamCount + 16u) {
    uint32_t* arrays[5] = { d.aLane, reinterpret_cast<uint32_t*>(d.aS), reinterpret_cast<uint32_t*>(d.aSpeed), d.aMode, reinterpret_cast<uint32_t*>(d.aLook) };
    uint32_t* a = arrays[slot - kStreamCount - 11u];
    a[to] = a[from];
  } else if (d.aRayLen && slot >= kStreamCount + 16u && slot < kStreamCount + 18u) {      // per-agent TrafficSensors values travel with their entity
    float* a = slot == kStreamCount + 16u ? d.aRayLen : d.aSafe;
    a[to] = a[from];
  }
}

// Children of a relocated entity: pairs (entity, new parent index); depth and flags are untouched.
__global__ __launch_bounds__(kTile) void k_patch_parents(const DeviceState d, const uint32_t* __restrict__ pairs, uint32_t count)
{
  const uint32_t t = blockIdx.x * kTile + threadIdx.x;
  if (t >= count) return;
  const uint32_t e = pairs[2u * t], q = pairs[2u * t + 1u];
  d.link[e] = (d.link[e] & ~kParentMask) | (q & kParentMask);
}

// RenderPrepStreamingSystem draw emission (sc_world_partition.cpp:1306-1329): the first `budget`
// visible entities, in order, become DrawItem{entity, mesh, material, worldMatrix}.
struct DrawItem80 { uint32_t dense, mesh, material, pad; float model[16]; };
__global__ __launch_bounds__(kTile) void k_emit_draws(const DeviceState d, uint32_t budget, DrawItem80* __restrict__ items)
{
  const uint32_t visible = d.counters[0];
  const uint32_t emitted = (budget > 0 && visible > budget) ? budget : visible;
  if (blockIdx.x == 0 && threadIdx.x == 0) { d.counters[4] = emitted; d.counters[5] = visible - emitted; }
  for (uint32_t t = blockIdx.x * kTile + threadIdx.x; t < emitted; t += gridDim.x * kTile) {
    const uint32_t j = d.visibleIdx[t];
    const float4 a = ldRow(d, 0, j), b = ldRow(d, 1, j), c = ldRow(d, 2, j);
    float4* o = reinterpret_cast<float4*>(&items[t]);
    o[0] = make_float4(__uint_as_float(j), __uint_as_float(d.meshId[j]), __uint_as_float(d.materialId[j]), 0.0f);
    o[1] = make_float4(a.x, b.x, c.x, 0.0f);     // column 0
    o[2] = make_float4(a.y, b.y, c.y, 0.0f);
    o[3] = make_float4(a.z, b.z, c.z, 0.0f);
    o[4] = make_float4(a.w, b.w, c.w, 1.0f);     // translation column
  }
}

// Frame read-back: the frame's counts, the head of the visible list and of the draw list, gathered into one block that a
// single device-to-host copy then takes (the lists themselves are overwritten by the next tick).
__global__ __launch_bounds__(kTile) void k_stage_frame(const DeviceState d, uint32_t* __restrict__ block, uint32_t maxVisible, uint32_t maxDraws,
                                                       const uint4* __restrict__ items, uint32_t drawMode, uint32_t tickLo, uint32_t tickHi)
{
  const uint32_t visible = d.counters[0];
  const uint32_t draws = drawMode == 2u ? d.counters[kCtrDrawsSorted] : (drawMode == 1u ? d.counters[4] : 0u);
  const uint32_t nv = visible < maxVisible ? visible : maxVisible, nd = draws < maxDraws ? draws : maxDraws;
  if (blockIdx.x == 0 && threadIdx.x < kFrameHeaderWords) {
    const uint32_t h[kFrameHeaderWords] = { visible, d.counters[1], d.counters[6], drawMode ? d.counters[4] : 0u, drawMode ? d.counters[5] : 0u,
                                            drawMode == 2u ? d.counters[kCtrDrawsSorted] : 0u, tickLo, tickHi, nv, nd, 0, 0, 0, 0, 0, 0 };
    block[threadIdx.x] = h[threadIdx.x];
  }
  uint32_t* vis = block + kFrameHeaderWords;
  for (uint32_t t = blockIdx.x * kTile + threadIdx.x; t < nv; t += gridDim.x * kTile) vis[t] = d.visibleIdx[t];
  uint4* out = reinterpret_cast<uint4*>(block + kFrameHeaderWords + maxVisible);      // (maxVisible is a multiple of 4: 16-byte aligned)
  for (uint32_t t = blockIdx.x * kTile + threadIdx.x; t < nd * 5u; t += gridDim.x * kTile) out[t] = items[t];
}

// Draw emission and frame staging in one launch (the common resident-mode frame: plain draw order, budget within the
// block): the items are written straight into the read-back block, next to the header and the head of the visible list.
__global__ __launch_bounds__(kTile) void k_emit_draws_staged(const DeviceState d, uint32_t budget, uint32_t* __restrict__ block, uint32_t maxVisible,
                                                             uint32_t tickLo, uint32_t tickHi)
{
  const uint32_t visible = d.counters[0];
  const uint32_t emitted = (budget > 0 && visible > budget) ? budget : visible;      // the host guarantees budget <= the block's draw capacity
  const uint32_t nv = visible < maxVisible ? visible : maxVisible;
  if (blockIdx.x == 0 && threadIdx.x == 0) { d.counters[4] = emitted; d.counters[5] = visible - emitted; }
  if (blockIdx.x == 0 && threadIdx.x < kFrameHeaderWords) {
    const uint32_t h[kFrameHeaderWords] = { visible, d.counters[1], d.counters[6], emitted, visible - emitted, 0u, tickLo, tickHi, nv, emitted, 0, 0, 0, 0, 0, 0 };
    block[threadIdx.x] = h[threadIdx.x];
  }
  uint32_t* vis = block + kFrameHeaderWords;
  DrawItem80* items = reinterpret_cast<DrawItem80*>(block + kFrameHeaderWords + maxVisible);
  for (uint32_t t = blockIdx.x * kTile + threadIdx.x; t < (emitted > nv ? emitted : nv); t += gridDim.x * kTile) {
    const uint32_t j = d.visibleIdx[t];
    if (t < nv) vis[t] = j;
    if (t < emitted) {
      const float4 a = ldRow(d, 0, j), b = ldRow(d, 1, j), c = ldRow(d, 2, j);
      float4* o = reinterpret_cast<float4*>(&items[t]);
      o[0] = make_float4(__uint_as_float(j), __uint_as_float(d.meshId[j]), __uint_as_float(d.materialId[j]), 0.0f);
      o[1] = make_float4(a.x, b.x, c.x, 0.0f);
      o[2] = make_float4(a.y, b.y, c.y, 0.0f);
      o[3] = make_float4(a.z, b.z, c.z, 0.0f);
      o[4] = make_float4(a.w, b.w, c.w, 1.0f);
    }
  }
}

// Learn tick, right behind the fused kernel (before the level kernels and the border merge add records of their own): the
// slots handed out so far are the remembered ones.
__global__ __launch_bounds__(kTile) void k_snapshot_home(const DeviceState d, uint32_t sectors)
{
  const uint32_t s = blockIdx.x * kTile + threadIdx.x;
  if (s >= sectors) return;
  const uint32_t c = d.binCount[s];
  d.homeCount[s] = c < kBinCap ? c : kBinCap;
  d.homeLayers[s] = d.binLayers[s];
}

// Ordered home slots (round 4).  The pair search spent most of its instructions per sector on finding out, on every tick, which of
// a bin's records can collide at all -- filter masks against the bin's layer summary, ballots, ranks, a re-ordered copy in LDS -- although
// for the records that keep their slots from one learn tick to the next the answer never changes.  So the learn tick sorts it out once:
// one wave per bin re-numbers the reserved records so that those that pass the group/mask filter against their own kind (dynamic
// bodies: "cast" records) hold slots [0, D) and everything else follows, MOVES the records accordingly (a record that does not change
// is not rewritten on later ticks: TickParams::cleanStay), notes D and whether the rest can meet each other (homeCast), and leaves
// the permutation in homePerm for k_home_flags, which re-numbers the owners' homeB bytes through it.  A bin that holds nothing but
// its reserved records on a later tick -- count == homeCount, the usual case -- is then searched as it lies ("fast sectors" in pairsBody).
__global__ __launch_bounds__(kTile) void k_order_home(const DeviceState d, uint32_t sectors)
{
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t s = blockIdx.x * (kTile / 64u) + (threadIdx.x >> 6);
  if (s >= sectors) return;
  const uint32_t hc = min(d.homeCount[s], kBinCap);
  float4 lo, hi; nullRecord(lo, hi);
  float4* bin = d.bins + 2u * ((size_t)s * kBinCap);
  if (lane < hc) { lo = bin[2u * lane]; hi = bin[2u * lane + 1u]; }
  const uint32_t lay = lane < hc ? __float_as_uint(lo.w) : 0u;
  const bool self = lane < hc && ((lay & 0xFFFFu) & (lay >> 16)) != 0u;       // filterPass(lay, lay)
  const unsigned long long castMask = ballot64(self), restMask = ballot64(lane < hc && !self);
  uint32_t rest = (lane < hc && !self) ? lay : 0u;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) rest |= (uint32_t)__shfl_xor((int)rest, o, 64);
  const bool cross = ((rest & 0xFFFFu) & (rest >> 16)) != 0u;                // two of the other records may pass against each other
  const uint32_t D = (uint32_t)__popcll(castMask);
  const unsigned long long below = (1ull << lane) - 1ull;
  const uint32_t slot = self ? (uint32_t)__popcll(castMask & below) : D + (uint32_t)__popcll(restMask & below);
  if (lane < hc) d.homePerm[(size_t)s * kBinCap + lane] = (uint8_t)slot;
  // (every lane's record is in registers before any lane stores: the stores wait for the loads of the whole wave)
  if (ballot64(lane < hc && slot != lane)) { if (lane < hc) { bin[2u * slot] = lo; bin[2u * slot + 1u] = hi; } }
  if (lane == 0) d.homeCast[s] = D | (cross ? 0u : kCastFast);
}

// ... and every remembered slot learns whether its bin is one that is written on every tick (lazy records): a bin whose
// reserved records can pass the group/mask filter against each other, or a ring sector (the border pack reads those).
// With the world's layer VOCABULARY declared (scTickSetWorldLayers) and the pair half pipelined, "written on every tick" means:
// some record that can exist anywhere in the world could pass the filter against one of the bin's own -- the bins that fail THAT
// test are never read by anybody, on any tick, so they need no rebuild either (which a pipelined pair half could not do: the
// owners' matrices are the next tick's by then).
__global__ __launch_bounds__(kTile) void k_home_flags(const DeviceState d, uint32_t n, uint32_t binSX, uint32_t binSZ, uint32_t vocabMode, uint32_t vocab, uint32_t ordered)
{
  const uint32_t i = blockIdx.x * kTile + threadIdx.x;
  if (i >= n) return;
  const uint32_t hA = d.homeA[i];
  if (hA == kNoHome) return;
  uint32_t hB = d.homeB[i];
#pragma unroll
  for (uint32_t k = 0; k < 4u; ++k) {
    const uint32_t byte = (hB >> (8u * k)) & 0xFFu;
    if (byte == kNoSlot) continue;
    const uint32_t sec = hA + (k & 1u) + (k >> 1) * binSX;
    // the slot the record was moved to when its bin was put in order (k_order_home)
    // (not on a world whose bins were left in the order of arrival: nothing can pair there, or the fast sectors are switched off)
    if (ordered) hB = (hB & ~(0xFFu << (8u * k))) | ((uint32_t)d.homePerm[(size_t)sec * kBinCap + (byte & kSlotMask)] << (8u * k));
    const uint32_t H = d.homeLayers[sec];
    if (binWrittenEveryTick(vocabMode ? layersThatCanMeet(H, vocab) : H, sec, binSX, binSZ)) hB |= kSlotAlways << (8u * k);
  }
  d.homeB[i] = hB;
}

// ------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------
void launchSnapshotHome(const DeviceState& d, uint32_t sectors, uint32_t n, uint32_t binSX, uint32_t binSZ, uint32_t vocabMode, uint32_t vocab, uint32_t ordered, hipStream_t s)
{
  if (!sectors) return;
  hipLaunchKernelGGL(k_snapshot_home, dim3((sectors + kTile - 1) / kTile), dim3(kTile), 0, s, d, sectors);
  // (ordered = this learn period's pair search takes fast sectors, TickParams::fastPairs: only then are the bins put in cast-first order)
  if (ordered) hipLaunchKernelGGL(k_order_home, dim3((sectors + kTile / 64u - 1) / (kTile / 64u)), dim3(kTile), 0, s, d, sectors);
  else if (d.homeCast) (void)hipMemsetAsync(d.homeCast, 0, (size_t)sectors * sizeof(uint32_t), s);      // no bin is a fast sector until a learn tick orders it (should the pair search ask in between)
  if (n) hipLaunchKernelGGL(k_home_flags, dim3((n + kTile - 1) / kTile), dim3(kTile), 0, s, d, n, binSX, binSZ, vocabMode, vocab, ordered);
}
// `done` (may be null): recorded by the dispatch itself -- the event the copy stream waits for, without a marker packet
void launchEmitDrawsStaged(const DeviceState& d, uint32_t budget, uint32_t* block, uint32_t maxVisible, uint64_t tick, hipStream_t s, hipEvent_t done)
{
  const uint32_t work = std::max(maxVisible, budget);
  const uint32_t blocks = std::max(1u, std::min((work + kTile - 1) / kTile, 64u));
  if (done) hipExtLaunchKernelGGL(k_emit_draws_staged, dim3(blocks), dim3(kTile), 0, s, nullptr, done, 0, d, budget, block, maxVisible, (uint32_t)tick, (uint32_t)(tick >> 32));
  else hipLaunchKernelGGL(k_emit_draws_staged, dim3(blocks), dim3(kTile), 0, s, d, budget, block, maxVisible, (uint32_t)tick, (uint32_t)(tick >> 32));
}
void launchStageFrame(const DeviceState& d, uint32_t* block, uint32_t maxVisible, uint32_t maxDraws, const void* items, uint32_t drawMode,
                      uint64_t tick, hipStream_t s, hipEvent_t done)
{
  const uint32_t work = std::max(maxVisible, maxDraws * 5u);
  const uint32_t blocks = std::max(1u, std::min((work + kTile - 1) / kTile, 64u));
  if (done) hipExtLaunchKernelGGL(k_stage_frame, dim3(blocks), dim3(kTile), 0, s, nullptr, done, 0, d, block, maxVisible, maxDraws, (const uint4*)items, drawMode,
                                  (uint32_t)tick, (uint32_t)(tick >> 32));
  else hipLaunchKernelGGL(k_stage_frame, dim3(blocks), dim3(kTile), 0, s, d, block, maxVisible, maxDraws, (const uint4*)items, drawMode,
                          (uint32_t)tick, (uint32_t)(tick >> 32));
}
// evA / evB (both or neither): events that take the kernel's own begin / end timestamps (hipExtLaunchKernelGGL), so the
// duration bench.py reports is the dispatch's, like the kernel trace's -- not the gap-inclusive span between two
// hipEventRecord calls on the stream
template <bool kCull, bool kAabb, uint32_t kChain, uint32_t kHome>
static void launchHome(const DeviceState& d, const TickParams& p, uint32_t grid, hipStream_t s, hipEvent_t evA, hipEvent_t evB)
{
  if (evA) hipExtLaunchKernelGGL((k_xform_cull<kCull, kAabb, kChain, kHome>), dim3(grid), dim3(kTile), 0, s, evA, evB, 0, d, p);
  else hipLaunchKernelGGL((k_xform_cull<kCull, kAabb, kChain, kHome>), dim3(grid), dim3(kTile), 0, s, d, p);
}
template <bool kCull, bool kAabb, uint32_t kChain>
static void launchOne(const DeviceState& d, const TickParams& p, uint32_t grid, hipStream_t s, hipEvent_t evA, hipEvent_t evB)
{
  // the bins' home slots: an instance per mode, and only where boxes are binned at all
  if (!kAabb || p.homeMode == kHomeOff) launchHome<kCull, kAabb, kChain, kHomeOff>(d, p, grid, s, evA, evB);
  else if (p.homeMode == kHomeUse) launchHome<kCull, kAabb, kChain, kHomeUse>(d, p, grid, s, evA, evB);
  else launchHome<kCull, kAabb, kChain, kHomeLearn>(d, p, grid, s, evA, evB);
}
template <uint32_t kChain>
static void launchXformCullChain(const DeviceState& d, const TickParams& p, uint32_t grid, hipStream_t s, hipEvent_t evA, hipEvent_t evB)
{
  const bool cull = (p.flags & SC_TICK_CULL) != 0, aabb = (p.flags & SC_TICK_BROADPHASE) != 0;
  if (cull && aabb) launchOne<true, true, kChain>(d, p, grid, s, evA, evB);
  else if (cull)    launchOne<true, false, kChain>(d, p, grid, s, evA, evB);
  else if (aabb)    launchOne<false, true, kChain>(d, p, grid, s, evA, evB);
  else              launchOne<false, false, kChain>(d, p, grid, s, evA, evB);
}
void launchXformCull(const DeviceState& d, const TickParams& p, uint32_t grid, hipStream_t s, hipEvent_t evA, hipEvent_t evB)
{
  // A flat world with binning goes through the chain-1 instance: hipcc allocates 74-84 VGPRs for the chain-0 one and
  // 56-72 for chain-1, and the unused level costs nothing at run time -- measured on a flat 1M world: 35.6 vs 44.6 us
  // for the full tick's fused kernel.
  const uint32_t chain = (p.chain == 0u && (p.flags & SC_TICK_BROADPHASE)) ? 1u : p.chain;
  switch (chain) {                                            // deepest level a lane walks: min(world depth, kMaxChain)
    case 0: launchXformCullChain<0>(d, p, grid, s, evA, evB); break;
    case 1: launchXformCullChain<1>(d, p, grid, s, evA, evB); break;
    case 2: launchXformCullChain<2>(d, p, grid, s, evA, evB); break;
    default: launchXformCullChain<3>(d, p, grid, s, evA, evB); break;
  }
}
void launchDeepLevel(const DeviceState& d, const TickParams& p, const uint32_t* list, uint32_t count, hipStream_t s)
{
  if (!count) return;
  hipLaunchKernelGGL(k_deep_level, dim3((count + kTile - 1) / kTile), dim3(kTile), 0, s, d, p, list, count);
}
// spans of the fused kernel per compaction workgroup (SC_TICK_VARIANT bits 4..6 override, tuning)
// Measured at 1M entities (1536 spans): alone, one span per workgroup is fastest (8.2 us; 9.2 / 10.9 with 2 / 4 -- the
// tiles of a workgroup are walked one after the other); sharing the launch with the pair search, two spans per
// workgroup win (10.5 us against 11.3 / 12.2 with 1 / 4): fewer workgroups leave room for the pair role's.
static uint32_t compactGroup(const TickParams& p, uint32_t grid, bool merged)
{
  const uint32_t forced = (p.variant >> 4) & 7u;
  return forced ? forced : ((merged && grid >= 512u) ? 2u : 1u);
}
void launchCompact(const DeviceState& d, const TickParams& p, uint32_t grid, hipStream_t s)
{
  const uint32_t g = compactGroup(p, grid, false);
  hipLaunchKernelGGL(k_compact, dim3((grid + g - 1) / g), dim3(kTile), 0, s, d, p, g);
}
// Pair-role geometry (round 4, second form): four pair workgroups for every CU -- the fifth resident workgroup of each CU is left to the
// compaction role, which shares the launch and comes behind the pair role in index order -- and the run as long as that needs:
// R = ceil(sectors / waves), 17 for 258 x 258 sectors, a run per wave.  (Round 3: runs of 2^k sectors and as many waves as runs, 1041
// workgroups for that world, a few CUs with five.)  What tools/wave_time.py shows of a searching world (-DSC_DIAG_WAVETIME): every wave
// starts within a microsecond, waves with the SAME work end anywhere between 0.1 and 1.0 of the role's span, in the order of their
// indices -- a SIMD serves its oldest wave first -- and the oldest wave of a SIMD gets through its sectors in a tenth of the span: the
// SIMDs are busy throughout, the role is bound by what its four or five waves per SIMD issue together, not by a slow wave.  Hence no
// gain from the evener spread itself (config 5: 34.0 us against 33.8, 37.4 with three workgroups per CU, 34.8 with five; config3dyn 23.2
// / 24.4 / 23.1: profiles/r04/ab_pair_geometry.log); the form stays because it keeps the compaction workgroups resident from the start.
// A world the host knows cannot pair (TickParams::sweepOnly) is a sweep over counters: runs of 64, about one workgroup per CU (the
// per-workgroup prologue is what that role costs).  SC_TICK_VARIANT bits 8+: workgroups (tuning).
uint32_t pairRunFor(uint32_t sectors, uint32_t cus, uint32_t variant, bool sweepOnly)
{
  if (!cus) cus = 256u;
  uint32_t wgs = (variant >> 8) ? (variant >> 8) : (sweepOnly ? cus : 4u * cus);
  wgs = std::min(wgs, kOvfWaves / (kTile / 64u));
  const uint32_t waves = wgs * (kTile / 64u);
  const uint32_t run = (sectors + waves - 1u) / waves;
  return std::max(1u, std::min(run, 64u));
}
static uint32_t pairGridFor(const TickParams& p)
{
  const uint32_t sectors = p.binSX * p.binSZ;
  const uint32_t run = std::max(1u, p.pairRun);
  const uint32_t runs = (sectors + run - 1u) / run;
  // (a wave takes 64 / run runs per round; the grid gives every wave the same number of rounds, within one)
  const uint32_t perRound = 64u / run;
  const uint32_t cap = std::min(((p.variant >> 8) ? (p.variant >> 8) : (p.sweepOnly ? 2u * p.cus : 4u * p.cus)) * (kTile / 64u), kOvfWaves);      // (sweep: runs of 64 are capped, leave room for the odd run)
  uint32_t waves = runs;                                   // a run per wave where the waves suffice (pairRunFor sized the run for that)
  if (runs > std::max(cap, 1u)) {
    const uint32_t rounds = (runs + cap * perRound - 1u) / (cap * perRound);
    waves = (runs + rounds * perRound - 1u) / (rounds * perRound);
  }
  return std::max(1u, (waves + kTile / 64u - 1u) / (kTile / 64u));
}
// `done` (may be null): recorded by the dispatch itself; false = nothing launched (no sectors)
bool launchPairs(const DeviceState& d, const TickParams& p, hipStream_t s, hipEvent_t done)
{
  if (!(p.binSX * p.binSZ)) return false;
  auto kernel = p.lazy == 2u ? k_pairs<true> : k_pairs<false>;
  if (done) hipExtLaunchKernelGGL(kernel, dim3(pairGridFor(p)), dim3(kTile), 0, s, nullptr, done, 0, d, p);
  else hipLaunchKernelGGL(kernel, dim3(pairGridFor(p)), dim3(kTile), 0, s, d, p);
  return true;
}
void launchCompactPairs(const DeviceState& d, const TickParams& p, uint32_t compactGrid, hipStream_t s, hipEvent_t evA, hipEvent_t evB)
{
  const uint32_t pairGrid = pairGridFor(p);
  const uint32_t g = compactGroup(p, compactGrid, true);
  const uint32_t blocks = (compactGrid + g - 1) / g;
  auto kernel = p.emitMode ? k_compact_pairs<true> : k_compact_pairs<false>;      // (draw emission in the compaction role: an instance of its own)
  if (evA || evB) hipExtLaunchKernelGGL(kernel, dim3(blocks + pairGrid), dim3(kTile), 0, s, evA, evB, 0, d, p, blocks, g);
  else hipLaunchKernelGGL(kernel, dim3(blocks + pairGrid), dim3(kTile), 0, s, d, p, blocks, g);
}
void launchGatherPairs(const DeviceState& d, const TickParams& p, uint32_t parity, uint2* dst, uint32_t* total, hipStream_t s)
{
  hipLaunchKernelGGL(k_gather_pairs, dim3(kPairShards), dim3(kTile), 0, s, d, p, parity, dst, total);
}
void launchCompactPack(const DeviceState& d, const TickParams& p, uint32_t grid, hipStream_t s, hipEvent_t done)
{
  const uint32_t g = compactGroup(p, grid, false);
  const uint32_t blocks = (grid + g - 1) / g;
  // `done`: recorded by the dispatch itself (its completion signal) -- no marker packet behind the kernel on the tick queue
  if (done) hipExtLaunchKernelGGL(k_compact_pack, dim3(blocks + 8u), dim3(kTile), 0, s, nullptr, done, 0, d, p, blocks, g);
  else hipLaunchKernelGGL(k_compact_pack, dim3(blocks + 8u), dim3(kTile), 0, s, d, p, blocks, g);
}
void launchBorderPack(const DeviceState& d, const TickParams& p, hipStream_t s)
{
  if (!p.neighbourMask) return;
  hipLaunchKernelGGL(k_border_pack, dim3(8), dim3(kTile), 0, s, d, p);
}
void launchBorderMerge(const DeviceState& d, const TickParams& p, hipStream_t s)
{
  if (!p.neighbourMask) return;
  hipLaunchKernelGGL(k_border_merge, dim3(8), dim3(kTile), 0, s, d, p);
}
void launchNudgeRootsX(const DeviceState& d, uint32_t n, float dx, hipStream_t s)
{
  if (!n) return;
  const uint32_t quads = (n + 3u) / 4u;
  hipLaunchKernelGGL(k_nudge_roots_x, dim3((quads + kTile - 1) / kTile), dim3(kTile), 0, s, d, n, dx);
}
void launchDenseAabbs(const DeviceState& d, uint32_t n, hipStream_t s)
{
  if (!n) return;
  hipLaunchKernelGGL(k_dense_aabbs, dim3((n + kTile - 1) / kTile), dim3(kTile), 0, s, d, n);
}
void launchAdvanceMovers(const DeviceState& d, uint32_t n, float dt, float trafficSmooth, float trafficMult, hipStream_t s)
{
  if (!n || !d.moverKind) return;
  hipLaunchKernelGGL(k_advance_movers, dim3((n + kTile - 1) / kTile), dim3(kTile), 0, s, d, n, dt, trafficSmooth, trafficMult);
}
void launchTrafficTiers(const DeviceState& d, uint32_t n, const TierParams& tp, hipStream_t s)
{
  if (!n || !d.moverKind || !d.aMode) return;
  hipLaunchKernelGGL(k_traffic_tiers, dim3((n + kTile - 1) / kTile), dim3(kTile), 0, s, d, n, tp);
}
void launchTrafficDespawnKeys(const DeviceState& d, uint32_t n, float px, float pz, uint32_t* count, uint32_t* outIdx, unsigned long long* outKey, hipStream_t s)
{
  if (!n || !d.moverKind || !d.aMode) return;
  hipLaunchKernelGGL(k_traffic_despawn_keys, dim3((n + kTile - 1) / kTile), dim3(kTile), 0, s, d, n, px, pz, count, outIdx, outKey);
}
void launchApplyTiers(const DeviceState& d, uint32_t n, const uint2* patches, uint32_t patchCount, hipStream_t s)
{
  if (!n || !d.moverKind || !d.aMode) return;
  hipLaunchKernelGGL(k_apply_tiers, dim3((n + kTile - 1) / kTile), dim3(kTile), 0, s, d, n);
  if (patchCount) hipLaunchKernelGGL(k_patch_tiers, dim3((patchCount + kTile - 1) / kTile), dim3(kTile), 0, s, d, patches, patchCount);
}
void launchSetDirtyRange(const DeviceState& d, uint32_t first, uint32_t count, hipStream_t s)
{
  if (!count) return;
  const uint32_t words = ((first + count + 31u) >> 5) - (first >> 5);
  hipLaunchKernelGGL(k_set_dirty_range, dim3((words + kTile - 1) / kTile), dim3(kTile), 0, s, d, first, count);
}
void launchSetDirtyIndices(const DeviceState& d, const uint32_t* idx, uint32_t count, hipStream_t s)
{
  if (!count) return;
  hipLaunchKernelGGL(k_set_dirty_indices, dim3((count + kTile - 1) / kTile), dim3(kTile), 0, s, d, idx, count);
}
void launchMoveEntities(const DeviceState& d, const uint32_t* src, const uint32_t* dst, uint32_t moves, hipStream_t s)
{
  if (!moves) return;
  const uint64_t threads = (uint64_t)moves * kMoveSlots;
  hipLaunchKernelGGL(k_move_entities, dim3((uint32_t)((threads + kTile - 1) / kTile)), dim3(kTile), 0, s, d, src, dst, moves);
}
void launchSetFrustum(const DeviceState& d, const Frustum6& fr, hipStream_t s)
{
  hipLaunchKernelGGL(k_set_frustum, dim3(1), dim3(64), 0, s, const_cast<float*>(d.frustum), fr);
}
void launchResetParity(const DeviceState& d, uint32_t q, hipStream_t s)
{
  hipLaunchKernelGGL(k_reset_parity, dim3(1), dim3(kTile), 0, s, d, q);
}
void launchPatchParents(const DeviceState& d, const uint32_t* pairs, uint32_t count, hipStream_t s)
{
  if (!count) return;
  hipLaunchKernelGGL(k_patch_parents, dim3((count + kTile - 1) / kTile), dim3(kTile), 0, s, d, pairs, count);
}
void launchGatherRows(const DeviceState& d, const uint32_t* idx, uint32_t count, float* out12, hipStream_t s)
{
  if (!count) return;
  hipLaunchKernelGGL(k_gather_rows, dim3((count + kTile - 1) / kTile), dim3(kTile), 0, s, d, idx, count, out12);
}
void launchEmitDraws(const DeviceState& d, uint32_t budget, void* items, hipStream_t s)
{
  hipLaunchKernelGGL(k_emit_draws, dim3(1024), dim3(kTile), 0, s, d, budget, (DrawItem80*)items);
}

} // namespace sctick

#ifdef SC_DIAG_WAVETIME
extern "C" int scTickDiagWaveTime(unsigned long long out[40], int reset)
{
  using namespace sctick;
  if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(g_waveDiag), sizeof(unsigned long long) * 40) != hipSuccess) return 0;
  if (reset) {
    unsigned long long z[40] = {0};
    z[3] = ~0ull;
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_waveDiag), z, sizeof z) != hipSuccess) return 0;
  }
  return 1;
}
extern "C" int scTickDiagWaveRows(uint32_t* out, uint32_t waves)
{
  using namespace sctick;
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_waveRows), sizeof(uint32_t) * 8u * (waves < 8192u ? waves : 8192u)) == hipSuccess ? 1 : 0;
}
#endif
