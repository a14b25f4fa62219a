// sc_tick_internal.h -- device-side data layout shared by the kernels and the C-ABI implementation.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <algorithm>

namespace sctick {

#ifdef __HIPCC__
// the wave's predicate mask straight from the compare (HIP's __ballot goes through an integer: v_cndmask + v_cmp_ne per call)
__device__ __forceinline__ unsigned long long ballot64(bool pred) { return __builtin_amdgcn_ballot_w64(pred); }
#endif


// ---- link word: one dword per entity carries topology + component flags --------------------
//   bits  0..23  dense index of the parent (kNoParent = root)          Transform::parent
//   bit   24     has RenderMesh  (culling candidate)                   sc_world_partition.cpp:1206-1210
//   bit   25     has Bounds                                            sc_world_partition.cpp:1252-1263
//   bits 26..28  hierarchy depth: 0..kMaxChain exact, kDeep = deeper (handled by level kernels),
//                kUnreachable = in/below a parent cycle (never visited, sc_ecs.cpp:173-210)
//   bits 29..31  rotation about X / Y / Z is trivial: the uploaded sin is exactly 0 and cos exactly 1
//                (angle 0), so the kernel substitutes the constants instead of streaming them
constexpr uint32_t kParentMask  = 0x00FFFFFFu;
constexpr uint32_t kNoParent    = 0x00FFFFFFu;
constexpr uint32_t kHasMesh     = 1u << 24;
constexpr uint32_t kHasBounds   = 1u << 25;
constexpr uint32_t kDepthShift  = 26;
constexpr uint32_t kDepthMask   = 7u;
constexpr uint32_t kMaxChain    = 3;      // ancestors the fused kernel walks itself
constexpr uint32_t kDeep        = 5;      // depth > kMaxChain: level kernels
constexpr uint32_t kUnreachable = 7;
constexpr uint32_t kRotTrivialX = 1u << 29, kRotTrivialY = 1u << 30, kRotTrivialZ = 1u << 31;
__host__ __device__ inline uint32_t linkDepth(uint32_t lk) { return (lk >> kDepthShift) & kDepthMask; }

constexpr uint32_t kTile = 256;           // entities per workgroup pass (4 waves of 64)
constexpr uint32_t kMaxSpanWords = 128;   // spans up to 4096 entities keep the fused producer's ballots in LDS

struct Frustum6 { float p[6][4]; };       // (nx, ny, nz, d) x 6, Frustum (sc_world_partition.h:39-43)

// All 4-byte per-entity streams live in ONE slab, stream k at byte offset k * capBytes, so a kernel
// addresses every stream from a single base (one SGPR pair, 32-bit lane offsets) instead of keeping
// thirty pointers live in scalar registers.  The world-matrix rows form a second slab of float4.
enum Stream : uint32_t {
  kPX, kPY, kPZ, kRSX, kRCX, kRSY, kRCY, kRSZ, kRCZ, kSX, kSY, kSZ,
  kBMINX, kBMINY, kBMINZ, kBMAXX, kBMAXY, kBMAXZ, kLINK, kLAYERS, kMESH, kMATERIAL, kStreamCount
};

// Lane graph of the traffic system (TrafficLaneGraph, src/engine/traffic/sc_traffic_lanes.h:13-30) as the on-rails advance
// reads it: per segment its start node's position and length, its direction and speed limit (laneSpeedLimit = the start
// node's, sc_traffic_lanes.cpp:392-400), its end node, the active flag, and sin / cos of yaw = atan2(dir.x, dir.z) taken with
// the HOST libm when the graph is set -- an on-rails agent's rotation only ever takes these per-segment values
// (sc_traffic_ai.cpp:455), so the device never evaluates a trigonometric function and stays bit-equal to the host.
struct LaneGraphDev {
  const float4* segA;           // start.xyz, length
  const float4* segB;           // dir.xyz, speed limit
  const uint4*  segC;           // end node, active, bits of sin(yaw), bits of cos(yaw)
  const float4* nodePos;        // xyz
  const uint32_t* nodeConnOff;  // [nodes + 1] offsets into nodeConn (LaneNode::connections)
  const uint32_t* nodeConn;     // segment ids
  uint32_t segments, nodes;
};
constexpr uint32_t kInvalidLane = 0xFFFFFFFFu;               // kInvalidLaneId, sc_traffic_common.h:9
constexpr uint32_t kTierPhysics = 0, kTierKinematic = 1, kTierOnRails = 2;   // TrafficSimMode, sc_traffic_common.h:11-16
constexpr uint32_t kMoverTraffic = 3;                        // moverKind of a traffic agent (1 / 2: SynthWorld's straight-line movers)
constexpr uint32_t kTierNearCap = 1u << 16;                  // agents whose desired tier is not OnRails, per selection

// Device SoA state of one context.  All arrays are sized to `cap` (padded to kTile).
struct DeviceState {
  const char* fslab;        // kStreamCount streams x cap x 4 B
  char* rslab;              // 3 row streams x cap x 16 B (w0, w1, w2)
  uint32_t capBytes;        // cap * 4
  uint32_t capBytes16;      // cap * 16
  // inputs: Transform locals (sc_ecs.h:63-71); rotation kept as host-libm sin/cos of the Euler angles
  float *px, *py, *pz;
  float *rsx, *rcx, *rsy, *rcy, *rsz, *rcz;
  float *sx, *sy, *sz;
  uint32_t* link;
  uint32_t* dirty;          // 1 bit per entity (Transform::dirty), word i/32
  uint32_t* unreach;        // 1 bit per entity: depth == kUnreachable (dirty survives the tick there)
  // Bounds::localAabb (sc_world_partition.h:298-301)
  float *bminx, *bminy, *bminz, *bmaxx, *bmaxy, *bmaxz;
  // RenderMesh + collision layers
  uint32_t *meshId, *materialId;
  uint32_t* layers;         // group | mask << 16
  // outputs
  float4 *w0, *w1, *w2;     // world matrix rows 0..2 (affine 3x4; row 3 is 0,0,0,1)
  uint64_t* vis;            // visibility bits, word i/64
  uint64_t* cand;           // candidate bits (only written when the culled list is requested)
  uint64_t* recomp;         // "recomputed this tick" bits (only when deep levels exist)
  uint32_t* blockVis;       // visible count per span
  uint32_t* blockCand;      // candidate count per span
  uint32_t* visibleIdx;     // ordered visible dense indices
  uint32_t* culledIdx;
  const float* frustum;     // the six planes (nx, ny, nz, d), device copy refreshed when the host sets them
  uint32_t* counters;       // [0] visible total, [1] culled total, [2] pairs, [3] bin overflow, [4] draws, [5] dropped
  // broadphase
  float4 *aabbMin, *aabbMax;   // dense-order world AABBs (debug / read-back only: SC_TICK_DENSE_AABBS)
  uint32_t* binCount;          // records per sector bin (self-cleaning: the pair kernel zeroes what it read)
  uint32_t* binLayers;         // OR of the records' (group | mask << 16) per bin: lets the pair kernel skip a bin unread
  float4* bins;                // [sector][kBinCap][2]: (min.xyz, layers) (max.xyz, id | primary<<31)
  // Home slots (round 3): where an entity's records went the last time the bins were filled with reservations ("learn" tick).
  // While the entity's box keeps its primary sector, its records go straight to those slots -- no reservation, i.e. no atomic,
  // which on this chip executes at the memory side (11 of the fused kernel's 39 us were its atomics).  Shared by every parity copy.
  uint32_t* homeA;             // [cap] primary sector of the entity's records at the learn tick (0xFFFFFFFF: none)
  uint32_t* homeB;             // [cap] four 8-bit slots: the copies in sectors homeA + {0, 1, binSX, binSX + 1}; 0xFF = no reservation
  uint32_t* homeCount;         // [sectors] slots of each bin that are reserved: what binCount starts a tick from
  uint32_t* homeLayers;        // [sectors] OR of the reserved records' layer words: what binLayers starts a tick from
  // Ordered home slots (round 4): at the learn tick the reserved records of every bin are re-numbered -- and moved -- so that the
  // records that pass the group/mask filter against their own kind ("cast" records: dynamic bodies) come first.  A bin that holds
  // nothing but its reserved records on a later tick is then searched without classifying anything: the cast records are slots
  // [0, D), everything else behind them (k_order_home, "fast sectors" in the pair role).
  uint32_t* homeCast;          // [sectors] bits 0..7: D, cast records among the reserved slots; kCastFast: the rest cannot meet each other
  const uint32_t* pairConst;   // what every pair-role workgroup copies into LDS: words [0, 1008) the triangular pair table (2016 x uint16), [kPairConstCast, +65) the cast table
  const float4* nullRec;       // one null record (nullRecord(): inverted box, layers 0) for lanes that have no bin slot to load
  uint8_t* homePerm;           // [sectors][kBinCap] learn tick only: slot a reserved record moved to (k_home_flags re-numbers homeB through it)
  // Lazy records (round 3): a bin whose reserved records cannot pass the group/mask filter against each other is read by the
  // pair search only when a record from elsewhere arrives that can -- so its owners do not write it (homeB bit 6 of a slot
  // byte = "write every tick"), and the wave that does need such a bin rebuilds its records from the owners' matrices.
  uint32_t* lazyCtl;           // [0] big boxes of the last pair search (!= 0: the next fused kernel writes every record),
                               // [1 + parity] 1 = that tick's fused kernel wrote every reserved record (what its pair search goes by),
                               // [1 + kMaxParity + parity] records / big boxes that arrived from a neighbour outside the declared layer vocabulary
  float4* bigList;             // [cap][2] boxes that cannot be binned (too large, outside the rect, bin full)
  float4* spill;               // [ovfCap][2] sector OVERFLOW list: records that found their sector bin full -- this tile's own
                               // (fused kernel) and the neighbours' border records (merge) alike ...
  uint32_t* spillSector;       // ... and the sector each belongs to: the wave that searches that sector gathers them back
  uint32_t* ovfLo;             // per sector: lowest overflow-list index tagged with it (0xFFFFFFFF: none) ...
  uint32_t* ovfHi;             // ... and one past the highest: the slice a sector's wave has to sweep (entities of a sector sit
                               // together in pool order, so its overflow records sit together in the list)
  uint32_t* crowdQueue;        // [kMaxParity][sectors] crowded sectors of the tick -- queued by whoever was handed slot 64 of a bin -- for the pair search
  uint32_t* ovfIdx;            // [pair-role waves][kOvfPerSector] scratch: overflow-list indices of the sector a wave is working on
  uint2* pairs;                // (a, b) ids, a < b; id = rank << 24 | dense index; kPairShards segments of shardCap
  uint32_t* pairShardCount;    // [kMaxParity parities + snapshot][kPairShards] counters, one per 128-byte line (kShardStride words apart)
  // upstream movers (allocated on first scTickUploadMovers)
  uint32_t* moverKind; float *mvx, *mvz, *mlox, *mloz, *mhix, *mhiz;
  // traffic agents (TrafficAgent + TrafficVehicle, sc_traffic_common.h:26-44; allocated on first scTickUploadTrafficAgents)
  LaneGraphDev lanes;
  uint32_t* aLane; float* aS; float* aSpeed; uint32_t* aMode; float* aLook; uint32_t* aDesired;
  float* aBrake;               // obstacleBrake per agent from its front ray (sc_traffic_ai.cpp:300-345); nullptr = no sensors: brake 0
  // per-agent TrafficSensors (sc_traffic_common.h:46-53): frontRayLength / safeDistance per entity, and what the AI leaves in them (lastHitDistance,
  // lastHitType: 0 none, 2 vehicle, 3 world); allocated with aBrake
  float* aRayLen; float* aSafe; float* aHitDist; uint32_t* aHitType;
  float4* agentRays;           // [cap][2] with the rays cast in the PAIR half (tiled, in order): (origin xyz, ray length) (forward x, forward z, safe distance, -)
                               // of every listed agent as the tick half found them -- the frame producer may have moved the agents on by then
  uint32_t* agentList;         // dense indices of the OnRails agents, rebuilt every tick that casts their rays
  uint32_t* agentCount;
  uint32_t* tierCounts;        // [0..2] desired tiers, [3] entries in tierNear
  uint2* tierNear;             // (dense index, bits of the distance) of agents whose desired tier is Physics or Kinematic
  // multi-GPU border exchange: one message per neighbour direction (caller-owned device buffers)
  uint32_t* borderSend[8];
  uint32_t* borderRecv[8];
};

// Border message layout (uint32 words): [0] records in the spill-over area, [1] overflow flag, [2..2+L) per-cell counts,
// then the record capacity (8 words a record): borderFixedSlots() fixed slots for every cell, cell after cell, and behind
// them the spill-over area (what cells hold beyond their fixed slots, packed cell after cell); then the big-box section:
// [0] boxes, [1] overflow flag, boxes (8 words each); then, with traffic sensors on a tiled world, the halo section.
// L = sectors on that ring side.
constexpr uint32_t kPairConstCast = 1008;      // DeviceState::pairConst: first word of the cast table
constexpr uint32_t kPairConstWords = 1076;     // (padded to 16 bytes)
constexpr uint32_t kBorderHeader = 2;
constexpr uint32_t kBorderRecsPerBin = 16;    // default: a message holds L * 16 records (shared by the side's sectors: a crowded ring sector
                                              // may take more than its share), at least one full sector; scTickSetBorderCapacity raises it
constexpr uint32_t kOvfPerSector = 1024;      // overflow records one sector can hold besides its bin (what lies beyond is counted, never silent)
constexpr uint32_t kOvfWaves = 2048u * 4u;    // pair-role waves that can own an ovfIdx scratch row (the launcher's workgroup cap x 4)
constexpr uint32_t kBorderBigCap = 128;       // big boxes (wider than 2x2 sectors, outside the rectangle, bin full) per message
constexpr uint32_t kBorderBigWords = 2u + kBorderBigCap * 8u;
constexpr float kBigReach = 2.0f;             // sectors around a tile's owned region within which it must know a big box

constexpr uint32_t kBinCap = 64;          // one wave lane per record of a bin
// Pair output is sharded: a wave buffers its hits in LDS and appends them to the segment of its workgroup's
// shard with one atomic per flush; 64 counters on separate cache lines instead of one hot word.
constexpr uint32_t kPairShards = 64, kShardStride = 32, kWavePairBuf = 192;
constexpr uint32_t kPrimary = 0x80000000u;
// counters[]: 0 visible, 1 culled, 4 draws, 5 dropped, 6 renderables; per tick parity q: 8+8q+{0 pairs, 1 big, 2 bin-full}
constexpr uint32_t kCtrPar = 8, kCtrPairs = 0, kCtrBig = 1, kCtrBinFull = 2, kCtrBorderLost = 3, kCtrBigLocal = 4, kCtrSpill = 5;
constexpr uint32_t kCtrCrowdTail = 6, kCtrCrowdHead = 7;     // crowded sectors queued by the binning / taken by the pair search (reset with the parity's counters)
constexpr uint32_t kHomeOff = 0, kHomeLearn = 1, kHomeUse = 2;
constexpr uint32_t kNoHome = 0xFFFFFFFFu, kNoSlot = 0xFFu;
constexpr uint32_t kHomeHot = 0x80000000u;                         // homeCount: the bin was rebuilt since the learn tick, its owners write it again
constexpr uint32_t kCastFast = 0x100u;                            // homeCast: no two of the bin's reserved non-cast records pass the filter against each other
constexpr uint32_t kSlotMask = 0x3Fu, kSlotAlways = 0x40u;     // a homeB slot byte: slot in the bin, "written on every tick" (lazy records)
// Tick "parity": which copy of the per-tick broadphase state a tick works on.  The in-order flows alternate between two
// copies; pipelined tiles rotate through `depth` (2..kMaxParity) copies, so that the pair half of tick t may still run while
// the fused kernels of ticks t+1 .. t+depth-1 refill the others.
constexpr uint32_t kMaxParity = 4, kCounterWords = kCtrPar + 8u * (kMaxParity + 1u);
// (kCtrBig counts the big list: this tile's boxes, then -- after the border merge -- its neighbours' that reach it;
//  kCtrBigLocal keeps this tile's own count; kCtrBorderLost: records or boxes a border message had no room for, or
//  big boxes that reach beyond the eight neighbouring tiles -- pairs may be missing)

struct TickParams {
  uint32_t n;               // entities
  uint32_t span;            // entities per workgroup (multiple of kTile)
  uint32_t flags;           // SC_TICK_* | internal bits below
  uint32_t freeze;          // CullingState::freezeCulling
  uint32_t frustumValid;
  // broadphase grid: sectors [binOx, binOx+binSX) x [binOz, binOz+binSZ), keyed like worldToSector
  float binOx, binOz, invSector;
  uint32_t binSX, binSZ;
  uint32_t parity;          // tick parity: selects the counter set (and, through the context, the bins / lists)
  uint32_t maxPairs;
  uint32_t rankBits;        // rank << 24, OR-ed into every box id
  uint32_t neighbourMask;   // bit d set: a neighbour tile exists in direction d (its ring side is foreign)
  uint32_t variant;         // kernel variant selector (A/B tuning; 0 = default)
  uint32_t chain;           // min(deepest hierarchy level, kMaxChain): selects the fused kernel's specialisation
  uint32_t resetParity;     // kFlagDeferredReset: the parity whose counters this tick's end-of-tick kernel clears (the next tick's)
  uint32_t tileX, tileZ, tilesX, tilesZ;   // this tile's place in the grid of equal tiles (tilesX == 0: unknown, no big-box exchange)
  uint32_t producerKind; float producerParam;   // with SC_TICK_PRODUCE_NEXT: the frame producer fused into the end-of-tick kernel
  float trafficSmooth, trafficMult;             // movers: 1 - exp(-2.5 dt) (smoothExp, sc_traffic_ai.cpp:58-62; host libm) and dbg->speedMultiplier
  uint32_t bigCap;          // entries the big list can hold (capacity + room for the neighbours' boxes): every index into it is held below this
  uint32_t ovfCap;          // entries the sector overflow list can hold
  uint32_t pairRun;         // pair role: a wave takes its sectors in runs of pairRun consecutive ones (pairRunFor())
  uint32_t cus;             // compute units of the device (launch geometry)
  uint32_t homeMode;        // bins: 0 every record reserves its slot (atomics), 1 the same and the slots are remembered (learn tick),
                            // 2 records with a remembered slot are stored there directly (kHome*)
  uint32_t homeReset;       // the pair search leaves binCount / binLayers at homeCount / homeLayers instead of zero
  uint32_t borderRecs;      // border messages: records per ring sector of a side, on average (scTickSetBorderCapacity; kBorderRecsPerBin)
  // draw emission folded into the end-of-tick kernel's compaction role (RenderPrepStreamingSystem, sc_world_partition.cpp:1306-1329): the
  // workgroup that places a visible entity in the ordered list knows its position there, i.e. the index of its draw item
  uint32_t emitMode;        // 0 off, 1 items to emitTarget (DrawItem80[]), 2 the frame read-back block at emitTarget (header, visible head, items)
  uint32_t emitBudget;      // RenderPrepStreaming's draw budget (0 = none)
  uint32_t emitMaxVisible;  // mode 2: entries of the visible list the block holds
  uint32_t emitTickLo, emitTickHi;
  uint32_t* emitTarget;
  uint32_t cleanStay;       // kHomeUse ticks: an entity whose matrix was not rebuilt leaves its (always written) slots as they are -- they hold
                            // this very record; 0 after anything else changed boxes (bounds / matrix uploads) and when the pair half runs pipelined
  uint32_t lazy;            // kHomeUse ticks: reserved records of bins that cannot produce a pair may be left unwritten (DeviceState::lazyCtl):
                            // 1 = bins whose own records cannot meet each other, rebuilt on demand (in-order flows); 2 = bins whose records
                            // can meet nothing in the world's declared vocabulary, never needed (pipelined tiles, scTickSetWorldLayers);
                            // 0 when something else reads the bins (ray queries, traffic sensors) or neither applies
  uint32_t vocab;           // lazy 2: group bits | mask bits << 16 of every collider that can exist in the tiled world
  uint32_t halo;            // border messages carry the halo section (traffic sensors on a tiled world): the neighbours' core-edge records land in the ring bins
  uint32_t vocabKnown;      // scTickSetWorldLayers declared the world's layer vocabulary (`vocab`): the border merge counts arrivals outside it
  uint32_t sweepOnly;       // host hint (worldCanPair): no two layer words of this world admit a pair -- the pair role only sweeps the counters; sizes its grid, nothing else
  uint32_t fastPairs;       // the pair role takes bins that hold nothing but their ordered reserved records through the fast path (homeCast)
};
// neighbour directions: d = (dz+1)*3 + (dx+1), skipping the centre -> 0..7; opposite(d) = 7 - d
__host__ __device__ inline void borderDir(uint32_t d, int& dx, int& dz) { const uint32_t k = d < 4 ? d : d + 1; dx = (int)(k % 3) - 1; dz = (int)(k / 3) - 1; }
// side messages span the whole ring side including its two end cells (used when no tile exists past them)
__host__ __device__ inline uint32_t borderLen(uint32_t d, uint32_t coreSX, uint32_t coreSZ) { int dx, dz; borderDir(d, dx, dz); return (dx != 0 && dz != 0) ? 1u : (dx == 0 ? coreSX + 2u : coreSZ + 2u); }
__host__ __device__ inline uint32_t borderDirOf(int dx, int dz) { const uint32_t k9 = (uint32_t)((dz + 1) * 3 + (dx + 1)); return k9 < 4 ? k9 : k9 - 1u; }
// hasNb(p, dx, dz): a tile exists one step in that direction
__host__ __device__ inline bool hasNb(const TickParams& p, int dx, int dz) { return (p.neighbourMask >> borderDirOf(dx, dz)) & 1u; }
// records one border message can carry: `recs` per ring sector of the side on average, at least one sector's bin and overflow list
constexpr uint32_t kSectorRecMax = 64u + kOvfPerSector;      // what one sector can hold at all: its bin + its share of the overflow list
__host__ __device__ inline uint32_t borderRecCap(uint32_t L, uint32_t recs) { return L * recs > kSectorRecMax ? L * recs : kSectorRecMax; }
// of a message's record capacity, every cell owns this many FIXED slots (the first records of a cell travel without a scan);
// the rest of the capacity is one spill-over area shared by the cells that hold more
constexpr uint32_t kBorderFixed = 4;
__host__ __device__ inline uint32_t borderFixedSlots(uint32_t L, uint32_t recs) { const uint32_t per = borderRecCap(L, recs) / L; return per < kBorderFixed ? per : kBorderFixed; }
__host__ __device__ inline uint32_t borderBinWords(uint32_t d, uint32_t coreSX, uint32_t coreSZ, uint32_t recs) { const uint32_t L = borderLen(d, coreSX, coreSZ); return kBorderHeader + L + borderRecCap(L, recs) * 8u; }
// The HALO section (round 4, traffic sensors on a tiled world): behind the big boxes, the records of the sender's CORE-EDGE sectors along that
// side -- what an obstacle ray cast from the receiver's edge needs to see in the neighbour's territory (sc_traffic_ai.cpp:300-345: the
// reference's ray sees the whole physics world).  Fixed slots: cell l owns kBinCap records at a fixed offset, so neither pack nor
// merge needs a prefix scan.  Layout: [L counts][L x kBinCap records of 8 words].  Present only when both tiles run with sensors.
__host__ __device__ inline uint32_t borderHaloWords(uint32_t d, uint32_t coreSX, uint32_t coreSZ) { const uint32_t L = borderLen(d, coreSX, coreSZ); return L + L * kBinCap * 8u; }
__host__ __device__ inline uint32_t borderWords(uint32_t d, uint32_t coreSX, uint32_t coreSZ, uint32_t recs, uint32_t halo = 0u)
{ return borderBinWords(d, coreSX, coreSZ, recs) + kBorderBigWords + (halo ? borderHaloWords(d, coreSX, coreSZ) : 0u); }
constexpr uint32_t kFlagHasDeep = 1u << 16;   // write recomp bits for the level kernels
constexpr uint32_t kFlagDeferredReset = 1u << 17;   // pipelined tiles: a pair kernel never clears the other parity's counters / big bits; a small kernel
                                                    // behind it on the pairs stream snapshots the results and clears its OWN parity
constexpr uint32_t kFlagDenseAabbs = 1u << 5; // == SC_TICK_DENSE_AABBS

// ---- renderer draw order (sc_tick_drawsort.hip) ----
constexpr uint32_t kSortThreads = 1024, kSortGroup = 8192;     // keys one workgroup orders per pass
constexpr uint64_t kDrawInvalid = 1ull << 55;                  // above every pipeline id (< 128): dropped draws sort last
constexpr uint32_t kNoMaterial = 0xFFu;                        // pipeline table entry of a material handle that does not exist
constexpr uint32_t kCtrDrawsSorted = 7;                        // counters[7]: draws that survived the renderer's handle checks
struct DrawSortState {
  uint64_t* key[2]; uint32_t* idx[2];     // ping-pong (key, rank in the visible list)
  uint32_t* hist;                         // [256][groups] digit totals for multi-workgroup passes
  const uint8_t* pipeline;                // Material::pipelineId per material handle, kNoMaterial = none
  uint32_t materialCount, meshCount;
  uint32_t passes; uint32_t shift[7];     // key bytes that can differ, least significant first
};
void launchSortedDraws(const DeviceState& d, const DrawSortState& st, uint32_t budget, uint32_t bound, void* items, hipStream_t s);

// ---- ray queries over the bins (sc_tick_queries.hip) ----
struct RayHit48 { uint32_t hit, id; float distance; float position[3]; float normal[3]; uint32_t layer; uint32_t pad; uint32_t pad2; };   // == ScTickRayHit
struct RayQueryState {
  const float4* origin;     // xyz + maxDist
  const float4* dir;        // xyz + mask (bit pattern)
  RayHit48* hits;
  uint32_t count;
};
void launchRayQueries(const DeviceState& d, const TickParams& p, const RayQueryState& q, hipStream_t s);
void launchAgentFrontRays(const DeviceState& d, const TickParams& p, hipStream_t s);
void launchAgentRaySnapshot(const DeviceState& d, const TickParams& p, hipStream_t s);         // tick half: list the agents, note their rays
void launchAgentFrontRaysFromSnapshot(const DeviceState& d, const TickParams& p, hipStream_t s); // pair half, behind the merge: cast them
void launchFillSensors(const DeviceState& d, uint32_t first, uint32_t count, float rayLen, float safe, hipStream_t s);
constexpr uint32_t kMaxOccupancyQueries = 256;
void launchOccupancy(const DeviceState& d, uint32_t n, const float4* q, uint32_t count, uint32_t* blocked, hipStream_t s);

// launchers (sc_tick_kernels.hip)
void launchXformCull(const DeviceState& d, const TickParams& p, uint32_t grid, hipStream_t s, hipEvent_t evA = nullptr, hipEvent_t evB = nullptr);
void launchSnapshotHome(const DeviceState& d, uint32_t sectors, uint32_t n, uint32_t binSX, uint32_t binSZ, uint32_t vocabMode, uint32_t vocab, uint32_t ordered, hipStream_t s);
void launchDeepLevel(const DeviceState& d, const TickParams& p, const uint32_t* levelList, uint32_t count, hipStream_t s);
void launchCompact(const DeviceState& d, const TickParams& p, uint32_t grid, hipStream_t s);
bool launchPairs(const DeviceState& d, const TickParams& p, hipStream_t s, hipEvent_t done = nullptr);
uint32_t pairRunFor(uint32_t sectors, uint32_t cus, uint32_t variant, bool sweepOnly);
void launchCompactPairs(const DeviceState& d, const TickParams& p, uint32_t compactGrid, hipStream_t s, hipEvent_t evA = nullptr, hipEvent_t evB = nullptr);
void launchGatherPairs(const DeviceState& d, const TickParams& p, uint32_t parity, uint2* dst, uint32_t* total, hipStream_t s);
void launchCompactPack(const DeviceState& d, const TickParams& p, uint32_t grid, hipStream_t s, hipEvent_t done = nullptr);
void launchBorderPack(const DeviceState& d, const TickParams& p, hipStream_t s);
void launchBorderMerge(const DeviceState& d, const TickParams& p, hipStream_t s);
void launchNudgeRootsX(const DeviceState& d, uint32_t n, float dx, hipStream_t s);
void launchAdvanceMovers(const DeviceState& d, uint32_t n, float dt, float trafficSmooth, float trafficMult, hipStream_t s);
struct TierParams { float px, pz, aEnter, aExit, bEnter, bExit; };
void launchTrafficTiers(const DeviceState& d, uint32_t n, const TierParams& tp, hipStream_t s);
void launchApplyTiers(const DeviceState& d, uint32_t n, const uint2* patches, uint32_t patchCount, hipStream_t s);
void launchTrafficDespawnKeys(const DeviceState& d, uint32_t n, float px, float pz, uint32_t* count, uint32_t* outIdx, unsigned long long* outKey, hipStream_t s);
void launchDenseAabbs(const DeviceState& d, uint32_t n, hipStream_t s);
void launchSetDirtyRange(const DeviceState& d, uint32_t first, uint32_t count, hipStream_t s);
void launchSetDirtyIndices(const DeviceState& d, const uint32_t* idx, uint32_t count, hipStream_t s);
void launchMoveEntities(const DeviceState& d, const uint32_t* src, const uint32_t* dst, uint32_t moves, hipStream_t s);
void launchSetFrustum(const DeviceState& d, const Frustum6& fr, hipStream_t s);
void launchResetParity(const DeviceState& d, uint32_t q, hipStream_t s);
void launchPatchParents(const DeviceState& d, const uint32_t* pairs, uint32_t count, hipStream_t s);
void launchGatherRows(const DeviceState& d, const uint32_t* idx, uint32_t count, float* out12, hipStream_t s);
void launchEmitDraws(const DeviceState& d, uint32_t budget, void* items, hipStream_t s);
// frame read-back block: header (kFrameHeaderWords), visible indices, draw items
constexpr uint32_t kFrameHeaderWords = 16;
void launchEmitDrawsStaged(const DeviceState& d, uint32_t budget, uint32_t* block, uint32_t maxVisible, uint64_t tick, hipStream_t s, hipEvent_t done = nullptr);
void launchStageFrame(const DeviceState& d, uint32_t* block, uint32_t maxVisible, uint32_t maxDraws, const void* items, uint32_t drawMode /*0 none, 1 plain, 2 sorted*/,
                      uint64_t tick, hipStream_t s, hipEvent_t done = nullptr);

} // namespace sctick
