// sc_tick_api.hip -- implementation of the C ABI in include/sc_tick.h (host side of libsc_tick.so).
//
// Host code here is compiled with -ffp-contract=off as well: the frustum planes and the sin/cos of
// the Euler angles are produced on the host with the same libm calls and the same operation order
// as the reference (sc_world_partition.cpp:1071-1103, sc_math.cpp:102-107).
#include "../../include/sc_tick.h"
#include "sc_tick_internal.h"
#include "sc_tick_rccl.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <unordered_map>
#include <unordered_set>
#include <vector>

using namespace sctick;

namespace {

thread_local std::string gCreateError;

struct EventPair { hipEvent_t a = nullptr, b = nullptr; };

} // namespace

// SC_TICK_HOSTPROBE=1: where scTickRun's host time goes (printed when the context is destroyed; a development aid)
struct HostProbe { double acc[8] = {0}; uint64_t n = 0; bool on = false; };
static HostProbe g_probe;
static inline double probeNow() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

struct ScTickContext
{
  ScTickContextDesc desc{};
  int device = 0;
  uint32_t cap = 0;          // padded capacity
  uint32_t n = 0;
  hipStream_t stream = nullptr;
  DeviceState d{};
  std::vector<void*> allocs;
  std::string err;

  // host mirrors needed to (re)build link words
  std::vector<int32_t> hParent;
  std::vector<uint32_t> hLayers;     // group | mask << 16 as uploaded (scTickUploadLayers), for the world-vocabulary contract
  bool layerSetStale = true, ownLayersCanPair = true; uint32_t layerSetN = 0;      // worldCanPair(): do any two of this context's layer words admit a pair?
  std::vector<uint8_t> hFlags;       // bit0 has mesh, bit1 has bounds, bits 2..4 rotation about X/Y/Z trivial (sin 0, cos 1)
  std::vector<uint32_t> hChildren;   // direct children per entity (valid while !linksStale)
  // child lists in dense-index space (-1 = none), valid while !linksStale: a despawn patches exactly the links that
  // name a relocated entity instead of re-linking the world
  std::vector<int32_t> hFirstChild, hNextSib, hPrevSib;
  bool linksStale = true;
  uint32_t maxDepth = 0, unreachable = 0, relinks = 0;
  std::vector<uint32_t> levelOffsets;   // offsets into dLevelList for depth kMaxChain+1, +2, ...
  uint32_t* dLevelList = nullptr;
  uint32_t levelListCap = 0;

  Frustum6 frustum{};
  int frustumValid = 0;
  bool frustumStale = true;          // the device copy (DeviceState::frustum) is behind the host's
  int freeze = 0;

  uint32_t spansWanted = 1536;
  uint32_t cus = 0;           // compute units (hipDeviceProp_t::multiProcessorCount)
  uint32_t variant = 0;       // SC_TICK_VARIANT: bit0 = chain-walk K1 instead of the wave-cooperative one
  uint32_t lastFlags = 0;

  // scratch device buffers for indexed read-back
  uint32_t* dIdx = nullptr; float* dRows = nullptr; uint32_t scratchCap = 0;
  void* dDraws = nullptr;
  void* lastDraws = nullptr;           // where the last SC_TICK_DRAWS wrote its items (dDraws, or the frame read-back block)
  RayQueryState rays{};                // scTickSetRayQueries: device copies of the batch + the hit buffer
  uint32_t rayCap = 0;
  DrawSortState sort{};                // renderer draw order (scTickSetDrawSortTable); key/idx buffers allocated on first use
  uint8_t* dPipeline = nullptr; uint32_t pipelineCap = 0;

  // profiling
  bool profiling = false;
  uint32_t profPeriod = 1;     // record events on every profPeriod-th tick only (event records are not free)
  uint32_t profMask = 0xFFFFFFFFu;   // ... and only around these kernels (bit SC_TICK_K_*; scTickSetProfilingKernels)
  uint64_t tickIndex = 0;
  std::vector<EventPair> times[SC_TICK_K_COUNT];
  std::vector<EventPair> eventPool;

  // graph
  bool graphMode = false;
  hipGraph_t graph[kMaxParity] = {};                   // one per broadphase tick parity
  hipGraphExec_t graphExec[kMaxParity] = {};
  TickParams graphParams[kMaxParity]{};
  bool graphWhole[kMaxParity] = {};                    // the captured graph holds the whole tile step (exchange + pair half)
  bool captureWholeStep = false;                       // set by scTickTileStep around its scTickRun
  bool ownStep = false;                                // scTickTileStep is running: the pair half follows the tick half at once, nothing of the host's in between
  uint64_t topoEpoch = 0, graphEpoch[kMaxParity] = { ~0ull, ~0ull, ~0ull, ~0ull };
  // pipelined tile + graph replay: the pair half (exchange, merge, queries, pair search, snapshot) is a graph of its own,
  // replayed on the pairs stream; the two graphs of a step are ordered by events recorded between them, outside any capture
  hipGraph_t pairGraph[kMaxParity] = {};
  hipGraphExec_t pairGraphExec[kMaxParity] = {};
  TickParams pairGraphParams[kMaxParity]{};
  bool pairGraphExchange[kMaxParity] = {};
  uint64_t pairGraphEpoch[kMaxParity] = { ~0ull, ~0ull, ~0ull, ~0ull };
  bool lastTickSampled = false;                        // the last scTickRun recorded profiling events (ran eagerly)
  bool lastTickLearn = false;                          // the last scTickRun was a learn tick of the home slots (ran eagerly)
  // home slots of the bins (binEntityWave): remembered at a learn tick, used until the world's shape changes or they age
  bool homeEnabled = true, homeValid = false, homeCountsLive = false;
  bool lazyEnabled = true;                             // lazy records (DeviceState::lazyCtl)
  bool fastPairs = true;                               // ordered home slots: bins without visitors take the pair role's fast path (DeviceState::homeCast)
  bool lastTickLazy = false, lastTickStay = false, lastTickSweepOnly = false; uint32_t learnTicks = 0;  // scTickGetBinStats
  bool worldLayersKnown = false; uint32_t worldLayers = 0;   // scTickSetWorldLayers: group bits | mask bits << 16 of every collider of the tiled world
  bool boxesTouched = false;                           // bounds or world matrices were uploaded since the last broadphase tick (TickParams::cleanStay)
  uint64_t homeEpoch = ~0ull; uint32_t homeAge = 0, homePeriod = 64;
  uint32_t homeXform = 0;                              // SC_TICK_XFORM of the learn tick: entities deeper than the fused kernel's chain are binned by the level
                                                       // kernels on transforming ticks and by the fused kernel otherwise -- slots learned one way are not valid the other
  bool capturing = false;                              // enqueueStages runs inside a stream capture
  bool packedRides = false;                            // this tick's `packed` event was attached to the compaction + pack dispatch

  // broadphase
  uint32_t sectors = 0, maxPairs = 0;
  uint2* dPairsOut = nullptr;          // gathered (contiguous) pair list for read-back
  uint32_t* dPairTotal = nullptr;      // [0] pairs found, [1] truncated flag
  uint32_t parity = 0, lastParity = 0;
  uint32_t rank = 0, neighbourMask = 0;
  uint32_t tileX = 0, tileZ = 0, tilesX = 0, tilesZ = 0;
  uint32_t producerKind = 0; float producerParam = 0.0f;      // part of the frame when set (scTickSetFrameProducer)
  float trafficMult = 1.0f;                                   // TrafficDebugState::speedMultiplier (sc_traffic_common.h:63)
  bool sensors = false; float sensorRay = 20.0f, sensorSafe = 10.0f;   // TrafficSensors defaults (sc_traffic_ai.cpp:307-309)
  bool halo = false;                                                    // border messages carry the halo section (sensors were on when the tile got its neighbours)
  void* laneAllocs[6] = {};                                   // device copies of the lane graph (scTickSetLaneGraph)
  uint2* dTierPatch = nullptr;                                // tier selection: the few modes the caps changed
  bool pairsPending = false;
  TickParams pendingParams{};
  hipStream_t ownStream = nullptr;
  // pipelined tiles (scTickSetPairsStream): merge + queries + pair search of tick t run on a second stream under the fused
  // kernel of tick t+1; everything the two halves share is double-buffered by tick parity (set 0 lives in `d`)
  hipStream_t pairsStream = nullptr;
  struct AltSet { uint32_t* binCount = nullptr; uint32_t* binLayers = nullptr; float4* bins = nullptr; float4* bigList = nullptr;
                  float4* spill = nullptr; uint32_t* spillSector = nullptr; uint32_t* ovfLo = nullptr; uint32_t* ovfHi = nullptr;
                  uint32_t* borderSend[8] = {}; uint32_t* borderRecv[8] = {}; };
  AltSet alt[kMaxParity - 1];          // parity q > 0 works on alt[q - 1]
  uint32_t pipeDepth = 3;              // copies a pipelined tile rotates through (scTickSetPipelined)
  uint32_t borderRecs = kBorderRecsPerBin;   // border messages: records per ring sector of a side on average (scTickSetBorderCapacity)
  hipEvent_t packed[kMaxParity] = {}, pairsDone[kMaxParity] = {};
  bool pairsInFlight[kMaxParity] = {};
  // library-owned exchange (scTickCommInit): one RCCL communicator per context, the border messages of both tick
  // parities in buffers of the library's own, the neighbour in direction d at rank peer[d]
  // per-frame read-back (scTickSetFrameReadback): staged on the tick stream, copied on a stream of its own, double-buffered
  struct FrameReadback {
    uint32_t maxVisible = 0, maxDraws = 0; size_t bytes = 0;
    uint32_t* dBlock[2] = {}; uint32_t* hBlock[2] = {};
    hipStream_t copyStream = nullptr; hipEvent_t staged[2] = {}, copied[2] = {};
    bool inFlight[2] = { false, false }; uint64_t frames = 0;
  } rb;
  ncclComm_t comm = nullptr;
  uint32_t commSize = 0, commRank = 0;
  int32_t peer[8] = { -1, -1, -1, -1, -1, -1, -1, -1 };
  bool peersSet = false;
  uint32_t* ownBorder[kMaxParity][8][2] = {};   // [parity][direction][send, recv]
  double hostAcc[2] = { 0.0, 0.0 }; uint64_t hostSteps = 0;     // scTickTileStep: host time issuing the tick half / the exchange + pair half
  hipStream_t ownPairsStream = nullptr;
  uint32_t* dGather = nullptr; uint32_t gatherCap = 0;          // scTickGatherVisibleCounts: one count per rank
};

namespace {

bool fail(ScTickContext* c, const char* what, hipError_t e = hipSuccess)
{
  char buf[512];
  if (e != hipSuccess) std::snprintf(buf, sizeof buf, "%s: %s", what, hipGetErrorString(e));
  else std::snprintf(buf, sizeof buf, "%s", what);
  if (c) c->err = buf; else gCreateError = buf;
  return false;
}

#define HIP_OK(c, call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { fail((c), #call, e_); return 0; } } while (0)

bool bind(ScTickContext* c)
{
  const hipError_t e = hipSetDevice(c->device);
  if (e != hipSuccess) return fail(c, "hipSetDevice", e);
  return true;
}

template <typename T>
bool dalloc(ScTickContext* c, T*& p, size_t count, bool zero = true)
{
  void* v = nullptr;
  const size_t bytes = std::max<size_t>(count, 1) * sizeof(T);
  hipError_t e = hipMalloc(&v, bytes);
  if (e != hipSuccess) return fail(c, "hipMalloc", e);
  // (the tick stream is non-blocking, i.e. not ordered against the null stream this memset runs on, and a memset of device memory may
  //  return before it is done: wait for it, so that whatever is enqueued next sees the zeros -- allocations are rare)
  if (zero) { e = hipMemset(v, 0, bytes); if (e == hipSuccess) e = hipStreamSynchronize(nullptr); if (e != hipSuccess) return fail(c, "hipMemset", e); }
  c->allocs.push_back(v);
  p = static_cast<T*>(v);
  return true;
}

bool sync(ScTickContext* c);

void dfree(ScTickContext* c, void* p)
{
  if (!p) return;
  auto it = std::find(c->allocs.begin(), c->allocs.end(), p);
  if (it != c->allocs.end()) c->allocs.erase(it);
  hipFree(p);
}

// scratch index / row buffers for indexed calls: grown geometrically, the old pair is released
bool needScratch(ScTickContext* c, size_t count)
{
  if (c->scratchCap >= count) return true;
  if (!sync(c)) return false;                         // nothing queued may still read the old buffers
  const size_t want = std::max(count, (size_t)c->scratchCap * 2u);
  dfree(c, c->dIdx); dfree(c, c->dRows);
  c->dIdx = nullptr; c->dRows = nullptr; c->scratchCap = 0;
  if (!dalloc(c, c->dIdx, want, false) || !dalloc(c, c->dRows, want * 12, false)) return false;
  c->scratchCap = (uint32_t)want;
  return true;
}

bool rangeOk(ScTickContext* c, uint32_t first, uint32_t count)
{
  if ((uint64_t)first + count > c->n) return fail(c, "range exceeds entity count");
  return true;
}

bool h2d(ScTickContext* c, void* dst, const void* src, size_t bytes)
{
  const hipError_t e = hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, c->stream);
  if (e != hipSuccess) return fail(c, "hipMemcpyAsync H2D", e);
  return true;
}
bool d2h(ScTickContext* c, void* dst, const void* src, size_t bytes)
{
  const hipError_t e = hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, c->stream);
  if (e != hipSuccess) return fail(c, "hipMemcpyAsync D2H", e);
  return true;
}
bool sync(ScTickContext* c)
{
  hipError_t e = hipStreamSynchronize(c->stream);
  if (e != hipSuccess) return fail(c, "hipStreamSynchronize", e);
  if (c->pairsStream) {                       // pipelined tiles: the pair search half runs on its own stream
    e = hipStreamSynchronize(c->pairsStream);
    if (e != hipSuccess) return fail(c, "hipStreamSynchronize (pairs stream)", e);
  }
  return true;
}

// the device state a tick of parity q works on: set 0 is `d` itself, set 1 swaps in the second copy of everything the
// pair search of one tick and the fused kernel of the next would otherwise share
DeviceState stateFor(const ScTickContext* c, uint32_t q)
{
  DeviceState s = c->d;
  if (c->pairsStream && q >= 1u && q < kMaxParity) {
    const ScTickContext::AltSet& a = c->alt[q - 1u];
    s.binCount = a.binCount; s.binLayers = a.binLayers; s.bins = a.bins; s.bigList = a.bigList;
    s.spill = a.spill; s.spillSector = a.spillSector; s.ovfLo = a.ovfLo; s.ovfHi = a.ovfHi;
    for (int k = 0; k < 8; ++k) { s.borderSend[k] = a.borderSend[k]; s.borderRecv[k] = a.borderRecv[k]; }
  }
  return s;
}

// results of the pair half (pairs, ray hits, broadphase counters) are read on the tick stream: let it finish first
bool joinPairs(ScTickContext* c)
{
  if (!c->pairsStream) return true;
  const hipError_t e = hipStreamSynchronize(c->pairsStream);
  if (e != hipSuccess) return fail(c, "hipStreamSynchronize (pairs stream)", e);
  return true;
}

// de-interleave [count][3] into three SoA streams and upload
bool upload3(ScTickContext* c, const float* src3, uint32_t first, uint32_t count, float* dx, float* dy, float* dz)
{
  std::vector<float> sx(count), sy(count), sz(count);
  for (uint32_t i = 0; i < count; ++i) { sx[i] = src3[3 * i]; sy[i] = src3[3 * i + 1]; sz[i] = src3[3 * i + 2]; }
  const bool ok = h2d(c, dx + first, sx.data(), count * 4u) && h2d(c, dy + first, sy.data(), count * 4u) &&
                  h2d(c, dz + first, sz.data(), count * 4u);
  return ok && sync(c);      // staging vectors die here
}

void computeSpan(const ScTickContext* c, uint32_t& span, uint32_t& grid)
{
  const uint32_t n = std::max(c->n, 1u);
  const uint32_t tiles = (n + kTile - 1) / kTile;
  const uint32_t g = std::min(std::max(c->spansWanted, 1u), tiles);
  const uint32_t tilesPer = (tiles + g - 1) / g;
  span = tilesPer * kTile;
  grid = (n + span - 1) / span;
}

// Transform::parent validation (sc_ecs.cpp:151-160) + depth / cycle classification.
// Returns the dense indices that were detached (they are marked dirty by the caller).
void rebuildLinks(ScTickContext* c, std::vector<uint32_t>& link, std::vector<uint32_t>& unreachBits,
                  std::vector<uint32_t>& detached, std::vector<std::vector<uint32_t>>& deepLevels)
{
  const uint32_t n = c->n;
  link.assign(n, 0);
  unreachBits.assign((c->cap + 31) / 32, 0);
  detached.clear();
  deepLevels.clear();

  std::vector<int32_t>& par = c->hParent;
  for (uint32_t i = 0; i < n; ++i) {
    const int32_t p = par[i];
    if (p == SC_TICK_NO_PARENT) continue;
    if (p < 0 || (uint32_t)p >= n || (uint32_t)p == i) { par[i] = SC_TICK_NO_PARENT; detached.push_back(i); }
  }

  std::fill(c->hChildren.begin(), c->hChildren.begin() + n, 0u);
  std::fill(c->hFirstChild.begin(), c->hFirstChild.begin() + n, -1);
  for (uint32_t i = n; i-- > 0;) {               // backwards, pushing at the front: lists come out in ascending order
    c->hNextSib[i] = -1; c->hPrevSib[i] = -1;
    if (par[i] == SC_TICK_NO_PARENT) continue;
    const uint32_t q = (uint32_t)par[i];
    c->hChildren[q]++;
    const int32_t head = c->hFirstChild[q];
    c->hNextSib[i] = head;
    if (head >= 0) c->hPrevSib[(uint32_t)head] = (int32_t)i;
    c->hFirstChild[q] = (int32_t)i;
  }

  // depth by walking up with memoisation; a walk that meets its own trail has found a cycle
  constexpr int32_t kUnknown = -1, kCycle = -2;
  std::vector<int32_t> depth(n, kUnknown);
  std::vector<uint32_t> trailMark(n, 0);
  std::vector<uint32_t> trail;
  for (uint32_t i = 0; i < n; ++i) {
    if (depth[i] != kUnknown) continue;
    trail.clear();
    uint32_t cur = i;
    int32_t above = -1;          // depth of the resolved node just above the trail (-1: the trail ends in a root)
    bool cyc = false;
    const uint32_t stamp = i + 1u;
    for (;;) {
      if (depth[cur] != kUnknown) { cyc = depth[cur] == kCycle; above = depth[cur]; break; }
      if (trailMark[cur] == stamp) { cyc = true; break; }       // met our own trail: a parent cycle
      trailMark[cur] = stamp;
      trail.push_back(cur);
      if (par[cur] == SC_TICK_NO_PARENT) break;
      cur = (uint32_t)par[cur];
    }
    // trail holds the unresolved nodes from i upward; number them from the top down.  Everything
    // that leads into a cycle has no root above it either, so it is unreachable as well.
    for (size_t k = trail.size(); k-- > 0;) depth[trail[k]] = cyc ? kCycle : ++above;
  }

  c->maxDepth = 0; c->unreachable = 0;
  for (uint32_t i = 0; i < n; ++i) {
    uint32_t dfield;
    if (depth[i] == kCycle) { dfield = kUnreachable; c->unreachable++; unreachBits[i >> 5] |= 1u << (i & 31u); }
    else {
      const uint32_t dv = (uint32_t)depth[i];
      c->maxDepth = std::max(c->maxDepth, dv);
      if (dv <= kMaxChain) dfield = dv;
      else {
        dfield = kDeep;
        const uint32_t lv = dv - kMaxChain - 1;
        if (deepLevels.size() <= lv) deepLevels.resize(lv + 1);
        deepLevels[lv].push_back(i);
      }
    }
    const uint32_t p = (par[i] == SC_TICK_NO_PARENT) ? kNoParent : (uint32_t)par[i];
    link[i] = p | ((c->hFlags[i] & 1u) ? kHasMesh : 0u) | ((c->hFlags[i] & 2u) ? kHasBounds : 0u) | (dfield << kDepthShift) |
              ((c->hFlags[i] & 4u) ? kRotTrivialX : 0u) | ((c->hFlags[i] & 8u) ? kRotTrivialY : 0u) | ((c->hFlags[i] & 16u) ? kRotTrivialZ : 0u);
  }
}

int flushLinks(ScTickContext* c)
{
  if (!c->linksStale) return 1;
  std::vector<uint32_t> link, unreachBits, detached;
  std::vector<std::vector<uint32_t>> deep;
  rebuildLinks(c, link, unreachBits, detached, deep);
  if (c->n) { if (!h2d(c, c->d.link, link.data(), (size_t)c->n * 4u)) return 0; }
  if (!h2d(c, c->d.unreach, unreachBits.data(), unreachBits.size() * 4u)) return 0;

  c->levelOffsets.assign(1, 0);
  std::vector<uint32_t> flat;
  for (auto& lv : deep) { flat.insert(flat.end(), lv.begin(), lv.end()); c->levelOffsets.push_back((uint32_t)flat.size()); }
  if (!flat.empty()) {
    if (flat.size() > c->levelListCap) {
      if (!sync(c)) return 0;
      dfree(c, c->dLevelList); c->dLevelList = nullptr; c->levelListCap = 0;
      uint32_t* p = nullptr;
      if (!dalloc(c, p, flat.size(), false)) return 0;
      c->dLevelList = p; c->levelListCap = (uint32_t)flat.size();
    }
    if (!h2d(c, c->dLevelList, flat.data(), flat.size() * 4u)) return 0;
  }
  uint32_t* dDet = nullptr;
  if (!detached.empty()) {
    // detached entities become dirty roots (sc_ecs.cpp:154-160)
    if (!needScratch(c, detached.size())) return 0;
    dDet = c->dIdx;
    if (!h2d(c, dDet, detached.data(), detached.size() * 4u)) return 0;
    launchSetDirtyIndices(c->d, dDet, (uint32_t)detached.size(), c->stream);
  }
  if (!sync(c)) return 0;
  c->linksStale = false;
  c->topoEpoch++;
  c->relinks++;
  return 1;
}

EventPair takeEvents(ScTickContext* c)
{
  EventPair p;
  if (!c->eventPool.empty()) { p = c->eventPool.back(); c->eventPool.pop_back(); return p; }
  hipEventCreate(&p.a); hipEventCreate(&p.b);
  return p;
}

struct Scoped
{
  ScTickContext* c; uint32_t k; EventPair p; bool on;
  Scoped(ScTickContext* c_, uint32_t k_) : c(c_), k(k_), on(c_->profiling && ((c_->profMask >> k_) & 1u) && (c_->tickIndex % c_->profPeriod) == 0)
  {
    if (on) { p = takeEvents(c); hipEventRecord(p.a, c->stream); }
  }
  ~Scoped() { if (on) { hipEventRecord(p.b, c->stream); c->times[k].push_back(p); } }
};

// Capacity of the sector overflow list, in records.  A box has at most four records (the sectors its xz range touches), a
// bin keeps 64 of a sector's, so 4 x capacity can never run out for a tile's own boxes however crowded its sectors are; the
// neighbours' border records that find their landing bin full come on top (what a border message can carry, eight messages).
uint32_t ovfRecords(const ScTickContext* c)
{
  const uint64_t own = 4ull * c->cap;
  uint64_t border = 0;
  for (uint32_t d = 0; d < 8; ++d) border += borderRecCap(borderLen(d, c->desc.tile_sectors_x, c->desc.tile_sectors_z), c->borderRecs);
  return (uint32_t)std::min<uint64_t>(own + border, 0xFFFFFF00ull);
}

void fillParams(ScTickContext* c, uint32_t flags, TickParams& p, uint32_t& grid)
{
  std::memset(&p, 0, sizeof p);
  p.n = c->n;
  computeSpan(c, p.span, grid);
  p.flags = flags & 0xFFFFu;        // SC_TICK_DENSE_AABBS == kFlagDenseAabbs
  if (c->levelOffsets.size() > 1 && (flags & SC_TICK_XFORM)) p.flags |= kFlagHasDeep;
  p.freeze = c->freeze ? 1u : 0u;
  p.frustumValid = c->frustumValid ? 1u : 0u;
  // the bin grid is the tile plus a ring of one sector: boxes that poke over the tile edge stay
  // binnable, and on a multi-GPU world the ring is where a neighbour's border boxes land
  p.binOx = (float)c->desc.tile_origin_x - 1.0f;
  p.binOz = (float)c->desc.tile_origin_z - 1.0f;
  p.invSector = 1.0f / c->desc.sector_size;                 // worldToSector, sc_world_partition.cpp:270
  p.binSX = c->desc.tile_sectors_x ? c->desc.tile_sectors_x + 2u : 0u;
  p.binSZ = c->desc.tile_sectors_x ? c->desc.tile_sectors_z + 2u : 0u;
  p.parity = c->parity;
  p.maxPairs = c->maxPairs;
  p.rankBits = c->rank << 24;
  p.neighbourMask = c->neighbourMask;
  p.variant = c->variant;
  p.chain = std::min(c->maxDepth, kMaxChain);
  p.ovfCap = ovfRecords(c);
  if (c->pairsStream && (flags & SC_TICK_BROADPHASE)) { p.flags |= kFlagDeferredReset; p.resetParity = (c->parity + 1u) % c->pipeDepth; }
  if (flags & SC_TICK_PRODUCE_NEXT) { p.producerKind = c->producerKind; p.producerParam = c->producerParam; }
  p.trafficSmooth = 1.0f - std::exp(-2.5f * c->producerParam);       // smoothExp(current, target, 2.5f, dt), sc_traffic_ai.cpp:58-62, :437
  p.trafficMult = c->trafficMult;
  p.bigCap = c->cap + 8u * kBorderBigCap;
  p.cus = c->cus;
  p.pairRun = pairRunFor(p.binSX * p.binSZ, c->cus, c->variant, false);
  p.borderRecs = c->borderRecs;
  p.halo = c->halo ? 1u : 0u;
  p.tileX = c->tileX; p.tileZ = c->tileZ; p.tilesX = c->tilesX; p.tilesZ = c->tilesZ;
}

// pipelined tiles: a tick refills the bins and counters of its parity, which the pair half of pipeDepth ticks ago read, and
// the tick before it clears those counters.  That half finished long ago unless the exchange is very slow; wait for it.
// (Cross-stream: never captured.)
void waitParityFree(ScTickContext* c, const TickParams& p)
{
  // this tick's end-of-tick kernel clears the NEXT tick's parity (TickParams::resetParity), whose last pair half -- pipeDepth - 1
  // ticks ago -- must be over by then; this tick's own parity was made free one tick ago the same way
  const uint32_t q = p.resetParity;
  if (!(p.flags & kFlagDeferredReset) || !c->pairsInFlight[q]) return;
  // (a queue-to-queue wait costs a bubble of ~10 us on the device even when it is already satisfied: ask first)
  if (hipEventQuery(c->pairsDone[q]) != hipSuccess) {
    // The host is usually ahead of the device here (the flow is device-bound), so the pair half of pipeDepth - 1 ticks ago has often not ended
    // when this tick is issued -- and the queue-to-queue wait then sits in front of this tick's fused kernel on every step: 4-5 us of
    // bubble on the tick stream (tools/trace_timeline.py).  The HOST waits instead (round 4): no tick is issued more than pipeDepth - 1
    // ticks ahead of a finished pair half, the tick stream never sees a barrier, and with the tick before this one already queued the
    // device does not run dry.  (SC_TICK_VARIANT bit 7: the device-side wait of before, for an A/B.)
    if (c->variant & 128u) hipStreamWaitEvent(c->stream, c->pairsDone[q], 0);
    else (void)hipEventSynchronize(c->pairsDone[q]);
  }
  (void)hipGetLastError();                        // hipErrorNotReady from the query is not an error
  c->pairsInFlight[q] = false;
}

// pipelined tiles: whatever is queued on the pairs stream from here on (the exchange) is ordered behind this tick's pack
void publishPacked(ScTickContext* c, const TickParams& p)
{
  if (!c->pairsStream || !(p.flags & SC_TICK_BROADPHASE) || !(p.flags & SC_TICK_SPLIT_PAIRS)) return;
  if (!c->packedRides) hipEventRecord(c->packed[p.parity], c->stream);
  c->packedRides = false;
  hipStreamWaitEvent(c->pairsStream, c->packed[p.parity], 0);
}

// Obstacle rays on a tiled world (sc_traffic_ai.cpp:300-345: the reference's ray sees the whole physics world): with neighbours, the halo
// section in the messages and an in-order step, the rays are cast in the PAIR half behind the merge.  A pipelined tile keeps them in
// the tick half (its own boxes only): the pair half of tick t runs under tick t + 1, whose frame producer reads the brakes.
bool raysInPairHalf(const ScTickContext* c, uint32_t flags)
{
  return c->sensors && c->halo && c->neighbourMask && !c->pairsStream && (flags & SC_TICK_SPLIT_PAIRS) && (flags & SC_TICK_BROADPHASE);
}

// the tick's launches on c->stream, nothing else: safe inside a stream capture (the callers put waitParityFree before
// and publishPacked behind it)
void enqueueStages(ScTickContext* c, const TickParams& p, uint32_t grid, bool allowProfile)
{
  const uint32_t flags = p.flags;
  const DeviceState ds = stateFor(c, p.parity);
  const bool prof = allowProfile && c->profiling;
  const bool saved = c->profiling;
  c->profiling = prof;
  if (c->producerKind && !(flags & SC_TICK_PRODUCE_NEXT)) {
    Scoped s(c, SC_TICK_K_NUDGE);
    if (c->producerKind == 1) launchNudgeRootsX(ds, c->n, c->producerParam, c->stream);
    else launchAdvanceMovers(ds, c->n, c->producerParam, p.trafficSmooth, p.trafficMult, c->stream);
  }
  if (flags & (SC_TICK_XFORM | SC_TICK_CULL | SC_TICK_BROADPHASE)) {
    // the dominant kernel is timed by its own begin / end timestamps (the figure the roofline uses)
    if (c->profiling && (c->profMask & (1u << SC_TICK_K_XFORM_CULL)) && (c->tickIndex % c->profPeriod) == 0) {
      const EventPair ev = takeEvents(c);
      launchXformCull(ds, p, grid, c->stream, ev.a, ev.b);
      c->times[SC_TICK_K_XFORM_CULL].push_back(ev);
    } else { const double q0 = g_probe.on ? probeNow() : 0.0; launchXformCull(ds, p, grid, c->stream); if (g_probe.on) g_probe.acc[3] += probeNow() - q0; }
    if (p.homeMode == kHomeLearn) {
      // the slots handed out by the fused kernel are the remembered ones (the level kernels' and the neighbours' records reserve
      // behind them on every tick); the other copies of the bins start their next tick from the same counts
      launchSnapshotHome(ds, c->sectors, c->n, p.binSX, p.binSZ, (c->pairsStream && c->worldLayersKnown) ? 1u : 0u, c->worldLayers, p.fastPairs, c->stream);
      if (c->pairsStream)
        for (uint32_t q = 0; q < c->pipeDepth; ++q) {
          if (q == p.parity) continue;
          const DeviceState o = stateFor(c, q);
          hipMemcpyAsync(o.binCount, ds.homeCount, (size_t)c->sectors * sizeof(uint32_t), hipMemcpyDeviceToDevice, c->stream);
          hipMemcpyAsync(o.binLayers, ds.homeLayers, (size_t)c->sectors * sizeof(uint32_t), hipMemcpyDeviceToDevice, c->stream);
        }
    }
  }
  if (flags & kFlagHasDeep) {
    for (size_t lv = 0; lv + 1 < c->levelOffsets.size(); ++lv) {
      const uint32_t b = c->levelOffsets[lv], e = c->levelOffsets[lv + 1];
      launchDeepLevel(ds, p, c->dLevelList + b, e - b, c->stream);
    }
  }
  // the traffic AI's obstacle rays (scTickSetTrafficSensors): against this tick's boxes, before the frame producer moves the agents
  // (a tile with neighbours, stepped in order: the rays are cast in the pair half, behind the merge, where the neighbours' boxes are in
  //  the bins too -- raysInPairHalf(); here the agents are only listed with their rays as this frame has them)
  if ((flags & SC_TICK_BROADPHASE) && c->sensors && ds.aLane) {
    if (raysInPairHalf(c, flags)) launchAgentRaySnapshot(ds, p, c->stream);
    else launchAgentFrontRays(ds, p, c->stream);
  }
  if ((flags & SC_TICK_BROADPHASE) && (flags & SC_TICK_DENSE_AABBS)) launchDenseAabbs(ds, c->n, c->stream);   // read-back aid, off the hot path
  const bool needCompact = (flags & (SC_TICK_XFORM | SC_TICK_CULL)) != 0;
  const bool pairsNow = (flags & SC_TICK_BROADPHASE) && !(flags & SC_TICK_SPLIT_PAIRS);
  if (pairsNow && (flags & SC_TICK_RAYS)) launchRayQueries(ds, p, c->rays, c->stream);      // the bins are full, not yet consumed
  // Draw emission rides in the end-of-tick kernel when the order is the plain one: the compaction role knows every visible
  // entity's place in the list, i.e. its draw item (emitVisible).  With the frame read-back on (and a budget that fits the block)
  // the items, the head of the visible list and the header go straight into the block -- no emission kernel, no staging kernel.
  const uint32_t drawBudget = c->desc.max_draws_budget;
  const bool stagedEmit = c->rb.bytes && (flags & SC_TICK_DRAWS) && !(flags & SC_TICK_SORT_DRAWS) && drawBudget && drawBudget <= c->rb.maxDraws;
  const bool foldEmit = needCompact && pairsNow && !(c->variant & (8u | 64u)) && (flags & SC_TICK_CULL) && (flags & SC_TICK_DRAWS) &&
                        !(flags & SC_TICK_SORT_DRAWS) && (stagedEmit || !c->rb.bytes);
  bool stagedByEot = false;
  if (needCompact && pairsNow && !(c->variant & 8u)) {
    TickParams pe = p;
    hipEvent_t done = nullptr;
    if (foldEmit) {
      pe.emitBudget = drawBudget;
      if (stagedEmit) {
        ScTickContext::FrameReadback& rb = c->rb;
        const uint32_t f = (uint32_t)(rb.frames & 1u);
        if (rb.inFlight[f]) {               // the copy of two frames ago still reads this block? (asked first: a satisfied wait costs a bubble too)
          if (hipEventQuery(rb.copied[f]) != hipSuccess) hipStreamWaitEvent(c->stream, rb.copied[f], 0);
          (void)hipGetLastError();
        }
        pe.emitMode = 2u; pe.emitTarget = rb.dBlock[f]; pe.emitMaxVisible = rb.maxVisible;
        pe.emitTickLo = (uint32_t)rb.frames; pe.emitTickHi = (uint32_t)(rb.frames >> 32);
        done = rb.staged[f];
        stagedByEot = true;
      } else { pe.emitMode = 1u; pe.emitTarget = reinterpret_cast<uint32_t*>(c->dDraws); }
    }
    // both depend only on the fused kernel: one launch, workgroups split by role (timed as K_PAIRS, by the dispatch's own
    // begin / end timestamps like the fused kernel: no marker packets on the queue)
    if (c->profiling && (c->profMask & (1u << SC_TICK_K_PAIRS)) && (c->tickIndex % c->profPeriod) == 0) {
      const EventPair ev = takeEvents(c);
      launchCompactPairs(ds, pe, grid, c->stream, ev.a, ev.b);
      c->times[SC_TICK_K_PAIRS].push_back(ev);
      if (done) hipEventRecord(done, c->stream);
    } else launchCompactPairs(ds, pe, grid, c->stream, nullptr, done);      // (`staged` rides on the dispatch: its completion signal)
  } else {
    const bool packToo = needCompact && (flags & SC_TICK_BROADPHASE) && (flags & SC_TICK_SPLIT_PAIRS) && !(c->variant & 8u);
    if (packToo) {
      Scoped s(c, SC_TICK_K_COMPACT);
      // pipelined tile, eager: the `packed` event rides on the dispatch (publishPacked then only makes the pairs stream wait)
      const bool ride = c->pairsStream && !c->capturing && !s.on && (c->variant & 4u) == 0u;
      const double q0 = g_probe.on ? probeNow() : 0.0;
      launchCompactPack(ds, p, grid, c->stream, ride ? c->packed[p.parity] : nullptr);      // compaction and pack share a launch
      if (g_probe.on) g_probe.acc[4] += probeNow() - q0;
      c->packedRides = ride;
    } else if (needCompact) {
      Scoped s(c, SC_TICK_K_COMPACT);
      launchCompact(ds, p, grid, c->stream);
    }
    if (flags & SC_TICK_BROADPHASE) {
      if (flags & SC_TICK_SPLIT_PAIRS) {                                            // the caller exchanges, then scTickRunPairs
        if (!packToo) {
          launchBorderPack(ds, p, c->stream);
          if (flags & kFlagDeferredReset) launchResetParity(ds, p.resetParity, c->stream);      // (the fused launch does it itself)
        }
      }
      else { Scoped s(c, SC_TICK_K_PAIRS); launchPairs(ds, p, c->stream); }
    }
  }
  // (not folded -- sorted draws, split flows, a budget beyond the block: with the frame read-back on and a plain draw list whose
  //  budget fits the block, emission and staging are one launch; else emission, then staging)
  c->lastDraws = c->dDraws;
  if ((flags & SC_TICK_DRAWS) && !stagedEmit && !foldEmit) {
    const uint32_t budget = drawBudget;
    if (flags & SC_TICK_SORT_DRAWS) launchSortedDraws(ds, c->sort, budget, (budget && budget < c->n) ? budget : c->n, c->dDraws, c->stream);
    else launchEmitDraws(ds, budget, c->dDraws, c->stream);
  }
  if (c->rb.bytes) {
    ScTickContext::FrameReadback& rb = c->rb;
    const uint32_t f = (uint32_t)(rb.frames & 1u);
    if (rb.inFlight[f] && !stagedByEot) {  // the copy of two frames ago still reads this block? (asked first: a satisfied wait costs a bubble too)
      if (hipEventQuery(rb.copied[f]) != hipSuccess) hipStreamWaitEvent(c->stream, rb.copied[f], 0);
      (void)hipGetLastError();
    }
    const uint32_t drawMode = (flags & SC_TICK_DRAWS) ? ((flags & SC_TICK_SORT_DRAWS) ? 2u : 1u) : 0u;
    if (stagedByEot) c->lastDraws = rb.dBlock[f] + kFrameHeaderWords + rb.maxVisible;       // the end-of-tick kernel wrote the block
    else if (stagedEmit) {
      launchEmitDrawsStaged(ds, drawBudget, rb.dBlock[f], rb.maxVisible, rb.frames, c->stream, rb.staged[f]);
      c->lastDraws = rb.dBlock[f] + kFrameHeaderWords + rb.maxVisible;       // what scTickReadDraws returns for this tick
    } else launchStageFrame(ds, rb.dBlock[f], rb.maxVisible, rb.maxDraws, c->dDraws, drawMode, rb.frames, c->stream, rb.staged[f]);
    // (`staged` rides on the staging dispatch: its completion signal, no marker packet on the tick queue)
    hipStreamWaitEvent(rb.copyStream, rb.staged[f], 0);
    hipMemcpyAsync(rb.hBlock[f], rb.dBlock[f], rb.bytes, hipMemcpyDeviceToHost, rb.copyStream);
    hipEventRecord(rb.copied[f], rb.copyStream);
    rb.inFlight[f] = true;
    rb.frames++;
  }
  c->profiling = saved;
}

void dropGraph(ScTickContext* c, int q = -1)
{
  for (int k = 0; k < (int)kMaxParity; ++k) {
    if (q >= 0 && k != q) continue;
    if (c->graphExec[k]) { hipGraphExecDestroy(c->graphExec[k]); c->graphExec[k] = nullptr; }
    if (c->graph[k]) { hipGraphDestroy(c->graph[k]); c->graph[k] = nullptr; }
  }
}

void dropPairGraph(ScTickContext* c, int q = -1)
{
  for (int k = 0; k < (int)kMaxParity; ++k) {
    if (q >= 0 && k != q) continue;
    if (c->pairGraphExec[k]) { hipGraphExecDestroy(c->pairGraphExec[k]); c->pairGraphExec[k] = nullptr; }
    if (c->pairGraph[k]) { hipGraphDestroy(c->pairGraph[k]); c->pairGraph[k] = nullptr; }
  }
}

void rowsToMat4(const float* r12, float* m16)
{
  // rows -> column-major Mat4 (m[c*4+r]); last row (0,0,0,1)
  for (int r = 0; r < 3; ++r) for (int col = 0; col < 4; ++col) m16[col * 4 + r] = r12[r * 4 + col];
  m16[3] = 0.0f; m16[7] = 0.0f; m16[11] = 0.0f; m16[15] = 1.0f;
}

} // namespace

extern "C" {

uint32_t scTickGetApiVersion(void) { return SC_TICK_API_VERSION; }

const char* scTickGetLastError(const ScTickContext* ctx) { return ctx ? ctx->err.c_str() : gCreateError.c_str(); }

ScTickContext* scTickCreateContext(const ScTickContextDesc* desc)
{
  if (!desc) { fail(nullptr, "null desc"); return nullptr; }
  if (desc->capacity == 0 || desc->capacity > SC_TICK_MAX_ENTITIES) { fail(nullptr, "capacity must be 1..2^24-1"); return nullptr; }
  int count = 0;
  hipError_t e = hipGetDeviceCount(&count);
  if (e != hipSuccess || count <= 0) { fail(nullptr, "no HIP device available (libsc_tick needs an AMD GPU; there is no CPU fallback)", e); return nullptr; }
  if (desc->device_ordinal < 0 || desc->device_ordinal >= count) { fail(nullptr, "device ordinal out of range"); return nullptr; }

  ScTickContext* c = new ScTickContext();
  c->desc = *desc;
  c->device = desc->device_ordinal;
  if (c->desc.sector_size <= 0.001f) c->desc.sector_size = 64.0f;     // WorldPartition::configure, sc_world_partition.cpp:222-223
  c->cap = ((desc->capacity + kTile - 1) / kTile) * kTile;
  if (const char* s = std::getenv("SC_TICK_SPANS")) { const int v = std::atoi(s); if (v > 0) c->spansWanted = (uint32_t)v; }
  if (const char* s = std::getenv("SC_TICK_VARIANT")) c->variant = (uint32_t)std::atoi(s);

  bool ok = bind(c);
  if (ok) {
    int cu = 0;
    if (hipDeviceGetAttribute(&cu, hipDeviceAttributeMultiprocessorCount, c->device) == hipSuccess && cu > 0) c->cus = (uint32_t)cu;
    else c->cus = 256u;                                   // MI355X
  }
  if (ok) {
    // (keeping compute units out of the tick stream's reach with a CU mask, so that the RCCL kernel of the pair half finds
    //  free ones at once, was measured: 124-232 us per step against 89 -- masked queues schedule badly here)
    e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e != hipSuccess) ok = fail(c, "hipStreamCreate", e);
    c->ownStream = c->stream;
  }
  DeviceState& d = c->d;
  const size_t N = c->cap;
  // one slab for every 4-byte stream, one for the matrix rows (see Stream in sc_tick_internal.h)
  float* fslab = nullptr; float4* rslab = nullptr;
  ok = ok && dalloc(c, fslab, (size_t)kStreamCount * N) && dalloc(c, rslab, 3 * N);
  if (ok) {
    auto F = [&](uint32_t k) { return fslab + (size_t)k * N; };
    d.fslab = reinterpret_cast<const char*>(fslab); d.rslab = reinterpret_cast<char*>(rslab);
    d.capBytes = (uint32_t)(N * 4u); d.capBytes16 = (uint32_t)(N * 16u);
    d.px = F(kPX); d.py = F(kPY); d.pz = F(kPZ);
    d.rsx = F(kRSX); d.rcx = F(kRCX); d.rsy = F(kRSY); d.rcy = F(kRCY); d.rsz = F(kRSZ); d.rcz = F(kRCZ);
    d.sx = F(kSX); d.sy = F(kSY); d.sz = F(kSZ);
    d.bminx = F(kBMINX); d.bminy = F(kBMINY); d.bminz = F(kBMINZ); d.bmaxx = F(kBMAXX); d.bmaxy = F(kBMAXY); d.bmaxz = F(kBMAXZ);
    d.link = reinterpret_cast<uint32_t*>(F(kLINK)); d.layers = reinterpret_cast<uint32_t*>(F(kLAYERS));
    d.meshId = reinterpret_cast<uint32_t*>(F(kMESH)); d.materialId = reinterpret_cast<uint32_t*>(F(kMATERIAL));
    d.w0 = rslab; d.w1 = rslab + N; d.w2 = rslab + 2 * N;
  }
  ok = ok && dalloc(c, d.dirty, N / 32) && dalloc(c, d.unreach, N / 32)
          && dalloc(c, d.vis, N / 64) && dalloc(c, d.cand, N / 64) && dalloc(c, d.recomp, N / 64)
          && dalloc(c, d.blockVis, N / kTile) && dalloc(c, d.blockCand, N / kTile)
          && dalloc(c, d.visibleIdx, N) && dalloc(c, d.culledIdx, N) && dalloc(c, d.counters, kCounterWords)
          && dalloc(c, d.aabbMin, N) && dalloc(c, d.aabbMax, N);
  { float* f = nullptr; ok = ok && dalloc(c, f, 32); d.frustum = f; }
  c->sectors = desc->tile_sectors_x ? (desc->tile_sectors_x + 2u) * (desc->tile_sectors_z + 2u) : 0u;
  c->maxPairs = desc->max_pairs ? desc->max_pairs : desc->capacity * 4u;
  c->maxPairs = ((c->maxPairs + kPairShards - 1u) / kPairShards) * kPairShards;   // equal shard segments
  if (ok && c->sectors) {
    if ((uint64_t)desc->tile_sectors_x * desc->tile_sectors_z > (1u << 24) || desc->tile_sectors_x > 65533u || desc->tile_sectors_z > 65533u)
      ok = fail(c, "tile rectangle too large");           // (the pair search carries a sector's grid coordinates as 16 + 16 bits)
    ok = ok && dalloc(c, d.binCount, c->sectors) && dalloc(c, d.binLayers, c->sectors) && dalloc(c, d.bins, (size_t)c->sectors * kBinCap * 2u, false)
            && dalloc(c, d.bigList, (N + 8u * kBorderBigCap) * 2u, false) && dalloc(c, d.spill, 2u * (size_t)ovfRecords(c), false) && dalloc(c, d.spillSector, ovfRecords(c))
            && dalloc(c, d.crowdQueue, (size_t)kMaxParity * c->sectors)
            && dalloc(c, d.ovfIdx, (size_t)kOvfWaves * kOvfPerSector, false) && dalloc(c, d.ovfLo, c->sectors, false) && dalloc(c, d.ovfHi, c->sectors)
            && dalloc(c, d.pairs, c->maxPairs, false) && dalloc(c, d.pairShardCount, (kMaxParity + 1u) * kPairShards * kShardStride)
            && dalloc(c, c->dPairsOut, c->maxPairs, false) && dalloc(c, c->dPairTotal, 4)
            && dalloc(c, d.homeA, N, false) && dalloc(c, d.homeB, N, false) && dalloc(c, d.homeCount, c->sectors) && dalloc(c, d.homeLayers, c->sectors) && dalloc(c, d.lazyCtl, 1u + 2u * kMaxParity)
            && dalloc(c, d.homeCast, c->sectors) && dalloc(c, d.homePerm, (size_t)c->sectors * kBinCap);
    if (ok) { e = hipMemset(d.homeA, 0xFF, N * sizeof(uint32_t)); if (e == hipSuccess) e = hipMemset(d.homeB, 0xFF, N * sizeof(uint32_t)); if (e != hipSuccess) ok = fail(c, "hipMemset", e); }
  }
  if (ok && c->sectors) {
    float4* nr = nullptr;
    ok = dalloc(c, nr, 2);
    if (ok) {
      const uint32_t inf = 0x7F800000u, ninf = 0xFF800000u;
      const uint32_t words[8] = { inf, inf, inf, 0u, ninf, ninf, ninf, 0x00FFFFFFu };      // nullRecord() in sc_tick_kernels.hip
      ok = h2d(c, nr, words, sizeof words) && sync(c);
      d.nullRec = nr;
    }
    // the pair role's constant tables (sc_tick_kernels.hip: pairsBody's prologue copies them into LDS)
    uint32_t* pc = nullptr;
    ok = ok && dalloc(c, pc, kPairConstWords);
    if (ok) {
      std::vector<uint32_t> words(kPairConstWords, 0u);
      uint16_t* tab = reinterpret_cast<uint16_t*>(words.data());
      for (uint32_t i = 1; i < kBinCap; ++i)
        for (uint32_t j = 0; j < i; ++j) tab[i * (i - 1u) / 2u + j] = (uint16_t)(i << 8 | j);
      for (uint32_t t = 1; t <= kBinCap; ++t) { const uint32_t G = 64u / t; words[kPairConstCast + t] = G | ((65536u / G + 1u) << 8); }
      ok = h2d(c, pc, words.data(), words.size() * sizeof(uint32_t)) && sync(c);
      d.pairConst = pc;
    }
  }
  if (c->variant & 32u) c->lazyEnabled = false;         // SC_TICK_VARIANT bit 5: every remembered slot is written on every tick (A/B)
  if (c->variant & 2u) c->homeEnabled = false;          // SC_TICK_VARIANT bit 1: every record reserves its slot on every tick (A/B)
  if (std::getenv("SC_TICK_HOSTPROBE")) g_probe.on = true;
  if (const char* fp = std::getenv("SC_TICK_FAST_PAIRS")) c->fastPairs = std::atoi(fp) != 0;      // 0: every bin goes through the general pair search (A/B)
  if (const char* hp = std::getenv("SC_TICK_HOME_PERIOD")) { const int v = std::atoi(hp); if (v > 0) c->homePeriod = (uint32_t)v; }
  if (ok && c->sectors) { e = hipMemset(d.ovfLo, 0xFF, (size_t)c->sectors * sizeof(uint32_t)); if (e != hipSuccess) ok = fail(c, "hipMemset", e); }
  if (ok) { void* p = nullptr; e = hipMalloc(&p, N * sizeof(ScTickDrawItem)); if (e != hipSuccess) ok = fail(c, "hipMalloc draws", e); else { c->allocs.push_back(p); c->dDraws = p; } }
  if (ok) {
    // Transform{}: worldMatrix = identity, scale = 1, cos = 1 (sc_ecs.h:63-71)
    std::vector<float> ones(N, 1.0f);
    std::vector<float4> e0(N, make_float4(1, 0, 0, 0)), e1(N, make_float4(0, 1, 0, 0)), e2(N, make_float4(0, 0, 1, 0));
    ok = h2d(c, d.sx, ones.data(), N * 4) && h2d(c, d.sy, ones.data(), N * 4) && h2d(c, d.sz, ones.data(), N * 4)
      && h2d(c, d.rcx, ones.data(), N * 4) && h2d(c, d.rcy, ones.data(), N * 4) && h2d(c, d.rcz, ones.data(), N * 4)
      && h2d(c, d.w0, e0.data(), N * 16) && h2d(c, d.w1, e1.data(), N * 16) && h2d(c, d.w2, e2.data(), N * 16) && sync(c);
  }
  if (!ok) { gCreateError = c->err; scTickDestroyContext(c); return nullptr; }
  c->hParent.assign(desc->capacity, SC_TICK_NO_PARENT);
  c->hFlags.assign(desc->capacity, 0);
  c->hChildren.assign(desc->capacity, 0);
  c->hFirstChild.assign(desc->capacity, -1); c->hNextSib.assign(desc->capacity, -1); c->hPrevSib.assign(desc->capacity, -1);
  return c;
}

void scTickDestroyContext(ScTickContext* c)
{
  if (g_probe.on && g_probe.n) {
    std::fprintf(stderr, "[sc_tick host probe] %llu ticks: waitParityFree %.2f us, enqueueStages %.2f (fused launch %.2f, compaction+pack launch %.2f), publishPacked %.2f\n", (unsigned long long)g_probe.n,
                 g_probe.acc[0] / g_probe.n, g_probe.acc[1] / g_probe.n, g_probe.acc[3] / g_probe.n, g_probe.acc[4] / g_probe.n, g_probe.acc[2] / g_probe.n);
    g_probe = HostProbe(); g_probe.on = true;
  }
  if (!c) return;
  hipSetDevice(c->device);
  if (c->stream) hipStreamSynchronize(c->stream);
  if (c->pairsStream) hipStreamSynchronize(c->pairsStream);
  c->stream = c->ownStream;
  dropGraph(c); dropPairGraph(c);
  for (auto& v : c->times) for (auto& p : v) { hipEventDestroy(p.a); hipEventDestroy(p.b); }
  for (auto& p : c->eventPool) { hipEventDestroy(p.a); hipEventDestroy(p.b); }
  for (uint32_t k = 0; k < kMaxParity; ++k) { if (c->packed[k]) hipEventDestroy(c->packed[k]); if (c->pairsDone[k]) hipEventDestroy(c->pairsDone[k]); }
  if (c->rb.copyStream) {
    hipStreamSynchronize(c->rb.copyStream);
    for (int k = 0; k < 2; ++k) { if (c->rb.hBlock[k]) hipHostFree(c->rb.hBlock[k]); if (c->rb.staged[k]) hipEventDestroy(c->rb.staged[k]); if (c->rb.copied[k]) hipEventDestroy(c->rb.copied[k]); }
    hipStreamDestroy(c->rb.copyStream);
  }
  if (c->pairsStream) hipStreamSynchronize(c->pairsStream);
  if (c->comm) { std::string why; if (const RcclApi* r = rccl(&why)) r->CommDestroy(c->comm); c->comm = nullptr; }
  for (void* p : c->allocs) hipFree(p);
  if (c->ownPairsStream) hipStreamDestroy(c->ownPairsStream);
  if (c->stream) hipStreamDestroy(c->stream);
  delete c;
}

int scTickSetEntityCount(ScTickContext* c, uint32_t count)
{
  if (!c) return 0;
  if (count > c->desc.capacity) return fail(c, "count exceeds capacity");
  c->n = count;
  c->linksStale = true;
  return 1;
}

int scTickUploadLocals(ScTickContext* c, uint32_t first, uint32_t count, const float* pos3, const float* rot3,
                       const float* scale3, uint8_t* repaired)
{
  if (!c || !pos3 || !rot3 || !scale3) return c ? fail(c, "null argument") : 0;
  if (!bind(c) || !rangeOk(c, first, count)) return 0;
  if (!count) return 1;
  std::vector<float> s[6], k(3 * (size_t)count);
  for (auto& v : s) v.resize(count);
  for (uint32_t i = 0; i < count; ++i) {
    // host libm, float overloads -- what std::sin/std::cos resolve to in mat4_rotation_xyz
    s[0][i] = std::sin(rot3[3 * i]);     s[1][i] = std::cos(rot3[3 * i]);
    s[2][i] = std::sin(rot3[3 * i + 1]); s[3][i] = std::cos(rot3[3 * i + 1]);
    s[4][i] = std::sin(rot3[3 * i + 2]); s[5][i] = std::cos(rot3[3 * i + 2]);
    float a = scale3[3 * i], b = scale3[3 * i + 1], z = scale3[3 * i + 2];
    const bool rep = (a == 0.0f && b == 0.0f && z == 0.0f);          // sc_ecs.cpp:143-149
    if (rep) { a = b = z = 1.0f; }
    if (repaired) repaired[i] = rep ? 1 : 0;
    k[3 * (size_t)i] = a; k[3 * (size_t)i + 1] = b; k[3 * (size_t)i + 2] = z;
    // an axis whose sin/cos came out as exactly (0, 1) need not be streamed by the kernels (link-word flag)
    const uint8_t triv = (uint8_t)(((s[0][i] == 0.0f && s[1][i] == 1.0f) ? 4u : 0u) | ((s[2][i] == 0.0f && s[3][i] == 1.0f) ? 8u : 0u) |
                                   ((s[4][i] == 0.0f && s[5][i] == 1.0f) ? 16u : 0u));
    uint8_t& f = c->hFlags[first + i];
    const uint8_t want = (f & 32u) ? (uint8_t)(triv & ~8u) : triv;       // traffic agents: the device rewrites their yaw -- Y is always streamed
    if ((f & 28u) != want) { f = (uint8_t)((f & ~28u) | want); c->linksStale = true; }
  }
  DeviceState& d = c->d;
  float* dst[6] = { d.rsx, d.rcx, d.rsy, d.rcy, d.rsz, d.rcz };
  for (int q = 0; q < 6; ++q) if (!h2d(c, dst[q] + first, s[q].data(), (size_t)count * 4u)) return 0;
  if (!upload3(c, pos3, first, count, d.px, d.py, d.pz)) return 0;
  if (!upload3(c, k.data(), first, count, d.sx, d.sy, d.sz)) return 0;
  launchSetDirtyRange(d, first, count, c->stream);
  return sync(c) ? 1 : 0;
}

int scTickUploadPositions(ScTickContext* c, uint32_t first, uint32_t count, const float* pos3)
{
  if (!c || !pos3) return c ? fail(c, "null argument") : 0;
  if (!bind(c) || !rangeOk(c, first, count)) return 0;
  if (!count) return 1;
  if (!upload3(c, pos3, first, count, c->d.px, c->d.py, c->d.pz)) return 0;
  launchSetDirtyRange(c->d, first, count, c->stream);
  return sync(c) ? 1 : 0;
}

int scTickUploadBounds(ScTickContext* c, uint32_t first, uint32_t count, const float* min3, const float* max3, const uint8_t* has)
{
  if (!c || !min3 || !max3) return c ? fail(c, "null argument") : 0;
  if (!bind(c) || !rangeOk(c, first, count)) return 0;
  if (!count) return 1;
  c->boxesTouched = true;
  if (!upload3(c, min3, first, count, c->d.bminx, c->d.bminy, c->d.bminz)) return 0;
  if (!upload3(c, max3, first, count, c->d.bmaxx, c->d.bmaxy, c->d.bmaxz)) return 0;
  for (uint32_t i = 0; i < count; ++i) {
    uint8_t& f = c->hFlags[first + i];
    const uint8_t nf = (uint8_t)((f & ~2u) | ((!has || has[i]) ? 2u : 0u));
    if (nf != f) { f = nf; c->linksStale = true; }          // link words only change when membership does
  }
  return 1;
}

int scTickUploadRenderMeshes(ScTickContext* c, uint32_t first, uint32_t count, const uint8_t* has, const uint32_t* mesh, const uint32_t* material)
{
  if (!c) return 0;
  if (!bind(c) || !rangeOk(c, first, count)) return 0;
  if (!count) return 1;
  if (mesh && !h2d(c, c->d.meshId + first, mesh, (size_t)count * 4u)) return 0;
  if (material && !h2d(c, c->d.materialId + first, material, (size_t)count * 4u)) return 0;
  if (!sync(c)) return 0;
  for (uint32_t i = 0; i < count; ++i) {
    uint8_t& f = c->hFlags[first + i];
    const uint8_t nf = (uint8_t)((f & ~1u) | ((!has || has[i]) ? 1u : 0u));
    if (nf != f) { f = nf; c->linksStale = true; }
  }
  return 1;
}

int scTickUploadLayers(ScTickContext* c, uint32_t first, uint32_t count, const uint32_t* group, const uint32_t* mask)
{
  if (!c || !group || !mask) return c ? fail(c, "null argument") : 0;
  if (!bind(c) || !rangeOk(c, first, count)) return 0;
  if (!count) return 1;
  std::vector<uint32_t> packed(count);
  for (uint32_t i = 0; i < count; ++i) {
    const uint32_t g = group[i], m = mask[i];
    if ((g != 0xFFFFFFFFu && (g >> 16)) || (m != 0xFFFFFFFFu && (m >> 16))) return fail(c, "group/mask bits above 15 are not supported");
    packed[i] = (g & 0xFFFFu) | ((m & 0xFFFFu) << 16);
    // the declared layer vocabulary of the tiled world is a contract (scTickSetWorldLayers): a collider outside it could meet bins that
    // this or another tile leaves unwritten for good -- refused here instead of missing pairs silently there
    if (c->worldLayersKnown && (packed[i] & ~c->worldLayers))
      return fail(c, "layers outside the declared world vocabulary (scTickSetWorldLayers): declare the wider vocabulary on every tile first");
  }
  if (!h2d(c, c->d.layers + first, packed.data(), (size_t)count * 4u)) return 0;
  if (c->hLayers.size() < (size_t)first + count) c->hLayers.resize((size_t)first + count, 0u);
  std::copy(packed.begin(), packed.end(), c->hLayers.begin() + first);      // (host mirror: what a later, narrower vocabulary is checked against)
  c->layerSetStale = true;
  c->homeValid = false;                 // the bins' remembered layer summaries are behind
  return sync(c) ? 1 : 0;
}

int scTickSetTopology(ScTickContext* c, const int32_t* parent, uint32_t count)
{
  if (!c || (!parent && count)) return c ? fail(c, "null argument") : 0;
  if (count != c->n) return fail(c, "topology must cover every entity (count != entity count)");
  if (!bind(c)) return 0;
  std::copy(parent, parent + count, c->hParent.begin());
  c->linksStale = true;
  return flushLinks(c);
}

// ---- sector residency -------------------------------------------------------------------------------
static uint32_t linkWordOfRoot(uint8_t f)
{
  return kNoParent | ((f & 1u) ? kHasMesh : 0u) | ((f & 2u) ? kHasBounds : 0u) |
         ((f & 4u) ? kRotTrivialX : 0u) | ((f & 8u) ? kRotTrivialY : 0u) | ((f & 16u) ? kRotTrivialZ : 0u);
}

int scTickAppendEntities(ScTickContext* c, uint32_t count, const float* pos3, const float* rot3, const float* scale3,
                         const float* bmin3, const float* bmax3, const uint32_t* mesh, const uint32_t* material,
                         const uint32_t* group, const uint32_t* mask, const int32_t* parent, uint32_t* firstOut)
{
  if (!c) return 0;
  if (count && (!pos3 || !rot3 || !scale3)) return fail(c, "null argument");
  if ((bmin3 == nullptr) != (bmax3 == nullptr) || (group == nullptr) != (mask == nullptr)) return fail(c, "bounds / layers come in pairs");
  if (!bind(c)) return 0;
  const uint32_t first = c->n;
  if ((uint64_t)first + count > c->desc.capacity) return fail(c, "append exceeds capacity");
  if (firstOut) *firstOut = first;
  if (!count) return 1;
  if (parent) for (uint32_t i = 0; i < count; ++i)
    if (parent[i] != SC_TICK_NO_PARENT && (parent[i] < 0 || (uint32_t)parent[i] >= first + count)) return fail(c, "parent index out of range");

  const bool wasStale = c->linksStale;
  c->n = first + count;
  for (uint32_t i = first; i < first + count; ++i) {
    c->hFlags[i] = 0; c->hParent[i] = SC_TICK_NO_PARENT; c->hChildren[i] = 0;
    c->hFirstChild[i] = -1; c->hNextSib[i] = -1; c->hPrevSib[i] = -1;
  }

  std::vector<float> cube;
  if (!bmin3) { cube.assign((size_t)count * 6, 0.5f); for (size_t i = 0; i < (size_t)count * 3; ++i) cube[i] = -0.5f; }   // kUnitCubeBounds
  std::vector<uint32_t> all;
  if (!group) all.assign(count, 0xFFFFFFFFu);
  std::vector<uint32_t> zeros;
  if (!mesh || !material) zeros.assign(count, 0u);
  const bool ok =
      scTickUploadLocals(c, first, count, pos3, rot3, scale3, nullptr) &&
      scTickUploadBounds(c, first, count, bmin3 ? bmin3 : cube.data(), bmax3 ? bmax3 : cube.data() + (size_t)count * 3, nullptr) &&
      scTickUploadRenderMeshes(c, first, count, nullptr, mesh ? mesh : zeros.data(), material ? material : zeros.data()) &&
      scTickUploadLayers(c, first, count, group ? group : all.data(), mask ? mask : all.data());
  if (!ok) { c->n = first; c->linksStale = true; return 0; }
  for (uint32_t i = first; i < first + count; ++i) c->hFlags[i] &= (uint8_t)~32u;
  if (c->d.moverKind) {       // an appended entity is no mover until scTickUploadMovers says so
    const hipError_t e = hipMemsetAsync(c->d.moverKind + first, 0, (size_t)count * 4u, c->stream);
    if (e != hipSuccess) return fail(c, "hipMemsetAsync", e);
  }

  if (parent || wasStale) {
    if (parent) std::copy(parent, parent + count, c->hParent.begin() + first);
    c->linksStale = true;
    return flushLinks(c);
  }
  // roots only and the rest of the hierarchy untouched: write just the new link words
  std::vector<uint32_t> link(count);
  for (uint32_t i = 0; i < count; ++i) link[i] = linkWordOfRoot(c->hFlags[first + i]);
  if (!h2d(c, c->d.link + first, link.data(), (size_t)count * 4u) || !sync(c)) { c->linksStale = true; return 0; }
  c->linksStale = false;
  c->topoEpoch++;
  return 1;
}

int scTickRemoveEntities(ScTickContext* c, const uint32_t* idx, uint32_t count, uint32_t* movedFrom, uint32_t* movedTo, uint32_t* movedCount)
{
  if (!c) return 0;
  if (movedCount) *movedCount = 0;
  if (!idx && count) return fail(c, "null argument");
  if (!bind(c)) return 0;
  if (!count) return 1;
  const uint32_t n0 = c->n;
  if (count > n0) return fail(c, "more removals than entities");

  // Replay the swap-removes on an index map that only holds the slots they touch.
  std::unordered_map<uint32_t, uint32_t> occupant;   // slot -> original index of the entity now in it
  std::unordered_map<uint32_t, uint32_t> where;      // original index -> slot it was moved to
  std::unordered_set<uint32_t> removed;
  occupant.reserve(count * 2u); where.reserve(count * 2u); removed.reserve(count * 2u);
  uint32_t size = n0;
  for (uint32_t k = 0; k < count; ++k) {
    const uint32_t r = idx[k];
    if (r >= n0) return fail(c, "dense index out of range");
    if (!removed.insert(r).second) return fail(c, "dense index listed twice");
    const auto w = where.find(r);
    const uint32_t slot = w == where.end() ? r : w->second;
    const uint32_t last = size - 1u;
    const auto o = occupant.find(last);
    const uint32_t lastOrig = o == occupant.end() ? last : o->second;
    if (slot != last) { occupant[slot] = lastOrig; where[lastOrig] = slot; }
    size = last;
  }
  const uint32_t n1 = size;
  std::vector<uint32_t> src, dst;
  for (const auto& kv : occupant) if (kv.first < n1 && kv.second != kv.first) { dst.push_back(kv.first); src.push_back(kv.second); }
  // deterministic order for the caller
  {
    std::vector<uint32_t> order(dst.size());
    for (uint32_t i = 0; i < order.size(); ++i) order[i] = i;
    std::sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return dst[a] < dst[b]; });
    std::vector<uint32_t> s2(src.size()), d2(dst.size());
    for (uint32_t i = 0; i < order.size(); ++i) { s2[i] = src[order[i]]; d2[i] = dst[order[i]]; }
    src.swap(s2); dst.swap(d2);
  }
  const uint32_t moves = (uint32_t)src.size();

  std::unordered_map<uint32_t, uint32_t> newIndexOf;
  newIndexOf.reserve(moves * 2u);
  for (uint32_t k = 0; k < moves; ++k) newIndexOf[src[k]] = dst[k];
  auto remap = [&](int32_t i) -> int32_t {          // index before the call -> index after it (sources are >= n1, targets < n1)
    if (i < 0) return i;
    const auto m = newIndexOf.find((uint32_t)i);
    return m == newIndexOf.end() ? i : (int32_t)m->second;
  };

  // The hierarchy is patched in place unless an index sits in a level list or a cycle, or a removed entity
  // leaves a child behind (orphans change the depth of whole subtrees: full re-link, sc_ecs.cpp:151-160).
  bool relink = c->linksStale || c->unreachable > 0 || c->maxDepth > kMaxChain;
  for (uint32_t k = 0; k < count && !relink; ++k)
    for (int32_t ch = c->hFirstChild[idx[k]]; ch >= 0 && !relink; ch = c->hNextSib[(uint32_t)ch]) relink = !removed.count((uint32_t)ch);

  // device: relocate every per-entity array
  if (moves) {
    if (!needScratch(c, 2u * (size_t)moves)) return 0;
    if (!h2d(c, c->dIdx, src.data(), (size_t)moves * 4u) || !h2d(c, c->dIdx + moves, dst.data(), (size_t)moves * 4u)) return 0;
    launchMoveEntities(c->d, c->dIdx, c->dIdx + moves, moves, c->stream);
    if (!sync(c)) return 0;                          // the scratch buffer is reused for the parent patches below
    for (uint32_t k = 0; k < moves; ++k)             // (the layer words' host mirror moves along: sources lie beyond every target)
      if (src[k] < c->hLayers.size()) { if (c->hLayers.size() <= dst[k]) c->hLayers.resize((size_t)dst[k] + 1u, 0u); c->hLayers[dst[k]] = c->hLayers[src[k]]; }
    c->layerSetStale = true;
  }

  std::vector<uint32_t> patch;                       // (entity, new parent) pairs for the device link words
  if (!relink) {
    std::vector<int32_t>&par = c->hParent, &fc = c->hFirstChild, &ns = c->hNextSib, &ps = c->hPrevSib;
    // 1. unlink the removed entities from surviving parents (indices as before the call)
    for (uint32_t k = 0; k < count; ++k) {
      const uint32_t r = idx[k];
      const int32_t q = par[r];
      if (q < 0 || removed.count((uint32_t)q)) continue;
      if (ps[r] >= 0) ns[(uint32_t)ps[r]] = ns[r]; else fc[(uint32_t)q] = ns[r];
      if (ns[r] >= 0) ps[(uint32_t)ns[r]] = ps[r];
      c->hChildren[(uint32_t)q]--;
    }
    // 2. gather every write a relocation needs, reading only the records as they stand (sources are never written)
    struct Set { std::vector<int32_t>* arr; uint32_t at; int32_t value; };
    std::vector<Set> sets;
    for (uint32_t k = 0; k < moves; ++k) {
      const uint32_t s0 = src[k], d0 = dst[k];
      sets.push_back({ &par, d0, remap(par[s0]) });
      sets.push_back({ &fc, d0, remap(fc[s0]) });
      sets.push_back({ &ns, d0, remap(ns[s0]) });
      sets.push_back({ &ps, d0, remap(ps[s0]) });
      if (ps[s0] >= 0) sets.push_back({ &ns, (uint32_t)remap(ps[s0]), (int32_t)d0 });
      else if (par[s0] >= 0) sets.push_back({ &fc, (uint32_t)remap(par[s0]), (int32_t)d0 });
      if (ns[s0] >= 0) sets.push_back({ &ps, (uint32_t)remap(ns[s0]), (int32_t)d0 });
      for (int32_t ch = fc[s0]; ch >= 0; ch = ns[(uint32_t)ch]) {
        const uint32_t at = (uint32_t)remap(ch);
        sets.push_back({ &par, at, (int32_t)d0 });
        patch.push_back(at); patch.push_back(d0);
      }
      c->hFlags[d0] = c->hFlags[s0];
      c->hChildren[d0] = c->hChildren[s0];
    }
    // 3. apply (writes that meet on one field carry the same value)
    for (const Set& w : sets) (*w.arr)[w.at] = w.value;
    if (!patch.empty()) {
      const uint32_t pairs = (uint32_t)(patch.size() / 2);
      if (!needScratch(c, patch.size())) return 0;
      if (!h2d(c, c->dIdx, patch.data(), patch.size() * 4u)) return 0;
      launchPatchParents(c->d, c->dIdx, pairs, c->stream);
    }
  } else {
    // parents are rewritten through the relocation map; a child of a removed entity gets an invalid parent,
    // which the re-link detaches and marks dirty (sc_ecs.cpp:151-160)
    for (uint32_t k = 0; k < moves; ++k) { c->hParent[dst[k]] = c->hParent[src[k]]; c->hFlags[dst[k]] = c->hFlags[src[k]]; }
    constexpr int32_t kGone = -2;
    for (uint32_t i = 0; i < n1; ++i) {
      const int32_t q = c->hParent[i];
      if (q == SC_TICK_NO_PARENT) continue;
      if (removed.count((uint32_t)q)) c->hParent[i] = kGone;
      else c->hParent[i] = remap(q);
    }
    c->linksStale = true;
  }
  c->n = n1;
  c->topoEpoch++;
  if (movedFrom && movedTo) { std::copy(src.begin(), src.end(), movedFrom); std::copy(dst.begin(), dst.end(), movedTo); }
  if (movedCount) *movedCount = moves;
  if (relink) return flushLinks(c);
  return sync(c) ? 1 : 0;
}

int scTickMarkDirty(ScTickContext* c, uint32_t first, uint32_t count)
{
  if (!c) return 0;
  if (!bind(c) || !rangeOk(c, first, count)) return 0;
  launchSetDirtyRange(c->d, first, count, c->stream);
  return 1;
}

int scTickMarkDirtyIndices(ScTickContext* c, const uint32_t* idx, uint32_t count)
{
  if (!c || (!idx && count)) return c ? fail(c, "null argument") : 0;
  if (!bind(c)) return 0;
  if (!count) return 1;
  for (uint32_t i = 0; i < count; ++i) if (idx[i] >= c->n) return fail(c, "dense index out of range");
  if (!needScratch(c, count)) return 0;
  if (!h2d(c, c->dIdx, idx, (size_t)count * 4u)) return 0;
  launchSetDirtyIndices(c->d, c->dIdx, count, c->stream);
  return sync(c) ? 1 : 0;
}

int scTickSetDirtyFlags(ScTickContext* c, uint32_t first, uint32_t count, const uint8_t* flags)
{
  if (!c || !flags) return c ? fail(c, "null argument") : 0;
  if (!bind(c) || !rangeOk(c, first, count)) return 0;
  if (!count) return 1;
  // read-modify-write of the covered words on the host: exact flags for [first, first+count)
  const uint32_t w0 = first >> 5, w1 = (first + count + 31u) >> 5;
  std::vector<uint32_t> words(w1 - w0);
  if (!d2h(c, words.data(), c->d.dirty + w0, words.size() * 4u) || !sync(c)) return 0;
  for (uint32_t i = 0; i < count; ++i) {
    const uint32_t g = first + i, bit = 1u << (g & 31u);
    uint32_t& w = words[(g >> 5) - w0];
    w = flags[i] ? (w | bit) : (w & ~bit);
  }
  if (!h2d(c, c->d.dirty + w0, words.data(), words.size() * 4u)) return 0;
  return sync(c) ? 1 : 0;
}

int scTickSetDrawSortTable(ScTickContext* c, const uint8_t* pipelineOfMaterial, uint32_t materialCount, uint32_t meshCount)
{
  if (!c) return 0;
  if (!pipelineOfMaterial && materialCount) return fail(c, "null argument");
  if (materialCount > (1u << 24) || meshCount > (1u << 24)) return fail(c, "more than 2^24 material or mesh handles");
  for (uint32_t i = 0; i < materialCount; ++i)
    if (pipelineOfMaterial[i] >= 128u && pipelineOfMaterial[i] != kNoMaterial) return fail(c, "pipeline ids must be < 128 (0xFF = no such material)");
  if (!bind(c) || !sync(c)) return 0;
  DrawSortState& st = c->sort;
  if (!st.key[0]) {
    const size_t N = c->cap;
    const size_t groups = (N + kSortGroup - 1) / kSortGroup;
    if (!dalloc(c, st.key[0], N, false) || !dalloc(c, st.key[1], N, false) || !dalloc(c, st.idx[0], N, false) ||
        !dalloc(c, st.idx[1], N, false) || !dalloc(c, st.hist, 256u * groups, false)) return 0;
  }
  if (c->pipelineCap < std::max(materialCount, 1u)) {
    dfree(c, c->dPipeline); c->dPipeline = nullptr; c->pipelineCap = 0;
    const uint32_t want = std::max(materialCount, 64u);
    if (!dalloc(c, c->dPipeline, want, false)) return 0;
    c->pipelineCap = want;
  }
  if (materialCount && (!h2d(c, c->dPipeline, pipelineOfMaterial, materialCount) || !sync(c))) return 0;
  st.pipeline = c->dPipeline;
  st.materialCount = materialCount; st.meshCount = meshCount;
  // key = pipeline << 48 | material << 24 | mesh: sort only over the bytes the handle ranges can reach,
  // and always over the top byte (pipeline + the "dropped" mark)
  auto bytesOf = [](uint32_t count) { uint32_t top = count > 1u ? count - 1u : 0u, b = 0; while (top) { ++b; top >>= 8; } return b; };
  st.passes = 0;
  for (uint32_t b = 0; b < bytesOf(meshCount); ++b) st.shift[st.passes++] = 8u * b;
  for (uint32_t b = 0; b < bytesOf(materialCount); ++b) st.shift[st.passes++] = 24u + 8u * b;
  st.shift[st.passes++] = 48u;
  c->topoEpoch++;                      // captured graphs hold the old pass list
  return 1;
}

int scTickSetDrawBudget(ScTickContext* c, uint32_t maxDraws)
{
  if (!c) return 0;
  c->desc.max_draws_budget = maxDraws;
  return 1;
}

int scTickUploadWorldMatrices(ScTickContext* c, uint32_t first, uint32_t count, const float* m16)
{
  if (!c || !m16) return c ? fail(c, "null argument") : 0;
  if (!bind(c) || !rangeOk(c, first, count)) return 0;
  if (!count) return 1;
  c->boxesTouched = true;
  std::vector<float4> r0(count), r1(count), r2(count);
  for (uint32_t i = 0; i < count; ++i) {
    const float* m = m16 + 16 * (size_t)i;
    if (m[3] != 0.0f || m[7] != 0.0f || m[11] != 0.0f || m[15] != 1.0f) return fail(c, "world matrix is not affine (row 3 must be 0,0,0,1)");
    r0[i] = make_float4(m[0], m[4], m[8], m[12]);
    r1[i] = make_float4(m[1], m[5], m[9], m[13]);
    r2[i] = make_float4(m[2], m[6], m[10], m[14]);
  }
  if (!h2d(c, c->d.w0 + first, r0.data(), (size_t)count * 16u) || !h2d(c, c->d.w1 + first, r1.data(), (size_t)count * 16u) ||
      !h2d(c, c->d.w2 + first, r2.data(), (size_t)count * 16u)) return 0;
  return sync(c) ? 1 : 0;
}

int scTickSetFrustumPlanes(ScTickContext* c, const float planes[24], int valid)
{
  if (!c || !planes) return c ? fail(c, "null argument") : 0;
  std::memcpy(c->frustum.p, planes, sizeof c->frustum.p);
  c->frustumValid = valid ? 1 : 0;
  c->frustumStale = true;
  return 1;
}

int scTickGetFrustumPlanes(ScTickContext* c, float planes[24], int* valid)
{
  if (!c || !planes) return c ? fail(c, "null argument") : 0;
  std::memcpy(planes, c->frustum.p, sizeof c->frustum.p);
  if (valid) *valid = c->frustumValid;
  return 1;
}

// frustumFromViewProj, sc_world_partition.cpp:1071-1103: planes r3 +- r0, r3 +- r1, r3 +- r2 of the
// column-major matrix' rows, normalised by 1/sqrt(a^2+b^2+c^2) when that exceeds 1e-8 (else zero plane).
int scTickSetViewProj(ScTickContext* c, const float m[16])
{
  if (!c || !m) return c ? fail(c, "null argument") : 0;
  const float row[4][4] = { { m[0], m[4], m[8], m[12] }, { m[1], m[5], m[9], m[13] },
                            { m[2], m[6], m[10], m[14] }, { m[3], m[7], m[11], m[15] } };
  for (int pl = 0; pl < 6; ++pl) {
    const int axis = pl >> 1;
    float v[4];
    for (int k = 0; k < 4; ++k) v[k] = (pl & 1) ? row[3][k] - row[axis][k] : row[3][k] + row[axis][k];
    const float lenSq = v[0] * v[0] + v[1] * v[1] + v[2] * v[2];
    float* out = c->frustum.p[pl];
    out[0] = out[1] = out[2] = out[3] = 0.0f;
    if (lenSq > 1e-8f) {
      const float invLen = 1.0f / std::sqrt(lenSq);
      out[0] = v[0] * invLen; out[1] = v[1] * invLen; out[2] = v[2] * invLen; out[3] = v[3] * invLen;
    }
  }
  c->frustumValid = 1;
  c->frustumStale = true;
  return 1;
}

int scTickSetFreezeCulling(ScTickContext* c, int freeze)
{
  if (!c) return 0;
  c->freeze = freeze ? 1 : 0;
  return 1;
}

static int exchangeBorders(ScTickContext* c, uint32_t parity, hipStream_t s, bool inCapture);

// merge what the neighbours sent, answer the ray queries, search the pairs: the half of a tile's step behind the exchange
// `done` (may be null) rides on the half's last dispatch; false = it could not (nothing launched): record it on the stream
static bool enqueuePairHalf(ScTickContext* c, const TickParams& pp, hipStream_t ps, hipEvent_t done = nullptr)
{
  const DeviceState ds = stateFor(c, pp.parity);
  launchBorderMerge(ds, pp, ps);
  if (pp.flags & SC_TICK_RAYS) launchRayQueries(ds, pp, c->rays, ps);   // sees the neighbours' border boxes too
  if (raysInPairHalf(c, pp.flags) && ds.aLane) launchAgentFrontRaysFromSnapshot(ds, pp, ps);      // ... and so do the agents' obstacle rays
  return launchPairs(ds, pp, ps, done);
}

// Can any two colliders this context may ever hold in its bins meet (Bullet's filter, sc_physics.cpp:700-712 via SURVEY 8a)?  From the
// host's mirror of the uploaded layer words -- and, on a tile, the declared vocabulary of the world (records arrive from neighbours).
// An all-static city cannot: its pair role is a sweep over the bins' counters and nothing else, and the launcher sizes the grid for
// that (pairGridFor).  A HINT for launch shapes only: the search itself never goes by it.
static bool worldCanPair(ScTickContext* c)
{
  if (c->layerSetStale || c->layerSetN != c->n) {
    std::vector<uint32_t> words;
    uint32_t last = 0; bool any = false, many = false;
    const size_t upto = std::min<size_t>(c->hLayers.size(), c->n);
    for (size_t i = 0; i < upto && !many; ++i) {
      const uint32_t w = c->hLayers[i];
      if (any && w == last) continue;
      last = w; any = true;
      if (std::find(words.begin(), words.end(), w) == words.end()) { if (words.size() >= 64u) many = true; else words.push_back(w); }
    }
    bool can = many;
    for (size_t a = 0; a < words.size() && !can; ++a)
      for (size_t b = a; b < words.size() && !can; ++b)
        can = ((words[a] & 0xFFFFu) & (words[b] >> 16)) && ((words[b] & 0xFFFFu) & (words[a] >> 16));
    c->ownLayersCanPair = can; c->layerSetStale = false; c->layerSetN = c->n;
  }
  if (c->ownLayersCanPair) return true;
  if (!c->neighbourMask) return false;
  return !c->worldLayersKnown || ((c->worldLayers & 0xFFFFu) & (c->worldLayers >> 16)) != 0u;
}

int scTickRun(ScTickContext* c, uint32_t flags)
{
  if (!c) return 0;
  const double prA = g_probe.on ? probeNow() : 0.0;
  if (!bind(c)) return 0;
  if (!flushLinks(c)) return 0;
  if ((flags & SC_TICK_BROADPHASE) && c->desc.tile_sectors_x == 0) return fail(c, "broadphase requested but the context has no tile rectangle");
  // An empty context without a broadphase has nothing to launch: the per-tick counts read as zero.  With the broadphase
  // the stages still run (every kernel copes with n == 0): an emptied tile of a multi-GPU world must rewrite its border
  // messages (header-only) and take part in the exchange, the merge and the pair search of the boxes its neighbours send.
  if (c->n == 0 && !(flags & SC_TICK_BROADPHASE)) {
    c->lastFlags = flags;
    HIP_OK(c, hipMemsetAsync(c->d.counters, 0, 8 * sizeof(uint32_t), c->stream));
    return 1;
  }
  if (flags & SC_TICK_PRODUCE_NEXT) {
    if (!(flags & SC_TICK_XFORM)) return fail(c, "SC_TICK_PRODUCE_NEXT needs SC_TICK_XFORM (the producer rides on the end-of-tick kernel)");
    if (!c->producerKind) return fail(c, "SC_TICK_PRODUCE_NEXT needs scTickSetFrameProducer first");
  }
  if (c->pairsStream && (flags & SC_TICK_BROADPHASE) && !(flags & SC_TICK_SPLIT_PAIRS)) return fail(c, "a pairs stream is set: run the broadphase with SC_TICK_SPLIT_PAIRS + scTickRunPairs");
  if (c->rb.bytes && c->graphMode) return fail(c, "graph replay and the frame read-back cannot be combined");
  if ((flags & SC_TICK_RAYS) && !(flags & SC_TICK_BROADPHASE)) return fail(c, "SC_TICK_RAYS needs SC_TICK_BROADPHASE in the same run (the queries read this tick's bins)");
  if ((flags & SC_TICK_SORT_DRAWS) && !c->sort.pipeline) return fail(c, "SC_TICK_SORT_DRAWS needs scTickSetDrawSortTable first");
  TickParams p; uint32_t grid;
  fillParams(c, flags, p, grid);
  c->lastFlags = flags;
  c->lastTickLearn = false;
  if ((flags & SC_TICK_XFORM) && !(flags & SC_TICK_BROADPHASE)) c->boxesTouched = true;      // matrices change, the bins do not follow
  if ((flags & SC_TICK_BROADPHASE) && c->homeEnabled) {
    // home slots: a learn tick when nothing is remembered, the world's shape changed (entities, hierarchy, layers) or the
    // slots have aged; every bin copy must be idle and empty for it (a rare event: the streams are joined here)
    const uint32_t xf = (c->levelOffsets.size() > 1) ? (flags & SC_TICK_XFORM) : 0u;       // (only worlds with deep levels care)
    if (!c->homeValid || c->homeEpoch != c->topoEpoch || c->homeAge >= c->homePeriod || c->homeXform != xf) {
      c->homeXform = xf;
      if (c->pairsStream && !sync(c)) return 0;
      if (c->homeCountsLive) {
        for (uint32_t q = 0; q < (c->pairsStream ? c->pipeDepth : 1u); ++q) {
          const DeviceState o = stateFor(c, q);
          HIP_OK(c, hipMemsetAsync(o.binCount, 0, (size_t)c->sectors * sizeof(uint32_t), c->stream));
          HIP_OK(c, hipMemsetAsync(o.binLayers, 0, (size_t)c->sectors * sizeof(uint32_t), c->stream));
        }
      }
      p.homeMode = kHomeLearn;
      c->homeValid = true; c->homeEpoch = c->topoEpoch; c->homeAge = 0; c->lastTickLearn = true; c->learnTicks++;
    } else { p.homeMode = kHomeUse; c->homeAge++; }
    p.homeReset = 1u;
    p.sweepOnly = worldCanPair(c) ? 0u : 1u;
    c->lastTickSweepOnly = p.sweepOnly != 0u;
    if (p.sweepOnly) p.pairRun = pairRunFor(p.binSX * p.binSZ, c->cus, c->variant, true);
    p.fastPairs = (c->fastPairs && !p.sweepOnly) ? 1u : 0u;
    // lazy records: only while nothing but this tick's own pair search reads the bins, and that search runs before the next
    // tick rewrites the world matrices (it rebuilds unwritten records from them)
    // (a pipelined pair half cannot rebuild -- the matrices are the next tick's by then -- so there only the bins that nothing in the
    //  world's declared layer vocabulary can meet stay unwritten: they are never needed)
    // (and only while the pair half is this library's to issue: with a caller-owned exchange -- scTickRun(.. | SPLIT_PAIRS), the host's
    //  transport, scTickRunPairs -- the host may upload matrices, bounds or layers, append or remove entities, or run a transform-only
    //  tick between the halves, and a rebuild would then read the world of a later moment than tick t's: every record is written)
    const bool hostBetweenHalves = (flags & SC_TICK_SPLIT_PAIRS) && !c->pairsStream && !c->ownStep;
    p.lazy = (p.homeMode == kHomeUse && c->lazyEnabled && !(flags & SC_TICK_RAYS) && !c->sensors && !hostBetweenHalves) ? (!c->pairsStream ? 1u : (c->worldLayersKnown ? 2u : 0u)) : 0u;
    p.vocab = c->worldLayers;
    p.vocabKnown = c->worldLayersKnown ? 1u : 0u;
    c->lastTickLazy = p.lazy != 0u;
    // records of entities that did not move stay as they are, unless something else changed boxes since the last tick
    p.cleanStay = (p.homeMode == kHomeUse && c->lazyEnabled && !c->pairsStream && !c->boxesTouched) ? 1u : 0u;
    c->lastTickStay = p.cleanStay != 0u;
    c->boxesTouched = false;
    c->homeCountsLive = true;
  }
  if ((flags & SC_TICK_BROADPHASE) && (flags & SC_TICK_SPLIT_PAIRS) && c->neighbourMask) {
    // every message of THIS tick parity needs its buffers: a missing one would make the pack skip that neighbour silently
    const DeviceState ds = stateFor(c, c->parity);
    for (uint32_t d = 0; d < 8; ++d)
      if (((c->neighbourMask >> d) & 1u) && (!ds.borderSend[d] || !ds.borderRecv[d]))
        return fail(c, "border buffers of this tick parity are not bound (scTickBindBorderBuffers / scTickBindBorderBuffersParity / scTickCommInit)");
  }
  if (c->frustumStale && (flags & SC_TICK_CULL)) { launchSetFrustum(c->d, c->frustum, c->stream); c->frustumStale = false; }   // outside any graph

  const uint32_t q = (flags & SC_TICK_BROADPHASE) ? c->parity : 0u;
  const bool sampledTick = c->profiling && (c->tickIndex % c->profPeriod) == 0;     // events need eager launches
  c->lastTickSampled = sampledTick;
  const double pr0 = g_probe.on ? probeNow() : 0.0;
  if (g_probe.on) g_probe.acc[5] += pr0 - prA;
  waitParityFree(c, p);
  const double pr1 = g_probe.on ? probeNow() : 0.0;
  if (c->graphMode && !sampledTick && !c->lastTickLearn) {
    const bool stale = !c->graphExec[q] || c->graphEpoch[q] != c->topoEpoch || std::memcmp(&p, &c->graphParams[q], sizeof p) != 0 || c->graphWhole[q] != c->captureWholeStep;
    if (stale) {
      dropGraph(c, (int)q);
      // whole-step capture (scTickTileStep on a tile with neighbours): the RCCL group and the pair half join the graph, so a
      // step of an in-order tile is ONE hipGraphLaunch.  RCCL operations are captured in relaxed mode (they touch the
      // communicator's own resources from inside the capture).
      const bool whole = c->captureWholeStep;
      HIP_OK(c, hipStreamBeginCapture(c->stream, whole ? hipStreamCaptureModeRelaxed : hipStreamCaptureModeThreadLocal));
      c->capturing = true;
      enqueueStages(c, p, grid, false);
      c->capturing = false;
      int okx = 1;
      if (whole) { okx = exchangeBorders(c, p.parity, c->stream, true); if (okx) enqueuePairHalf(c, p, c->stream); }
      const hipError_t ce = hipStreamEndCapture(c->stream, &c->graph[q]);
      if (!okx) { if (c->graph[q]) { hipGraphDestroy(c->graph[q]); c->graph[q] = nullptr; } return 0; }       // (the RCCL error text is already set)
      if (ce != hipSuccess) return fail(c, "hipStreamEndCapture", ce);
      if (!c->graph[q]) return fail(c, "hipStreamEndCapture returned no graph (the capture was invalidated)");
      HIP_OK(c, hipGraphInstantiate(&c->graphExec[q], c->graph[q], nullptr, nullptr, 0));
      c->graphParams[q] = p; c->graphEpoch[q] = c->topoEpoch; c->graphWhole[q] = whole;
    }
    HIP_OK(c, hipGraphLaunch(c->graphExec[q], c->stream));
  } else {
    enqueueStages(c, p, grid, true);
  }
  const double pr2 = g_probe.on ? probeNow() : 0.0;
  publishPacked(c, p);
  double prEnd = 0.0;
  if (g_probe.on) { const double pr3 = probeNow(); g_probe.acc[0] += pr1 - pr0; g_probe.acc[1] += pr2 - pr1; g_probe.acc[2] += pr3 - pr2; g_probe.n++; prEnd = pr3; }
  if (flags & SC_TICK_BROADPHASE) {
    const bool pairHalfDone = c->graphMode && !sampledTick && !c->lastTickLearn && c->captureWholeStep;
    if ((flags & SC_TICK_SPLIT_PAIRS) && !pairHalfDone) { c->pairsPending = true; c->pendingParams = p; }
    else { c->lastParity = c->parity; c->parity ^= 1u; }           // (in-order flows alternate between two copies)
  }
  c->tickIndex++;
  const hipError_t e = hipGetLastError();
  if (g_probe.on) g_probe.acc[6] += probeNow() - prEnd;
  if (e != hipSuccess) return fail(c, "kernel launch", e);
  return 1;
}

// the pair half of a pending tick; withExchange: the library's own RCCL group goes first (scTickTileStep).  On a pipelined
// tile with graph replay on, the half is a graph of its own on the pairs stream (captured once per tick parity, relaxed mode
// when the RCCL group is inside: it touches the communicator's resources during capture).
static int runPendingPairs(ScTickContext* c, bool withExchange)
{
  const uint32_t q = c->pendingParams.parity;
  hipStream_t ps = c->pairsStream ? c->pairsStream : c->stream;
  if (c->pairsStream) {
    const TickParams& pp = c->pendingParams;
    if (c->graphMode && !c->lastTickSampled && !c->lastTickLearn) {
      const bool stale = !c->pairGraphExec[q] || c->pairGraphEpoch[q] != c->topoEpoch || c->pairGraphExchange[q] != withExchange ||
                         std::memcmp(&pp, &c->pairGraphParams[q], sizeof pp) != 0;
      if (stale) {
        dropPairGraph(c, (int)q);
        HIP_OK(c, hipStreamBeginCapture(ps, withExchange ? hipStreamCaptureModeRelaxed : hipStreamCaptureModeThreadLocal));
        const int okx = withExchange ? exchangeBorders(c, q, ps, true) : 1;
        if (okx) enqueuePairHalf(c, pp, ps);
        const hipError_t ce = hipStreamEndCapture(ps, &c->pairGraph[q]);
        if (!okx) { if (c->pairGraph[q]) { hipGraphDestroy(c->pairGraph[q]); c->pairGraph[q] = nullptr; } return 0; }   // (the RCCL error text is already set)
        if (ce != hipSuccess) return fail(c, "hipStreamEndCapture (pair half)", ce);
        if (!c->pairGraph[q]) return fail(c, "hipStreamEndCapture (pair half) returned no graph (the capture was invalidated)");
        HIP_OK(c, hipGraphInstantiate(&c->pairGraphExec[q], c->pairGraph[q], nullptr, nullptr, 0));
        c->pairGraphParams[q] = pp; c->pairGraphEpoch[q] = c->topoEpoch; c->pairGraphExchange[q] = withExchange;
      }
      HIP_OK(c, hipGraphLaunch(c->pairGraphExec[q], ps));
      HIP_OK(c, hipEventRecord(c->pairsDone[q], ps));
    } else {
      // a sampled tick times its pair chain too (exchange + merge + queries + pair search, on the pairs stream): SC_TICK_K_PAIRS
      EventPair ev; const bool timed = c->profiling && c->lastTickSampled && (c->profMask & (1u << SC_TICK_K_PAIRS));
      if (timed) { ev = takeEvents(c); hipEventRecord(ev.a, ps); }
      if (withExchange && !exchangeBorders(c, q, ps, false)) return 0;
      const bool rides = enqueuePairHalf(c, pp, ps, (c->variant & 4u) ? nullptr : c->pairsDone[q]);
      if (!rides || (c->variant & 4u)) HIP_OK(c, hipEventRecord(c->pairsDone[q], ps));
      if (timed) { hipEventRecord(ev.b, ps); c->times[SC_TICK_K_PAIRS].push_back(ev); }
    }
    c->pairsInFlight[q] = true;
  } else {
    if (withExchange && !exchangeBorders(c, q, ps, false)) return 0;
    Scoped s(c, SC_TICK_K_PAIRS);
    enqueuePairHalf(c, c->pendingParams, ps);
  }
  c->pairsPending = false;
  c->lastParity = c->parity;
  c->parity = c->pairsStream ? (c->parity + 1u) % c->pipeDepth : (c->parity ^ 1u);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(c, "kernel launch", e);
  return 1;
}

int scTickRunPairs(ScTickContext* c)
{
  if (!c) return 0;
  if (!bind(c)) return 0;
  if (!c->pairsPending) return fail(c, "scTickRunPairs without a preceding scTickRun(... | SC_TICK_BROADPHASE | SC_TICK_SPLIT_PAIRS)");
  return runPendingPairs(c, false);
}

int scTickSetTile(ScTickContext* c, uint32_t rank, uint32_t neighbourMask)
{
  if (!c) return 0;
  if (rank > 127u) return fail(c, "rank must be < 128 (7 id bits)");
  // records in remembered slots carry the rank in their ids, and which slots are written on every tick depends on the ring's ownership
  if (c->rank != rank || c->neighbourMask != (neighbourMask & 0xFFu)) { c->homeValid = false; c->topoEpoch++; }
  c->rank = rank;
  c->neighbourMask = neighbourMask & 0xFFu;
  return 1;
}

int scTickSetTileGrid(ScTickContext* c, uint32_t tileX, uint32_t tileZ, uint32_t tilesX, uint32_t tilesZ)
{
  if (!c) return 0;
  if (tilesX == 0 || tilesZ == 0 || tileX >= tilesX || tileZ >= tilesZ || (uint64_t)tilesX * tilesZ > 128u) return fail(c, "tile grid: need tile < tiles and at most 128 tiles");
  c->tileX = tileX; c->tileZ = tileZ; c->tilesX = tilesX; c->tilesZ = tilesZ;
  uint32_t mask = 0;
  for (uint32_t d = 0; d < 8; ++d) {
    int dx, dz; borderDir(d, dx, dz);
    const int x = (int)tileX + dx, z = (int)tileZ + dz;
    if (x >= 0 && z >= 0 && x < (int)tilesX && z < (int)tilesZ) mask |= 1u << d;
  }
  if (c->neighbourMask != mask) { c->homeValid = false; c->topoEpoch++; }
  c->neighbourMask = mask;
  return 1;
}

uint32_t scTickBorderBytes(ScTickContext* c, uint32_t dir)
{
  if (!c || dir > 7u || !c->desc.tile_sectors_x) return 0;
  return borderWords(dir, c->desc.tile_sectors_x, c->desc.tile_sectors_z, c->borderRecs, c->halo ? 1u : 0u) * 4u;
}

int scTickSetBorderCapacity(ScTickContext* c, uint32_t recordsPerRingSector)
{
  if (!c) return 0;
  if (!c->sectors) return fail(c, "the context has no broadphase");
  if (recordsPerRingSector < 1u || recordsPerRingSector > kSectorRecMax) return fail(c, "border capacity: 1..1088 records per ring sector (a sector holds at most 64 + 1024)");
  if (c->comm) return fail(c, "scTickSetBorderCapacity must precede scTickCommInit (the library's message buffers are sized there)");
  if (c->pairsPending) return fail(c, "scTickRunPairs is pending");
  if (recordsPerRingSector == c->borderRecs) return 1;
  if (!bind(c) || !sync(c)) return 0;
  const uint32_t before = ovfRecords(c);
  c->borderRecs = recordsPerRingSector;
  // the sector overflow list has room for every border record that finds its landing bin full: grow it with the messages
  if (ovfRecords(c) > before) {
    auto regrow = [&](float4*& spill, uint32_t*& tags) -> bool {
      if (!spill) return true;
      dfree(c, spill); dfree(c, tags); spill = nullptr; tags = nullptr;
      return dalloc(c, spill, 2u * (size_t)ovfRecords(c), false) && dalloc(c, tags, ovfRecords(c));
    };
    if (!regrow(c->d.spill, c->d.spillSector)) return 0;
    for (auto& a : c->alt) if (!regrow(a.spill, a.spillSector)) return 0;
  }
  // library-owned message buffers are allocated by scTickCommInit; caller-owned ones must be re-bound at the new scTickBorderBytes
  for (uint32_t d = 0; d < 8; ++d) {
    c->d.borderSend[d] = c->d.borderRecv[d] = nullptr;
    for (auto& a : c->alt) a.borderSend[d] = a.borderRecv[d] = nullptr;
    for (uint32_t q = 0; q < kMaxParity; ++q) for (int k = 0; k < 2; ++k) if (c->ownBorder[q][d][k]) { dfree(c, c->ownBorder[q][d][k]); c->ownBorder[q][d][k] = nullptr; }
  }
  dropGraph(c); dropPairGraph(c); c->topoEpoch++;
  return 1;
}

int scTickBindBorderBuffers(ScTickContext* c, uint32_t dir, void* send, void* recv)
{
  if (!c || dir > 7u) return c ? fail(c, "bad direction") : 0;
  c->d.borderSend[dir] = static_cast<uint32_t*>(send);
  c->d.borderRecv[dir] = static_cast<uint32_t*>(recv);
  for (auto& a : c->alt) { a.borderSend[dir] = static_cast<uint32_t*>(send); a.borderRecv[dir] = static_cast<uint32_t*>(recv); }
  return 1;
}

int scTickBindBorderBuffersParity(ScTickContext* c, uint32_t parity, uint32_t dir, void* send, void* recv)
{
  if (!c) return 0;
  if (dir > 7 || parity >= kMaxParity) return fail(c, "direction must be 0..7 and parity 0..3");
  if (parity == 0) { c->d.borderSend[dir] = static_cast<uint32_t*>(send); c->d.borderRecv[dir] = static_cast<uint32_t*>(recv); }
  else { c->alt[parity - 1u].borderSend[dir] = static_cast<uint32_t*>(send); c->alt[parity - 1u].borderRecv[dir] = static_cast<uint32_t*>(recv); }
  return 1;
}

void* scTickGetBorderBuffer(ScTickContext* c, uint32_t parity, uint32_t dir, int recv)
{
  if (!c || dir > 7u || parity >= kMaxParity) return nullptr;
  uint32_t* const* set = parity == 0 ? (recv ? c->d.borderRecv : c->d.borderSend) : (recv ? c->alt[parity - 1u].borderRecv : c->alt[parity - 1u].borderSend);
  return set[dir];
}

int scTickSetPairsStream(ScTickContext* c, void* stream)
{
  if (!c) return 0;
  if (!bind(c)) return 0;
  if (!sync(c)) return 0;
  if (c->pairsPending) return fail(c, "scTickRunPairs is pending");
  // The two flows clear the per-parity broadphase state differently (in order: a pair kernel clears the OTHER parity for the
  // next tick; pipelined: a small kernel behind the pair kernel clears its OWN parity), so a switch in either direction
  // would leave one parity's counters, shard counters and big-box bits holding the last tick's values.  Everything is idle
  // here (synchronised above): start both parities from a clean slate.
  auto resetBroadphaseState = [&]() -> bool {
    if (!c->sectors) return true;
    hipError_t e = hipMemsetAsync(c->d.counters + kCtrPar, 0, (kCounterWords - kCtrPar) * sizeof(uint32_t), c->stream);
    if (e == hipSuccess) e = hipMemsetAsync(c->d.pairShardCount, 0, (kMaxParity + 1u) * kPairShards * kShardStride * sizeof(uint32_t), c->stream);
    if (e == hipSuccess) e = hipMemsetAsync(c->d.binCount, 0, (size_t)c->sectors * sizeof(uint32_t), c->stream);
    if (e == hipSuccess) e = hipMemsetAsync(c->d.binLayers, 0, (size_t)c->sectors * sizeof(uint32_t), c->stream);
    if (e == hipSuccess) e = hipMemsetAsync(c->d.ovfLo, 0xFF, (size_t)c->sectors * sizeof(uint32_t), c->stream);
    if (e == hipSuccess) e = hipMemsetAsync(c->d.ovfHi, 0, (size_t)c->sectors * sizeof(uint32_t), c->stream);
    for (auto& a : c->alt) {
      if (e == hipSuccess && a.binCount) e = hipMemsetAsync(a.binCount, 0, (size_t)c->sectors * sizeof(uint32_t), c->stream);
      if (e == hipSuccess && a.binLayers) e = hipMemsetAsync(a.binLayers, 0, (size_t)c->sectors * sizeof(uint32_t), c->stream);
      if (e == hipSuccess && a.ovfLo) e = hipMemsetAsync(a.ovfLo, 0xFF, (size_t)c->sectors * sizeof(uint32_t), c->stream);
      if (e == hipSuccess && a.ovfHi) e = hipMemsetAsync(a.ovfHi, 0, (size_t)c->sectors * sizeof(uint32_t), c->stream);
    }
    if (e != hipSuccess) return fail(c, "hipMemsetAsync (broadphase state)", e);
    c->parity = 0; c->lastParity = 0;
    c->homeValid = false; c->homeCountsLive = false;      // every bin copy is empty again: the next tick learns its slots afresh
    return sync(c);
  };
  if (!stream) {
    if (c->pairsStream && !resetBroadphaseState()) return 0;
    c->pairsStream = nullptr; for (bool& f : c->pairsInFlight) f = false; c->topoEpoch++;
    return 1;
  }
  if (!c->sectors) return fail(c, "the context has no broadphase");
  for (uint32_t q = 1; q < c->pipeDepth; ++q) {
    ScTickContext::AltSet& a = c->alt[q - 1u];
    const size_t N = c->cap;
    if (!a.bins && (!dalloc(c, a.binCount, c->sectors) || !dalloc(c, a.binLayers, c->sectors) ||
        !dalloc(c, a.bins, (size_t)c->sectors * kBinCap * 2u, false) || !dalloc(c, a.bigList, (N + 8u * kBorderBigCap) * 2u, false) ||
        !dalloc(c, a.spill, 2u * (size_t)ovfRecords(c), false) || !dalloc(c, a.spillSector, ovfRecords(c)) || !dalloc(c, a.ovfLo, c->sectors, false) || !dalloc(c, a.ovfHi, c->sectors))) return 0;
    // (border buffers: scTickBindBorderBuffers writes every copy's slots itself, scTickBindBorderBuffersParity and
    //  scTickCommInit bind per parity -- nothing is inherited here.  Round 2 copied parity 1's pointers over a parity whose
    //  direction 0 was unbound, which is every tile without a (-1,-1) neighbour: parities 1..3 then shared one message set.)
  }
  for (uint32_t k = 0; k < kMaxParity; ++k) {
    if (!c->packed[k]) HIP_OK(c, hipEventCreateWithFlags(&c->packed[k], hipEventDisableTiming | hipEventReleaseToDevice));
    if (!c->pairsDone[k]) HIP_OK(c, hipEventCreateWithFlags(&c->pairsDone[k], hipEventDisableTiming | hipEventReleaseToDevice));
  }
  if (!c->pairsStream && !resetBroadphaseState()) return 0;
  c->pairsStream = static_cast<hipStream_t>(stream);
  for (bool& f : c->pairsInFlight) f = false;
  c->topoEpoch++;
  return 1;
}

int scTickSetStream(ScTickContext* c, void* stream, int external)
{
  if (!c) return 0;
  if (!bind(c) || !sync(c)) return 0;
  dropGraph(c);
  c->stream = external ? static_cast<hipStream_t>(stream) : c->ownStream;
  return 1;
}

int scTickSynchronize(ScTickContext* c)
{
  if (!c) return 0;
  if (!bind(c)) return 0;
  return sync(c) ? 1 : 0;
}

int scTickNudgeRootsX(ScTickContext* c, float dx)
{
  if (!c) return 0;
  if (!bind(c) || !flushLinks(c)) return 0;
  Scoped s(c, SC_TICK_K_NUDGE);
  launchNudgeRootsX(c->d, c->n, dx, c->stream);
  return 1;
}

int scTickUploadMovers(ScTickContext* c, uint32_t first, uint32_t count, const uint8_t* kind, const float* vel, const float* lo, const float* hi)
{
  if (!c || !kind || !vel || !lo || !hi) return c ? fail(c, "null argument") : 0;
  if (!bind(c) || !rangeOk(c, first, count)) return 0;
  DeviceState& d = c->d;
  if (!d.moverKind) {
    const size_t N = c->cap;
    if (!dalloc(c, d.moverKind, N) || !dalloc(c, d.mvx, N) || !dalloc(c, d.mvz, N) || !dalloc(c, d.mlox, N) ||
        !dalloc(c, d.mloz, N) || !dalloc(c, d.mhix, N) || !dalloc(c, d.mhiz, N)) return 0;
  }
  if (!count) return 1;
  std::vector<uint32_t> k32(count);
  std::vector<float> a[6];
  for (auto& v : a) v.resize(count);
  for (uint32_t i = 0; i < count; ++i) {
    if (kind[i] > 2) return fail(c, "mover kind must be 0 (none), 1 (vehicle: wrap) or 2 (ped: reflect)");
    k32[i] = kind[i];
    a[0][i] = vel[2 * i]; a[1][i] = vel[2 * i + 1]; a[2][i] = lo[2 * i]; a[3][i] = lo[2 * i + 1]; a[4][i] = hi[2 * i]; a[5][i] = hi[2 * i + 1];
  }
  float* dst[6] = { d.mvx, d.mvz, d.mlox, d.mloz, d.mhix, d.mhiz };
  if (!h2d(c, d.moverKind + first, k32.data(), (size_t)count * 4u)) return 0;
  for (int q = 0; q < 6; ++q) if (!h2d(c, dst[q] + first, a[q].data(), (size_t)count * 4u)) return 0;
  return sync(c) ? 1 : 0;
}

int scTickAdvanceMovers(ScTickContext* c, float dt)
{
  if (!c) return 0;
  if (!bind(c) || !flushLinks(c)) return 0;
  if (!c->d.moverKind) return fail(c, "no movers uploaded");
  Scoped s(c, SC_TICK_K_NUDGE);
  launchAdvanceMovers(c->d, c->n, dt, 1.0f - std::exp(-2.5f * dt), c->trafficMult, c->stream);
  return 1;
}

int scTickSetFrameProducer(ScTickContext* c, uint32_t kind, float param)
{
  if (!c) return 0;
  if (kind > 2u) return fail(c, "producer kind must be 0 (none), 1 (nudge roots) or 2 (advance movers)");
  if (kind == 2u && !c->d.moverKind) return fail(c, "no movers uploaded");
  if (!bind(c) || !sync(c)) return 0;
  dropGraph(c);                                   // the captured frame changes
  c->producerKind = kind;
  c->producerParam = param;
  return 1;
}

int scTickReadMoverVelocities(ScTickContext* c, uint32_t first, uint32_t count, float* vel)
{
  if (!c || !vel) return c ? fail(c, "null argument") : 0;
  if (!bind(c) || !rangeOk(c, first, count)) return 0;
  if (!c->d.moverKind) return fail(c, "no movers uploaded");
  if (!count) return 1;
  std::vector<float> x(count), z(count);
  if (!d2h(c, x.data(), c->d.mvx + first, (size_t)count * 4u) || !d2h(c, z.data(), c->d.mvz + first, (size_t)count * 4u) || !sync(c)) return 0;
  for (uint32_t i = 0; i < count; ++i) { vel[2 * i] = x[i]; vel[2 * i + 1] = z[i]; }
  return 1;
}

int scTickGetCounts(ScTickContext* c, ScTickCounts* out)
{
  if (!c || !out) return c ? fail(c, "null argument") : 0;
  if (!bind(c) || !joinPairs(c)) return 0;
  uint32_t k[kCounterWords] = {};
  if (!d2h(c, k, c->d.counters, sizeof k) || !sync(c)) return 0;
  std::memset(out, 0, sizeof *out);
  const uint32_t resultSlot = c->lastParity;      // (a pipelined tile's parity keeps its results until the tick before its next use clears it)
  const uint32_t* bp = k + kCtrPar + 8u * resultSlot;
  out->entities = c->n;
  out->renderables_total = k[6];
  out->visible = k[0];
  out->culled = k[1];
  if (c->sectors && (c->lastFlags & SC_TICK_BROADPHASE) && !c->pairsPending) {
    TickParams pp{}; pp.maxPairs = c->maxPairs;
    launchGatherPairs(c->d, pp, resultSlot, c->dPairsOut, c->dPairTotal, c->stream);
    uint32_t tot[2] = {};
    if (!d2h(c, tot, c->dPairTotal, sizeof tot) || !sync(c)) return 0;
    out->pairs = tot[0];
    out->pairs_truncated = tot[1];
  }
  out->bin_overflow = bp[kCtrSpill];                 // length of the sector overflow list (own boxes + border records that landed in a full bin)
  out->big_boxes = (c->lastFlags & SC_TICK_SPLIT_PAIRS) ? bp[kCtrBigLocal] : bp[kCtrBig];
  out->border_lost = bp[kCtrBorderLost];
  if (c->sectors) {
    uint32_t lc[1u + 2u * kMaxParity] = {};
    if (!d2h(c, lc, c->d.lazyCtl, sizeof lc) || !sync(c)) return 0;
    out->vocabulary_violations = lc[1u + kMaxParity + resultSlot];
  }
  out->draws_emitted = k[4];
  out->draws_dropped = k[5];
  out->draws_sorted = (c->lastFlags & SC_TICK_SORT_DRAWS) ? k[kCtrDrawsSorted] : 0u;
  out->max_depth = c->maxDepth;
  out->unreachable = c->unreachable;
  out->relinks = c->relinks;
  return 1;
}

static int readIndexList(ScTickContext* c, const uint32_t* dev, uint32_t counterSlot, uint32_t* out, uint32_t cap, uint32_t* count)
{
  if (!c || !count) return c ? fail(c, "null argument") : 0;
  if (!bind(c)) return 0;
  uint32_t k[16] = {};
  if (!d2h(c, k, c->d.counters, sizeof k) || !sync(c)) return 0;
  const uint32_t total = k[counterSlot];
  *count = total;
  const uint32_t take = std::min(total, cap);
  if (take && out) { if (!d2h(c, out, dev, (size_t)take * 4u) || !sync(c)) return 0; }
  return 1;
}

int scTickReadVisible(ScTickContext* c, uint32_t* out, uint32_t cap, uint32_t* count)
{
  return c ? readIndexList(c, c->d.visibleIdx, 0, out, cap, count) : 0;
}

int scTickReadCulled(ScTickContext* c, uint32_t* out, uint32_t cap, uint32_t* count)
{
  if (!c) return 0;
  if (!(c->lastFlags & SC_TICK_CULLED_LIST)) return fail(c, "the last scTickRun did not request SC_TICK_CULLED_LIST");
  return readIndexList(c, c->d.culledIdx, 1, out, cap, count);
}

int scTickReadVisibilityBits(ScTickContext* c, uint64_t* words, uint32_t wordCap)
{
  if (!c || !words) return c ? fail(c, "null argument") : 0;
  if (!bind(c)) return 0;
  const uint32_t nWords = (c->n + 63u) / 64u;
  if (wordCap < nWords) return fail(c, "word capacity too small");
  if (nWords && (!d2h(c, words, c->d.vis, (size_t)nWords * 8u) || !sync(c))) return 0;
  if (c->n & 63u) words[nWords - 1] &= (1ull << (c->n & 63u)) - 1ull;
  return 1;
}

int scTickReadWorldMatrices(ScTickContext* c, uint32_t first, uint32_t count, float* m16)
{
  if (!c || !m16) return c ? fail(c, "null argument") : 0;
  if (!bind(c) || !rangeOk(c, first, count)) return 0;
  if (!count) return 1;
  std::vector<float4> r0(count), r1(count), r2(count);
  if (!d2h(c, r0.data(), c->d.w0 + first, (size_t)count * 16u) || !d2h(c, r1.data(), c->d.w1 + first, (size_t)count * 16u) ||
      !d2h(c, r2.data(), c->d.w2 + first, (size_t)count * 16u) || !sync(c)) return 0;
  for (uint32_t i = 0; i < count; ++i) {
    const float rows[12] = { r0[i].x, r0[i].y, r0[i].z, r0[i].w, r1[i].x, r1[i].y, r1[i].z, r1[i].w, r2[i].x, r2[i].y, r2[i].z, r2[i].w };
    rowsToMat4(rows, m16 + 16 * (size_t)i);
  }
  return 1;
}

int scTickReadWorldMatricesIndexed(ScTickContext* c, const uint32_t* idx, uint32_t count, float* m16)
{
  if (!c || !m16 || (!idx && count)) return c ? fail(c, "null argument") : 0;
  if (!bind(c)) return 0;
  if (!count) return 1;
  for (uint32_t i = 0; i < count; ++i) if (idx[i] >= c->n) return fail(c, "dense index out of range");
  if (!needScratch(c, count)) return 0;
  if (!h2d(c, c->dIdx, idx, (size_t)count * 4u)) return 0;
  launchGatherRows(c->d, c->dIdx, count, c->dRows, c->stream);
  std::vector<float> rows((size_t)count * 12);
  if (!d2h(c, rows.data(), c->dRows, rows.size() * 4u) || !sync(c)) return 0;
  for (uint32_t i = 0; i < count; ++i) rowsToMat4(rows.data() + 12 * (size_t)i, m16 + 16 * (size_t)i);
  return 1;
}

int scTickReadDirty(ScTickContext* c, uint32_t first, uint32_t count, uint8_t* out)
{
  if (!c || !out) return c ? fail(c, "null argument") : 0;
  if (!bind(c) || !rangeOk(c, first, count)) return 0;
  if (!count) return 1;
  const uint32_t w0 = first >> 5, w1 = (first + count + 31u) >> 5;
  std::vector<uint32_t> words(w1 - w0);
  if (!d2h(c, words.data(), c->d.dirty + w0, words.size() * 4u) || !sync(c)) return 0;
  for (uint32_t i = 0; i < count; ++i) { const uint32_t g = first + i; out[i] = (words[(g >> 5) - w0] >> (g & 31u)) & 1u; }
  return 1;
}

int scTickReadPositions(ScTickContext* c, uint32_t first, uint32_t count, float* pos3)
{
  if (!c || !pos3) return c ? fail(c, "null argument") : 0;
  if (!bind(c) || !rangeOk(c, first, count)) return 0;
  if (!count) return 1;
  std::vector<float> x(count), y(count), z(count);
  if (!d2h(c, x.data(), c->d.px + first, (size_t)count * 4u) || !d2h(c, y.data(), c->d.py + first, (size_t)count * 4u) ||
      !d2h(c, z.data(), c->d.pz + first, (size_t)count * 4u) || !sync(c)) return 0;
  for (uint32_t i = 0; i < count; ++i) { pos3[3 * i] = x[i]; pos3[3 * i + 1] = y[i]; pos3[3 * i + 2] = z[i]; }
  return 1;
}

int scTickReadWorldAabbs(ScTickContext* c, uint32_t first, uint32_t count, float* min3, float* max3)
{
  if (!c || !min3 || !max3) return c ? fail(c, "null argument") : 0;
  if (!bind(c) || !rangeOk(c, first, count)) return 0;
  if ((c->lastFlags & (SC_TICK_BROADPHASE | SC_TICK_DENSE_AABBS)) != (SC_TICK_BROADPHASE | SC_TICK_DENSE_AABBS))
    return fail(c, "the last scTickRun did not request SC_TICK_BROADPHASE | SC_TICK_DENSE_AABBS");
  if (!count) return 1;
  std::vector<float4> a(count), b(count);
  if (!d2h(c, a.data(), c->d.aabbMin + first, (size_t)count * 16u) || !d2h(c, b.data(), c->d.aabbMax + first, (size_t)count * 16u) || !sync(c)) return 0;
  for (uint32_t i = 0; i < count; ++i) {
    min3[3 * i] = a[i].x; min3[3 * i + 1] = a[i].y; min3[3 * i + 2] = a[i].z;
    max3[3 * i] = b[i].x; max3[3 * i + 1] = b[i].y; max3[3 * i + 2] = b[i].z;
  }
  return 1;
}

int scTickReadPairs(ScTickContext* c, uint32_t* pairs2, uint32_t cap, uint32_t* count)
{
  if (!c || !count) return c ? fail(c, "null argument") : 0;
  if (!bind(c)) return 0;
  if (!(c->lastFlags & SC_TICK_BROADPHASE)) return fail(c, "the last scTickRun did not request SC_TICK_BROADPHASE");
  if (c->pairsPending) return fail(c, "scTickRunPairs has not been called for the last tick");
  if (!joinPairs(c)) return 0;
  // the pair list is kept in per-shard segments on the device; gather them into one list first
  TickParams pp{}; pp.maxPairs = c->maxPairs;
  const uint32_t slot = c->lastParity;
  launchGatherPairs(c->d, pp, slot, c->dPairsOut, c->dPairTotal, c->stream);
  uint32_t tot[2] = {};
  if (!d2h(c, tot, c->dPairTotal, sizeof tot) || !sync(c)) return 0;
  *count = tot[0];
  // each shard keeps at most maxPairs / 64 pairs; recount what the gather could place
  std::vector<uint32_t> sc((kMaxParity + 1u) * kPairShards * kShardStride);
  if (!d2h(c, sc.data(), c->d.pairShardCount, sc.size() * sizeof(uint32_t)) || !sync(c)) return 0;
  uint32_t placed = 0;
  const uint32_t shardCap = c->maxPairs / kPairShards;
  for (uint32_t s = 0; s < kPairShards; ++s) placed += std::min(sc[(slot * kPairShards + s) * kShardStride], shardCap);
  const uint32_t take = std::min(placed, cap);
  if (take && pairs2) { if (!d2h(c, pairs2, c->dPairsOut, (size_t)take * 8u) || !sync(c)) return 0; }
  return 1;
}

int scTickReadDraws(ScTickContext* c, ScTickDrawItem* items, uint32_t cap, uint32_t* count)
{
  if (!c || !count) return c ? fail(c, "null argument") : 0;
  if (!bind(c)) return 0;
  if (!(c->lastFlags & SC_TICK_DRAWS)) return fail(c, "the last scTickRun did not request SC_TICK_DRAWS");
  uint32_t k[16] = {};
  if (!d2h(c, k, c->d.counters, sizeof k) || !sync(c)) return 0;
  const uint32_t have = (c->lastFlags & SC_TICK_SORT_DRAWS) ? k[kCtrDrawsSorted] : k[4];
  *count = have;
  const uint32_t take = std::min(have, cap);
  if (take && items) { if (!d2h(c, items, c->lastDraws ? c->lastDraws : c->dDraws, (size_t)take * sizeof(ScTickDrawItem)) || !sync(c)) return 0; }
  return 1;
}

int scTickSetRayQueries(ScTickContext* c, uint32_t count, const float* origin3, const float* dir3, const float* maxDist, const uint32_t* mask)
{
  if (!c) return 0;
  if (count && (!origin3 || !dir3 || !maxDist || !mask)) return fail(c, "null argument");
  if (!bind(c) || !sync(c)) return 0;
  if (count > c->rayCap) {
    dfree(c, const_cast<float4*>(c->rays.origin)); dfree(c, const_cast<float4*>(c->rays.dir)); dfree(c, c->rays.hits);
    c->rays = RayQueryState{}; c->rayCap = 0;
    const uint32_t want = std::max(count, 1024u);
    float4 *o = nullptr, *dd = nullptr; RayHit48* h = nullptr;
    if (!dalloc(c, o, want, false) || !dalloc(c, dd, want, false) || !dalloc(c, h, want)) return 0;
    c->rays.origin = o; c->rays.dir = dd; c->rays.hits = h; c->rayCap = want;
  }
  c->rays.count = count;
  c->topoEpoch++;                      // a captured frame holds the old batch size
  if (!count) return 1;
  std::vector<float4> o(count), dd(count);
  for (uint32_t i = 0; i < count; ++i) {
    o[i] = make_float4(origin3[3 * i], origin3[3 * i + 1], origin3[3 * i + 2], maxDist[i]);
    float w; std::memcpy(&w, &mask[i], 4);
    dd[i] = make_float4(dir3[3 * i], dir3[3 * i + 1], dir3[3 * i + 2], w);
  }
  if (!h2d(c, const_cast<float4*>(c->rays.origin), o.data(), (size_t)count * 16u) ||
      !h2d(c, const_cast<float4*>(c->rays.dir), dd.data(), (size_t)count * 16u)) return 0;
  return sync(c) ? 1 : 0;
}

int scTickReadRayHits(ScTickContext* c, ScTickRayHit* hits, uint32_t cap, uint32_t* count)
{
  if (!c || !count) return c ? fail(c, "null argument") : 0;
  if (!bind(c)) return 0;
  if (!(c->lastFlags & SC_TICK_RAYS)) return fail(c, "the last scTickRun did not request SC_TICK_RAYS");
  if (c->pairsPending) return fail(c, "ray hits are ready after scTickRunPairs");
  if (!joinPairs(c)) return 0;
  static_assert(sizeof(ScTickRayHit) == sizeof(RayHit48), "ray hit layouts differ");
  *count = c->rays.count;
  const uint32_t take = std::min(c->rays.count, cap);
  if (take && hits) { if (!d2h(c, hits, c->rays.hits, (size_t)take * sizeof(ScTickRayHit)) || !sync(c)) return 0; }
  return 1;
}

int scTickQueryOccupied(ScTickContext* c, uint32_t count, const float* pos3, const float* radius, const uint32_t* mask, uint8_t* blocked)
{
  if (!c) return 0;
  if (count && (!pos3 || !radius || !mask || !blocked)) return fail(c, "null argument");
  if (count > kMaxOccupancyQueries) return fail(c, "at most 256 occupancy queries per call");
  if (!bind(c)) return 0;
  if (!count) return 1;
  std::memset(blocked, 0, count);
  if (!c->n) return 1;
  // queries (16 B each) and the answer bits share the scratch row buffer
  if (!needScratch(c, kMaxOccupancyQueries)) return 0;
  std::vector<float4> q(count);
  for (uint32_t k = 0; k < count; ++k) { float w; std::memcpy(&w, &mask[k], 4); q[k] = make_float4(pos3[3 * k], pos3[3 * k + 2], radius[k], w); }
  float4* dq = reinterpret_cast<float4*>(c->dRows);
  uint32_t* bits = c->dIdx;
  uint32_t out[kMaxOccupancyQueries / 32] = {};
  HIP_OK(c, hipMemsetAsync(bits, 0, sizeof out, c->stream));
  if (!h2d(c, dq, q.data(), (size_t)count * 16u)) return 0;
  launchOccupancy(c->d, c->n, dq, count, bits, c->stream);
  if (!d2h(c, out, bits, sizeof out) || !sync(c)) return 0;
  for (uint32_t k = 0; k < count; ++k) blocked[k] = (uint8_t)((out[k >> 5] >> (k & 31u)) & 1u);
  return 1;
}

int scTickSetProfilingKernels(ScTickContext* c, uint32_t mask)
{
  if (!c) return 0;
  c->profMask = mask ? mask : 0xFFFFFFFFu;
  return 1;
}

int scTickSetProfiling(ScTickContext* c, int enable)
{
  if (!c) return 0;
  if (!bind(c)) return 0;
  if (enable) {
    sync(c);
    for (auto& v : c->times) { for (auto& p : v) c->eventPool.push_back(p); v.clear(); }
  }
  c->profiling = enable != 0;
  c->profPeriod = enable > 1 ? (uint32_t)enable : 1u;
  c->tickIndex = 0;
  return 1;
}

int scTickGetKernelTimes(ScTickContext* c, uint32_t kernel, float* ms, uint32_t cap, uint32_t* count)
{
  if (!c || !count || kernel >= SC_TICK_K_COUNT) return c ? fail(c, "bad argument") : 0;
  if (!bind(c) || !sync(c)) return 0;
  auto& v = c->times[kernel];
  *count = (uint32_t)v.size();
  for (uint32_t i = 0; i < v.size() && i < cap && ms; ++i) {
    float t = 0.0f;
    HIP_OK(c, hipEventElapsedTime(&t, v[i].a, v[i].b));
    ms[i] = t;
  }
  return 1;
}

int scTickSetGraphMode(ScTickContext* c, int enable)
{
  if (!c) return 0;
  if (!bind(c)) return 0;
  c->graphMode = enable != 0;
  if (!enable) { sync(c); dropGraph(c); dropPairGraph(c); }
  return 1;
}

void* scTickGetStream(ScTickContext* c) { return c ? (void*)c->stream : nullptr; }

// ---- per-frame read-back ---------------------------------------------------------------------------------------------
int scTickSetFrameReadback(ScTickContext* c, uint32_t maxVisible, uint32_t maxDraws)
{
  if (!c) return 0;
  if (!bind(c) || !sync(c)) return 0;
  ScTickContext::FrameReadback& rb = c->rb;
  if (rb.copyStream) HIP_OK(c, hipStreamSynchronize(rb.copyStream));
  for (int k = 0; k < 2; ++k) { dfree(c, rb.dBlock[k]); rb.dBlock[k] = nullptr; if (rb.hBlock[k]) { hipHostFree(rb.hBlock[k]); rb.hBlock[k] = nullptr; } rb.inFlight[k] = false; }
  rb.bytes = 0; rb.frames = 0; rb.maxVisible = rb.maxDraws = 0;
  if (!maxVisible && !maxDraws) return 1;
  if (maxVisible > c->cap || maxDraws > c->cap) return fail(c, "read-back sizes exceed the context's capacity");
  maxVisible = (maxVisible + 3u) & ~3u;                        // keeps the draw items 16-byte aligned inside the block
  const size_t bytes = (size_t)kFrameHeaderWords * 4u + (size_t)maxVisible * 4u + (size_t)maxDraws * sizeof(ScTickDrawItem);
  if (!rb.copyStream) {
    HIP_OK(c, hipStreamCreateWithFlags(&rb.copyStream, hipStreamNonBlocking));
    for (int k = 0; k < 2; ++k) {
      HIP_OK(c, hipEventCreateWithFlags(&rb.staged[k], hipEventDisableTiming));
      HIP_OK(c, hipEventCreateWithFlags(&rb.copied[k], hipEventDisableTiming));
    }
  }
  for (int k = 0; k < 2; ++k) {
    if (!dalloc(c, rb.dBlock[k], bytes / 4u)) return 0;
    void* h = nullptr;
    HIP_OK(c, hipHostMalloc(&h, bytes, hipHostMallocDefault));
    std::memset(h, 0, bytes);
    rb.hBlock[k] = static_cast<uint32_t*>(h);
  }
  rb.maxVisible = maxVisible; rb.maxDraws = maxDraws; rb.bytes = bytes;
  dropGraph(c);
  return 1;
}

int scTickAcquireFrame(ScTickContext* c, uint32_t framesBack, ScTickFrame* out)
{
  if (!c || !out) return c ? fail(c, "null argument") : 0;
  ScTickContext::FrameReadback& rb = c->rb;
  if (!rb.bytes) return fail(c, "scTickSetFrameReadback first");
  if (framesBack > 1u) return fail(c, "frames_back must be 0 (the latest frame) or 1 (the one before): two frames are kept");
  if (rb.frames <= framesBack) return fail(c, "that frame has not been produced yet");
  if (!bind(c)) return 0;
  const uint32_t f = (uint32_t)((rb.frames - 1u - framesBack) & 1u);
  HIP_OK(c, hipEventSynchronize(rb.copied[f]));               // this frame's copy only: later work keeps running
  const uint32_t* h = rb.hBlock[f];
  std::memset(out, 0, sizeof *out);
  out->tick = (uint64_t)h[6] | ((uint64_t)h[7] << 32);
  out->visible = h[0]; out->culled = h[1]; out->renderables_total = h[2];
  out->draws_emitted = h[3]; out->draws_dropped = h[4]; out->draws_sorted = h[5];
  out->visible_in_buffer = h[8]; out->draws_in_buffer = h[9];
  out->visible_indices = h + kFrameHeaderWords;
  out->draws = reinterpret_cast<const ScTickDrawItem*>(h + kFrameHeaderWords + rb.maxVisible);
  return 1;
}

// ---- on-rails traffic: lane graph, agents, tier selection ----------------------------------------------
int scTickSetLaneGraph(ScTickContext* c, const ScTickLaneGraph* g)
{
  if (!c || !g) return c ? fail(c, "null argument") : 0;
  if (g->segments && (!g->seg_start3 || !g->seg_dir3 || !g->seg_length || !g->seg_end_node || !g->seg_speed_limit)) return fail(c, "null segment array");
  if (g->nodes && (!g->node_pos3 || !g->node_conn_offset)) return fail(c, "null node array");
  if (g->connections && !g->node_conn) return fail(c, "null connection array");
  for (uint32_t i = 0; i < g->segments; ++i) if (g->seg_end_node[i] >= g->nodes) return fail(c, "segment end node out of range");
  for (uint32_t i = 0; i < g->nodes; ++i) if (g->node_conn_offset[i] > g->node_conn_offset[i + 1]) return fail(c, "node connection offsets must ascend");
  if (g->nodes && g->node_conn_offset[g->nodes] != g->connections) return fail(c, "node_conn_offset[nodes] != connections");
  if (!bind(c) || !sync(c)) return 0;
  for (void*& p : c->laneAllocs) { dfree(c, p); p = nullptr; }
  std::vector<float4> A(g->segments), B(g->segments), P(g->nodes);
  std::vector<uint4> C(g->segments);
  for (uint32_t i = 0; i < g->segments; ++i) {
    const float* st = g->seg_start3 + 3 * (size_t)i; const float* dr = g->seg_dir3 + 3 * (size_t)i;
    A[i] = make_float4(st[0], st[1], st[2], g->seg_length[i]);
    B[i] = make_float4(dr[0], dr[1], dr[2], g->seg_speed_limit[i]);
    // yawFromDir (sc_traffic_ai.cpp:72-75) and then the sin / cos mat4_rotation_xyz takes of it (sc_math.cpp:102-107), host libm
    const float yaw = std::atan2(dr[0], dr[2]);
    const float sy = std::sin(yaw), cy = std::cos(yaw);
    uint32_t sb, cb; std::memcpy(&sb, &sy, 4); std::memcpy(&cb, &cy, 4);
    C[i] = make_uint4(g->seg_end_node[i], (!g->seg_active || g->seg_active[i]) ? 1u : 0u, sb, cb);
  }
  for (uint32_t i = 0; i < g->nodes; ++i) P[i] = make_float4(g->node_pos3[3 * (size_t)i], g->node_pos3[3 * (size_t)i + 1], g->node_pos3[3 * (size_t)i + 2], 0.0f);
  float4 *dA = nullptr, *dB = nullptr, *dP = nullptr; uint4* dC = nullptr; uint32_t *dOff = nullptr, *dConn = nullptr;
  if (!dalloc(c, dA, g->segments, false) || !dalloc(c, dB, g->segments, false) || !dalloc(c, dC, g->segments, false) ||
      !dalloc(c, dP, g->nodes, false) || !dalloc(c, dOff, (size_t)g->nodes + 1u) || !dalloc(c, dConn, g->connections, false)) return 0;
  c->laneAllocs[0] = dA; c->laneAllocs[1] = dB; c->laneAllocs[2] = dC; c->laneAllocs[3] = dP; c->laneAllocs[4] = dOff; c->laneAllocs[5] = dConn;
  bool ok = true;
  if (g->segments) ok = h2d(c, dA, A.data(), A.size() * 16u) && h2d(c, dB, B.data(), B.size() * 16u) && h2d(c, dC, C.data(), C.size() * 16u);
  if (ok && g->nodes) ok = h2d(c, dP, P.data(), P.size() * 16u) && h2d(c, dOff, g->node_conn_offset, ((size_t)g->nodes + 1u) * 4u);
  if (ok && g->connections) ok = h2d(c, dConn, g->node_conn, (size_t)g->connections * 4u);
  if (!ok || !sync(c)) return 0;
  LaneGraphDev& L = c->d.lanes;
  L.segA = dA; L.segB = dB; L.segC = dC; L.nodePos = dP; L.nodeConnOff = dOff; L.nodeConn = dConn; L.segments = g->segments; L.nodes = g->nodes;
  c->topoEpoch++;                       // captured graphs hold the old table pointers
  return 1;
}

int scTickSetLaneActive(ScTickContext* c, const uint32_t* segIds, uint32_t count, int active)
{
  if (!c || (!segIds && count)) return c ? fail(c, "null argument") : 0;
  if (!bind(c) || !sync(c)) return 0;
  for (uint32_t k = 0; k < count; ++k) {
    if (segIds[k] >= c->d.lanes.segments) return fail(c, "segment id out of range");
    const uint32_t v = active ? 1u : 0u;
    if (!h2d(c, reinterpret_cast<uint32_t*>(const_cast<uint4*>(c->d.lanes.segC) + segIds[k]) + 1, &v, 4u)) return 0;
    if (!sync(c)) return 0;           // `v` dies with the iteration
  }
  return 1;
}

int scTickSetTrafficSpeedMultiplier(ScTickContext* c, float multiplier)
{
  if (!c) return 0;
  if (!bind(c) || !sync(c)) return 0;
  dropGraph(c);
  c->trafficMult = multiplier;
  return 1;
}

int scTickUploadTrafficAgents(ScTickContext* c, uint32_t first, uint32_t count, const uint8_t* isAgent, const uint32_t* laneId,
                              const float* laneS, const float* targetSpeed, const uint8_t* mode, const float* lookAhead)
{
  if (!c || !isAgent || !laneId || !laneS || !targetSpeed || !mode) return c ? fail(c, "null argument") : 0;
  if (!bind(c) || !rangeOk(c, first, count)) return 0;
  DeviceState& d = c->d;
  const size_t N = c->cap;
  if (!d.moverKind && (!dalloc(c, d.moverKind, N) || !dalloc(c, d.mvx, N) || !dalloc(c, d.mvz, N) || !dalloc(c, d.mlox, N) ||
                       !dalloc(c, d.mloz, N) || !dalloc(c, d.mhix, N) || !dalloc(c, d.mhiz, N))) return 0;
  if (!d.aLane && (!dalloc(c, d.aLane, N) || !dalloc(c, d.aS, N) || !dalloc(c, d.aSpeed, N) || !dalloc(c, d.aMode, N) || !dalloc(c, d.aLook, N) ||
                   !dalloc(c, d.aDesired, N) || !dalloc(c, d.tierCounts, 4) || !dalloc(c, d.tierNear, kTierNearCap, false) || !dalloc(c, c->dTierPatch, kTierNearCap, false))) return 0;
  if (!count) return 1;
  if (!sync(c)) return 0;
  std::vector<uint32_t> kind(count), md(count);
  std::vector<float> look(count);
  if (!d2h(c, kind.data(), d.moverKind + first, (size_t)count * 4u) || !sync(c)) return 0;
  for (uint32_t i = 0; i < count; ++i) {
    if (mode[i] > 2u) return fail(c, "mode must be 0 (Physics), 1 (Kinematic) or 2 (OnRails)");
    if (isAgent[i]) kind[i] = kMoverTraffic; else if (kind[i] == kMoverTraffic) kind[i] = 0u;
    md[i] = mode[i];
    look[i] = lookAhead ? lookAhead[i] : 12.0f;              // TrafficAgent::lookAheadDist, sc_traffic_common.h:32
    uint8_t& f = c->hFlags[first + i];
    const uint8_t nf = isAgent[i] ? (uint8_t)((f | 32u) & ~8u) : (uint8_t)(f & ~32u);
    if (nf != f) { f = nf; c->linksStale = true; }            // an agent's yaw is always streamed (the device rewrites it; X and Z it sets to (0, 1))
  }
  const bool ok = h2d(c, d.moverKind + first, kind.data(), (size_t)count * 4u) && h2d(c, d.aLane + first, laneId, (size_t)count * 4u) &&
                  h2d(c, d.aS + first, laneS, (size_t)count * 4u) && h2d(c, d.aSpeed + first, targetSpeed, (size_t)count * 4u) &&
                  h2d(c, d.aMode + first, md.data(), (size_t)count * 4u) && h2d(c, d.aLook + first, look.data(), (size_t)count * 4u);
  return ok && sync(c) ? 1 : 0;
}

int scTickSetTrafficSensors(ScTickContext* c, int enable, float frontRayLength, float safeDistance)
{
  if (!c) return 0;
  if (!bind(c) || !sync(c)) return 0;
  if (enable) {
    if (!c->d.aLane) return fail(c, "no traffic agents uploaded");
    if (!c->sectors) return fail(c, "the obstacle rays read the broadphase bins: the context has no tile rectangle");
    if (!(frontRayLength >= 0.0f) || !(safeDistance >= 0.0f)) return fail(c, "ray length and safe distance must be >= 0");
    if (!c->d.aBrake && (!dalloc(c, c->d.aBrake, c->cap) || !dalloc(c, c->d.agentList, c->cap, false) || !dalloc(c, c->d.agentCount, 4) ||
                         !dalloc(c, c->d.aRayLen, c->cap) || !dalloc(c, c->d.aSafe, c->cap) || !dalloc(c, c->d.aHitDist, c->cap) || !dalloc(c, c->d.aHitType, c->cap))) return 0;
    if (!c->d.agentRays && !dalloc(c, c->d.agentRays, 2u * (size_t)c->cap, false)) return 0;
    c->sensorRay = frontRayLength; c->sensorSafe = safeDistance;
    launchFillSensors(c->d, 0, c->cap, frontRayLength, safeDistance, c->stream);        // every agent's TrafficSensors = these defaults until scTickUploadTrafficSensors says otherwise
    if (!sync(c)) return 0;
  } else if (c->d.aBrake) {
    HIP_OK(c, hipMemsetAsync(c->d.aBrake, 0, (size_t)c->cap * sizeof(float), c->stream));      // no sensors: brake 0 from here on
    if (!sync(c)) return 0;
  }
  // On a tiled world an agent near a tile edge has to see the neighbour's boxes: the border messages then carry the halo section (the
  // sender's core-edge records), which changes their size -- so the sensors must be switched before the message buffers exist
  // (scTickBindBorderBuffers* / scTickCommInit), on every tile alike.
  const bool wantHalo = enable != 0;
  if (wantHalo != c->halo) {
    bool bound = c->comm != nullptr;
    for (uint32_t d = 0; d < 8 && !bound; ++d) bound = c->d.borderSend[d] || c->d.borderRecv[d];
    if (bound) return fail(c, "on a tile with border buffers or a communicator the traffic sensors cannot be switched any more: the messages' halo section changes their size (call scTickSetTrafficSensors before scTickBindBorderBuffers / scTickCommInit, on every tile)");
    c->halo = wantHalo;
  }
  c->sensors = enable != 0;
  dropGraph(c); dropPairGraph(c); c->topoEpoch++;
  return 1;
}

int scTickUploadTrafficSensors(ScTickContext* c, uint32_t first, uint32_t count, const float* frontRayLength, const float* safeDistance)
{
  if (!c || !frontRayLength || !safeDistance) return c ? fail(c, "null argument") : 0;
  if (!bind(c) || !rangeOk(c, first, count)) return 0;
  if (!c->d.aRayLen) return fail(c, "scTickSetTrafficSensors first");
  for (uint32_t i = 0; i < count; ++i)
    if (!(frontRayLength[i] >= 0.0f) || !(safeDistance[i] >= 0.0f)) return fail(c, "ray length and safe distance must be >= 0");
  if (!count) return 1;
  return h2d(c, c->d.aRayLen + first, frontRayLength, (size_t)count * 4u) && h2d(c, c->d.aSafe + first, safeDistance, (size_t)count * 4u) && sync(c) ? 1 : 0;
}

int scTickReadTrafficSensors(ScTickContext* c, uint32_t first, uint32_t count, float* lastHitDistance, uint8_t* lastHitType)
{
  if (!c || !lastHitDistance || !lastHitType) return c ? fail(c, "null argument") : 0;
  if (!bind(c) || !rangeOk(c, first, count)) return 0;
  if (!c->d.aHitDist) return fail(c, "scTickSetTrafficSensors first");
  if (!count) return 1;
  if (!joinPairs(c)) return 0;                            // (on a tiled world the rays are cast in the pair half)
  std::vector<uint32_t> typ(count);
  if (!d2h(c, lastHitDistance, c->d.aHitDist + first, (size_t)count * 4u) || !d2h(c, typ.data(), c->d.aHitType + first, (size_t)count * 4u) || !sync(c)) return 0;
  for (uint32_t i = 0; i < count; ++i) lastHitType[i] = (uint8_t)typ[i];
  return 1;
}

int scTickReadTrafficBrakes(ScTickContext* c, uint32_t first, uint32_t count, float* brake)
{
  if (!c || !brake) return c ? fail(c, "null argument") : 0;
  if (!bind(c) || !rangeOk(c, first, count)) return 0;
  if (!c->d.aBrake) return fail(c, "scTickSetTrafficSensors first");
  if (!count) return 1;
  return d2h(c, brake, c->d.aBrake + first, (size_t)count * 4u) && sync(c) ? 1 : 0;
}

int scTickReadTrafficAgents(ScTickContext* c, uint32_t first, uint32_t count, uint32_t* laneId, float* laneS, float* targetSpeed, uint8_t* mode)
{
  if (!c) return 0;
  if (!bind(c) || !rangeOk(c, first, count)) return 0;
  if (!c->d.aLane) return fail(c, "no traffic agents uploaded");
  if (!count) return 1;
  std::vector<uint32_t> md(count);
  bool ok = true;
  if (laneId) ok = ok && d2h(c, laneId, c->d.aLane + first, (size_t)count * 4u);
  if (laneS) ok = ok && d2h(c, laneS, c->d.aS + first, (size_t)count * 4u);
  if (targetSpeed) ok = ok && d2h(c, targetSpeed, c->d.aSpeed + first, (size_t)count * 4u);
  if (mode) ok = ok && d2h(c, md.data(), c->d.aMode + first, (size_t)count * 4u);
  if (!ok || !sync(c)) return 0;
  if (mode) for (uint32_t i = 0; i < count; ++i) mode[i] = (uint8_t)md[i];
  return 1;
}

int scTickSelectTrafficTiers(ScTickContext* c, const float playerPos[3], const ScTickTierParams* tp, ScTickTierCounts* out)
{
  if (!c || !playerPos || !tp) return c ? fail(c, "null argument") : 0;
  if (!bind(c) || !flushLinks(c)) return 0;
  if (!c->d.aLane) return fail(c, "no traffic agents uploaded");
  TierParams k;
  k.px = playerPos[0]; k.pz = playerPos[2];
  k.aEnter = tp->tier_a_enter; k.aExit = tp->tier_a_exit; k.bEnter = tp->tier_b_enter; k.bExit = tp->tier_b_exit;
  if (k.aExit < k.aEnter + 1.0f) k.aExit = k.aEnter + 1.0f;           // sc_traffic_lod.cpp:269-274
  if (k.bEnter < k.aExit + 1.0f) k.bEnter = k.aExit + 1.0f;
  if (k.bExit < k.bEnter + 1.0f) k.bExit = k.bEnter + 1.0f;
  HIP_OK(c, hipMemsetAsync(c->d.tierCounts, 0, 4 * sizeof(uint32_t), c->stream));
  launchTrafficTiers(c->d, c->n, k, c->stream);
  uint32_t cnt[4] = {};
  if (!d2h(c, cnt, c->d.tierCounts, sizeof cnt) || !sync(c)) return 0;
  if (cnt[3] > kTierNearCap) return fail(c, "more than 65536 agents want the Physics or Kinematic tier");
  std::vector<uint2> nearList(cnt[3]);
  if (cnt[3] && (!d2h(c, nearList.data(), c->d.tierNear, (size_t)cnt[3] * 8u) || !sync(c))) return 0;
  // the order ForEach would visit them in (Transform-pool dense order): the list arrives in atomic order
  std::sort(nearList.begin(), nearList.end(), [](const uint2& a, const uint2& b) { return a.x < b.x; });
  uint32_t physicsCount = cnt[0], kinematicCount = cnt[1], onRailsCount = cnt[2];
  std::vector<uint8_t> des(nearList.size());
  std::vector<float> dist(nearList.size());
  for (size_t q = 0; q < nearList.size(); ++q) {
    const uint32_t bits = nearList[q].y & 0x7FFFFFFFu;
    std::memcpy(&dist[q], &bits, 4);
    des[q] = (uint8_t)((nearList[q].y >> 31) ? kTierKinematic : kTierPhysics);
  }
  // the caps, sc_traffic_lod.cpp:355-417: candidates sorted by distance, descending (std::sort there; stable here, so equal
  // distances keep their pool order), everything past the cap is demoted
  std::vector<uint2> patches;
  auto byDistDesc = [&](uint32_t a, uint32_t b) { return dist[a] > dist[b]; };
  if (tp->max_physics > 0 && physicsCount > tp->max_physics) {
    std::vector<uint32_t> phys;
    for (uint32_t q = 0; q < des.size(); ++q) if (des[q] == kTierPhysics) phys.push_back(q);
    std::stable_sort(phys.begin(), phys.end(), byDistDesc);
    for (size_t a = tp->max_physics; a < phys.size(); ++a) {
      if (tp->max_kinematic == 0 || kinematicCount < tp->max_kinematic) { des[phys[a]] = (uint8_t)kTierKinematic; kinematicCount++; }
      else { des[phys[a]] = (uint8_t)kTierOnRails; onRailsCount++; }
      physicsCount--;
    }
  }
  if (tp->max_kinematic > 0 && kinematicCount > tp->max_kinematic) {
    std::vector<uint32_t> kin;
    for (uint32_t q = 0; q < des.size(); ++q) if (des[q] == kTierKinematic) kin.push_back(q);
    std::stable_sort(kin.begin(), kin.end(), byDistDesc);
    for (size_t a = tp->max_kinematic; a < kin.size(); ++a) { des[kin[a]] = (uint8_t)kTierOnRails; kinematicCount--; onRailsCount++; }
  }
  for (uint32_t q = 0; q < des.size(); ++q) patches.push_back(make_uint2(nearList[q].x, des[q]));
  if (!patches.empty() && !h2d(c, c->dTierPatch, patches.data(), patches.size() * 8u)) return 0;
  launchApplyTiers(c->d, c->n, c->dTierPatch, (uint32_t)patches.size(), c->stream);
  if (!sync(c)) return 0;
  if (out) { out->physics = physicsCount; out->kinematic = kinematicCount; out->on_rails = onRailsCount; out->total = physicsCount + kinematicCount + onRailsCount; }
  return 1;
}

int scTickSelectTrafficDespawns(ScTickContext* c, const float playerPos[3], uint32_t maxTotal, uint32_t* denseIndices, uint32_t capacity, uint32_t* count)
{
  if (!c || !playerPos || !count) return c ? fail(c, "null argument") : 0;
  *count = 0;
  if (!bind(c) || !flushLinks(c)) return 0;
  if (!c->d.aLane) return fail(c, "no traffic agents uploaded");
  if (maxTotal == 0 || !c->n) return 1;                                  // dbg.maxTrafficVehiclesTotal == 0: no cap (:421)
  // every agent with its ordering key, then the top of the order on the host (the list is short-lived scratch)
  uint32_t* dCount = nullptr; uint32_t* dIdx = nullptr; unsigned long long* dKey = nullptr;
  if (!dalloc(c, dCount, 4, false) || !dalloc(c, dIdx, c->n, false) || !dalloc(c, dKey, c->n, false)) return 0;
  // (zeroed on the tick stream itself: a hipMemset on the null stream is not ordered against a non-blocking stream)
  if (hipMemsetAsync(dCount, 0, 4 * sizeof(uint32_t), c->stream) != hipSuccess) { dfree(c, dCount); dfree(c, dIdx); dfree(c, dKey); return fail(c, "hipMemsetAsync"); }
  launchTrafficDespawnKeys(c->d, c->n, playerPos[0], playerPos[2], dCount, dIdx, dKey, c->stream);
  uint32_t agents = 0;
  int ok = d2h(c, &agents, dCount, sizeof agents) && sync(c);
  std::vector<uint32_t> idx; std::vector<unsigned long long> key;
  if (ok && agents > maxTotal) {
    idx.resize(agents); key.resize(agents);
    ok = d2h(c, idx.data(), dIdx, (size_t)agents * 4u) && d2h(c, key.data(), dKey, (size_t)agents * 8u) && sync(c);
  }
  dfree(c, dCount); dfree(c, dIdx); dfree(c, dKey);
  if (!ok) return 0;
  if (agents <= maxTotal) return 1;
  const uint32_t toRemove = agents - maxTotal;
  std::vector<uint32_t> order(agents);
  for (uint32_t k = 0; k < agents; ++k) order[k] = k;
  // OnRails before Kinematic before Physics, the farthest first; equal keys in pool order (std::sort there leaves it unspecified)
  auto before = [&](uint32_t a, uint32_t b) { return key[a] != key[b] ? key[a] > key[b] : idx[a] < idx[b]; };
  std::partial_sort(order.begin(), order.begin() + toRemove, order.end(), before);
  *count = toRemove;
  for (uint32_t k = 0; k < toRemove && k < capacity && denseIndices; ++k) denseIndices[k] = idx[order[k]];
  return 1;
}

int scTickSetWorldLayers(ScTickContext* c, uint32_t groupOr, uint32_t maskOr, int known)
{
  if (!c) return 0;
  if (!bind(c)) return 0;
  if (known && (((groupOr != 0xFFFFFFFFu) && (groupOr >> 16)) || ((maskOr != 0xFFFFFFFFu) && (maskOr >> 16)))) return fail(c, "group/mask bits above 15 are not supported");
  const uint32_t v = known ? ((groupOr & 0xFFFFu) | ((maskOr & 0xFFFFu) << 16)) : 0u;
  if (known) {                                        // the contract holds for what is already here, too
    const size_t upto = std::min<size_t>(c->hLayers.size(), c->n);
    for (size_t i = 0; i < upto; ++i)
      if ((c->hFlags[i] & 2u) && (c->hLayers[i] & ~v)) return fail(c, "the vocabulary does not cover the layers already uploaded to this tile");
  }
  if ((known != 0) != c->worldLayersKnown || v != c->worldLayers) {
    if (!sync(c)) return 0;
    c->worldLayersKnown = known != 0; c->worldLayers = v;
    c->homeValid = false;               // the slots' "written on every tick" bits follow the vocabulary
  }
  return 1;
}

int scTickGetLearnTicks(ScTickContext* c, uint32_t* learn_ticks)
{
  if (!c || !learn_ticks) return c ? fail(c, "null argument") : 0;
  *learn_ticks = c->learnTicks;            // (host-side: no read-back, no synchronisation)
  return 1;
}

int scTickGetBinStats(ScTickContext* c, uint32_t stats[4])
{
  if (!c || !stats) return c ? fail(c, "null argument") : 0;
  stats[0] = stats[1] = stats[2] = stats[3] = 0u;
  if (!bind(c)) return 0;
  if (!c->d.homeA || !c->n) return 1;
  stats[2] = (c->lastTickLazy ? 1u : 0u) | (c->lastTickStay ? 2u : 0u) | (c->lastTickSweepOnly ? 4u : 0u); stats[3] = c->learnTicks;
  if (!c->homeValid) return 1;
  if (!sync(c)) return 0;
  std::vector<uint32_t> a(c->n), b(c->n);
  if (!d2h(c, a.data(), c->d.homeA, (size_t)c->n * 4u) || !d2h(c, b.data(), c->d.homeB, (size_t)c->n * 4u) || !sync(c)) return 0;
  for (uint32_t i = 0; i < c->n; ++i) {
    if (a[i] == kNoHome) continue;
    for (uint32_t k = 0; k < 4u; ++k) {
      const uint32_t byte = (b[i] >> (8u * k)) & 0xFFu;
      if (byte == kNoSlot) continue;
      stats[0]++; if (byte & kSlotAlways) stats[1]++;
    }
  }
  return 1;
}

// ---- the border exchange, owned by the library -------------------------------------------------------------------
static const RcclApi* needRccl(ScTickContext* c)
{
  std::string why;
  const RcclApi* r = rccl(&why);
  if (!r) { if (c) fail(c, why.c_str()); else gCreateError = why; }
  return r;
}

static bool ncclOk(ScTickContext* c, const RcclApi* r, ncclResult_t res, const char* what)
{
  if (res == ncclSuccess) return true;
  char buf[256];
  std::snprintf(buf, sizeof buf, "%s: %s", what, r->GetErrorString(res));
  if (c) fail(c, buf); else gCreateError = buf;
  return false;
}

int scTickCommGetUniqueId(uint8_t id[SC_TICK_COMM_ID_BYTES])
{
  if (!id) return 0;
  const RcclApi* r = needRccl(nullptr);
  if (!r) return 0;
  static_assert(sizeof(ncclUniqueId) == SC_TICK_COMM_ID_BYTES, "ncclUniqueId is 128 bytes");
  ncclUniqueId u;
  if (!ncclOk(nullptr, r, r->GetUniqueId(&u), "ncclGetUniqueId")) return 0;
  std::memcpy(id, &u, sizeof u);
  return 1;
}

int scTickCommDestroy(ScTickContext* c)
{
  if (!c) return 0;
  if (!c->comm) return 1;
  if (!bind(c)) return 0;
  sync(c);
  const RcclApi* r = needRccl(c);
  // captured steps hold ncclSend / ncclRecv nodes bound to this communicator and to the peers' ranks
  dropGraph(c); dropPairGraph(c); c->topoEpoch++;
  if (r) r->CommDestroy(c->comm);
  c->comm = nullptr; c->commSize = 0; c->commRank = 0;
  return 1;
}

int scTickCommSetPeers(ScTickContext* c, const int32_t peerRank[8])
{
  if (!c || !peerRank) return c ? fail(c, "null argument") : 0;
  if (!bind(c) || !sync(c)) return 0;
  dropGraph(c); dropPairGraph(c); c->topoEpoch++;       // a captured exchange names the old ranks
  for (int d = 0; d < 8; ++d) c->peer[d] = peerRank[d];
  c->peersSet = true;
  return 1;
}

int scTickCommInit(ScTickContext* c, const uint8_t id[SC_TICK_COMM_ID_BYTES], uint32_t worldSize, uint32_t rank)
{
  if (!c || !id) return c ? fail(c, "null argument") : 0;
  if (!worldSize || rank >= worldSize) return fail(c, "need rank < world_size");
  if (!c->sectors) return fail(c, "the context has no broadphase: nothing to exchange");
  if (!c->tilesX) return fail(c, "scTickSetTileGrid first (the neighbours' ranks follow from the tile's place in the grid)");
  if (c->pairsPending) return fail(c, "scTickRunPairs is pending");
  if (!bind(c) || !sync(c)) return 0;
  const RcclApi* r = needRccl(c);
  if (!r) return 0;
  if (c->comm && !scTickCommDestroy(c)) return 0;
  if (!c->peersSet) {
    // tiles in row-major rank order, as the entities are created tile-major
    if ((uint64_t)c->tilesX * c->tilesZ != worldSize) return fail(c, "world_size differs from the tile grid (use scTickCommSetPeers for another rank layout)");
    for (uint32_t d = 0; d < 8; ++d) {
      int dx, dz; borderDir(d, dx, dz);
      c->peer[d] = ((c->neighbourMask >> d) & 1u) ? (int32_t)(((int)c->tileZ + dz) * (int)c->tilesX + (int)c->tileX + dx) : -1;
    }
  }
  for (uint32_t d = 0; d < 8; ++d)
    if (((c->neighbourMask >> d) & 1u) && (c->peer[d] < 0 || (uint32_t)c->peer[d] >= worldSize)) return fail(c, "a neighbour's rank lies outside the communicator");
  ncclUniqueId u;
  std::memcpy(&u, id, sizeof u);
  if (!ncclOk(c, r, r->CommInitRank(&c->comm, (int)worldSize, u, (int)rank), "ncclCommInitRank")) { c->comm = nullptr; return 0; }
  c->commSize = worldSize; c->commRank = rank;
  dropGraph(c); dropPairGraph(c); c->topoEpoch++;       // (a graph captured against an earlier communicator must not be replayed)
  // the messages of both tick parities live in buffers of the library's own (a caller that runs its own transport binds
  // its buffers with scTickBindBorderBuffers instead and never comes here)
  for (uint32_t q = 0; q < kMaxParity; ++q)
    for (uint32_t d = 0; d < 8; ++d) {
      if (!((c->neighbourMask >> d) & 1u)) continue;
      const size_t words = borderWords(d, c->desc.tile_sectors_x, c->desc.tile_sectors_z, c->borderRecs, c->halo ? 1u : 0u);
      for (int k = 0; k < 2; ++k) if (!c->ownBorder[q][d][k] && !dalloc(c, c->ownBorder[q][d][k], words)) return 0;
      if (q == 0) { c->d.borderSend[d] = c->ownBorder[0][d][0]; c->d.borderRecv[d] = c->ownBorder[0][d][1]; }
      else { c->alt[q - 1u].borderSend[d] = c->ownBorder[q][d][0]; c->alt[q - 1u].borderRecv[d] = c->ownBorder[q][d][1]; }
    }
  return sync(c) ? 1 : 0;
}

int scTickGetCommInfo(ScTickContext* c, ScTickCommInfo* out)
{
  if (!c || !out) return c ? fail(c, "null argument") : 0;
  std::memset(out, 0, sizeof *out);
  out->world_size = c->commSize; out->rank = c->commRank; out->has_communicator = c->comm ? 1u : 0u;
  out->neighbour_mask = c->neighbourMask;
  out->pipeline_depth = c->pairsStream ? c->pipeDepth : 0u;
  out->border_records_per_sector = c->borderRecs;
  for (uint32_t d = 0; d < 8; ++d) {
    out->peer_rank[d] = ((c->neighbourMask >> d) & 1u) ? c->peer[d] : -1;
    if (!((c->neighbourMask >> d) & 1u) || !c->sectors) continue;
    out->operations_per_group += 2u;                                   // one ncclSend + one ncclRecv per neighbour
    out->bytes_sent_per_step += (uint64_t)borderWords(d, c->desc.tile_sectors_x, c->desc.tile_sectors_z, c->borderRecs, c->halo ? 1u : 0u) * 4u;
  }
  if (c->comm) { std::string why; if (const RcclApi* r = rccl(&why)) { int v = 0; if (r->GetVersion(&v) == ncclSuccess) out->rccl_version = (uint32_t)v; } }
  out->host_steps = c->hostSteps;
  if (g_probe.on && g_probe.n) {
    std::fprintf(stderr, "[sc_tick host probe] %llu ticks since the reset: waitParityFree %.2f us, enqueueStages %.2f (fused launch %.2f, compaction+pack launch %.2f), publishPacked %.2f, before %.2f, after %.2f\n",
                 (unsigned long long)g_probe.n, g_probe.acc[0] / g_probe.n, g_probe.acc[1] / g_probe.n, g_probe.acc[3] / g_probe.n, g_probe.acc[4] / g_probe.n, g_probe.acc[2] / g_probe.n,
                 g_probe.acc[5] / g_probe.n, g_probe.acc[6] / g_probe.n);
    g_probe = HostProbe(); g_probe.on = true;
  }
  out->host_tick_half_us = c->hostSteps ? c->hostAcc[0] / (double)c->hostSteps : 0.0;
  out->host_pair_half_us = c->hostSteps ? c->hostAcc[1] / (double)c->hostSteps : 0.0;
  return 1;
}

int scTickGatherVisibleCounts(ScTickContext* c, uint32_t* countsOut, uint32_t capacity, uint64_t* offsetOut, uint64_t* totalOut)
{
  if (!c || !offsetOut || !totalOut) return c ? fail(c, "null argument") : 0;
  if (!bind(c) || !sync(c)) return 0;                      // both streams idle: the collective below is the communicator's only operation in flight
  const uint32_t ranks = c->comm ? c->commSize : 1u;
  std::vector<uint32_t> counts(ranks, 0u);
  if (!c->comm) {
    if (!d2h(c, counts.data(), c->d.counters, sizeof(uint32_t)) || !sync(c)) return 0;
  } else {
    const RcclApi* r = needRccl(c);
    if (!r) return 0;
    if (c->gatherCap < ranks) {
      dfree(c, c->dGather); c->dGather = nullptr; c->gatherCap = 0;
      if (!dalloc(c, c->dGather, ranks)) return 0;
      c->gatherCap = ranks;
    }
    if (!ncclOk(c, r, r->AllGather(c->d.counters, c->dGather, 1, ncclUint32, c->comm, c->stream), "ncclAllGather (visible counts)")) return 0;
    if (!d2h(c, counts.data(), c->dGather, (size_t)ranks * sizeof(uint32_t)) || !sync(c)) return 0;
  }
  uint64_t before = 0, total = 0;
  const uint32_t me = c->comm ? c->commRank : 0u;
  for (uint32_t k = 0; k < ranks; ++k) { if (k < me) before += counts[k]; total += counts[k]; }
  if (countsOut) for (uint32_t k = 0; k < ranks && k < capacity; ++k) countsOut[k] = counts[k];
  *offsetOut = before; *totalOut = total;
  return 1;
}

int scTickResetHostTimes(ScTickContext* c)
{
  if (!c) return 0;
  c->hostAcc[0] = c->hostAcc[1] = 0.0; c->hostSteps = 0;
  if (g_probe.on) { g_probe = HostProbe(); g_probe.on = true; }
  return 1;
}

int scTickSetPipelined(ScTickContext* c, int enable)
{
  if (!c) return 0;
  if (!bind(c)) return 0;
  if (!enable) return scTickSetPairsStream(c, nullptr);
  if (enable < 0 || enable > (int)kMaxParity) return fail(c, "pipeline depth must be 2..4 (1 = the default, 4; 0 = off)");
  // depth d: the pair half of a tick may take up to d - 2 ticks before it holds anything up (a tick's counters are cleared by
  // the tick before it, which therefore waits for the pair half d - 1 ticks back).  4 on the loop-back: 56 us per step
  // against 63 with 3; 2 leaves no overlap of the pair half with the next tick at all.
  const uint32_t depth = enable == 1 ? 4u : (uint32_t)enable;
  if (depth != c->pipeDepth) {
    if (c->pairsStream && !scTickSetPairsStream(c, nullptr)) return 0;     // re-enter with the new depth
    c->pipeDepth = depth;
  }
  // (a high-priority pairs stream was measured: no gain, 94 against 90 us per step with the loop-back exchange)
  if (!c->ownPairsStream) HIP_OK(c, hipStreamCreateWithFlags(&c->ownPairsStream, hipStreamNonBlocking));
  return scTickSetPairsStream(c, c->ownPairsStream);
}

// send[d] of this tile -> recv[7-d] of the neighbour in direction d, every existing neighbour, one RCCL group on `s`.
// A pair of ranks normally shares one edge or corner, i.e. one message each way; should a rank layout put several
// neighbours on one rank (scTickCommSetPeers: a periodic world, the loop-back test), point-to-point operations between
// two ranks match in posting order: sends go out by ascending direction, so the receives are posted by DESCENDING
// direction -- what a peer sent as its direction d arrives here as direction 7-d.
// inCapture: what the caller believes about `s`.  The group is only opened when the stream's capture state is that one: an
// RCCL group posted to a stream that is (or is not) being captured against the caller's expectation -- a capture that was
// invalidated half way, a stream forked into somebody else's capture -- is refused here instead of being handed to RCCL.
static int exchangeBorders(ScTickContext* c, uint32_t parity, hipStream_t s, bool inCapture)
{
  const RcclApi* r = needRccl(c);
  if (!r) return 0;
  if (!c->comm) return fail(c, "no communicator: scTickCommInit first");
  hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
  const hipError_t qe = hipStreamIsCapturing(s, &st);
  if (qe != hipSuccess) return fail(c, "hipStreamIsCapturing", qe);
  if ((st == hipStreamCaptureStatusActive) != inCapture)
    return fail(c, inCapture ? "border exchange: the stream is not being captured although a capture was begun (capture invalidated?)"
                             : "border exchange: the stream is being captured by somebody else");
  const DeviceState ds = stateFor(c, parity);
  if (!ncclOk(c, r, r->GroupStart(), "ncclGroupStart")) return 0;
  ncclResult_t res = ncclSuccess;
  for (int d = 0; d < 8 && res == ncclSuccess; ++d) {
    if (!((c->neighbourMask >> d) & 1u)) continue;
    if (!ds.borderSend[d] || !ds.borderRecv[d]) { r->GroupEnd(); return fail(c, "border buffers are not bound"); }
    res = r->Send(ds.borderSend[d], (size_t)borderWords((uint32_t)d, c->desc.tile_sectors_x, c->desc.tile_sectors_z, c->borderRecs, c->halo ? 1u : 0u), ncclUint32, c->peer[d], c->comm, s);
  }
  for (int d = 7; d >= 0 && res == ncclSuccess; --d) {
    if (!((c->neighbourMask >> d) & 1u)) continue;
    res = r->Recv(ds.borderRecv[d], (size_t)borderWords((uint32_t)d, c->desc.tile_sectors_x, c->desc.tile_sectors_z, c->borderRecs, c->halo ? 1u : 0u), ncclUint32, c->peer[d], c->comm, s);
  }
  const ncclResult_t end = r->GroupEnd();
  if (!ncclOk(c, r, res, "ncclSend/ncclRecv")) return 0;
  return ncclOk(c, r, end, "ncclGroupEnd") ? 1 : 0;
}

int scTickExchangeBorders(ScTickContext* c)
{
  if (!c) return 0;
  if (!bind(c)) return 0;
  if (!c->pairsPending) return fail(c, "scTickExchangeBorders without a preceding scTickRun(... | SC_TICK_BROADPHASE | SC_TICK_SPLIT_PAIRS)");
  if (!c->neighbourMask) return 1;
  if (!c->comm) return fail(c, "no communicator: scTickCommInit first");
  return exchangeBorders(c, c->pendingParams.parity, c->pairsStream ? c->pairsStream : c->stream, false);
}

int scTickTileStep(ScTickContext* c, uint32_t flags)
{
  if (!c) return 0;
  struct Own { ScTickContext* c; explicit Own(ScTickContext* c_) : c(c_) { c->ownStep = true; } ~Own() { c->ownStep = false; } } own(c);
  if (!(flags & SC_TICK_BROADPHASE) || !c->neighbourMask) {                                                                // nothing to exchange
    if (!(flags & SC_TICK_BROADPHASE) || !c->pairsStream) return scTickRun(c, flags & ~(uint32_t)SC_TICK_SPLIT_PAIRS);
    if (!scTickRun(c, flags | SC_TICK_SPLIT_PAIRS)) return 0;                 // a lone pipelined tile: the pair half still runs on its own stream
    return runPendingPairs(c, false);
  }
  if (!c->comm) return fail(c, "no communicator: scTickCommInit first (a tile with neighbours cannot skip the exchange)");
  auto now = [] { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  if (c->graphMode && !c->pairsStream) {
    // in-order tile, graph replay: the whole step -- producer, fused kernel, compaction + pack, the RCCL group, merge, pair
    // search -- is captured once per tick parity and replayed with one hipGraphLaunch
    const double g0 = now();
    c->captureWholeStep = true;
    const int okr = scTickRun(c, flags | SC_TICK_SPLIT_PAIRS);
    c->captureWholeStep = false;
    if (!okr) return 0;
    const int okp = c->pairsPending ? runPendingPairs(c, true) : 1;     // a sampled (profiled) tick ran eagerly: finish it the eager way
    c->hostAcc[0] += now() - g0; c->hostSteps++;
    return okp;
  }
  // eager, or a pipelined tile: the tick on its stream, then exchange + pair half on theirs (with graph replay on, each
  // half of a pipelined step is one hipGraphLaunch; the events that order them are recorded between the two)
  const double t0 = now();
  if (!scTickRun(c, flags | SC_TICK_SPLIT_PAIRS)) return 0;
  const double t1 = now();
  const int ok = runPendingPairs(c, true);
  c->hostAcc[0] += t1 - t0; c->hostAcc[1] += now() - t1; c->hostSteps++;
  return ok;
}

} // extern "C"
