// sc_tick_queries.hip -- ray queries over the broadphase bins of the current tick.
//
// Shaped like PhysicsWorld::raycast (src/engine/physics/sc_physics.cpp:740-777): direction normalised the same
// way (rejected when |dir|^2 <= 1e-6), segment from origin to origin + ndir * maxDist, Bullet's default filter
// with the callback's group 0xFFFF and mask = the caller's mask, closest hit wins.  The reference tests Bullet's
// exact shapes; Bullet is not in the tree, so -- as for the pair search -- the spec here is this build's own:
// candidates are the WORLD AABBs the broadphase already holds, and the ray-box arithmetic is the reference's
// own slab test, intersectRayAABB (tools/world_editor/editor_core/editor_core.cpp:438-470), with the far limit
// set to maxDist.  Equal distances go to the lower id (PickEntity keeps the first, :487-491).
//
// One wave per ray.  The wave walks the sectors under the ray's xz extent, skipping those the segment misses;
// lanes take one bin record each; then the big list.  Every lane keeps its own best hit; one 64-bit min over
// (distance bits, id) picks the winner, whose lane writes the result.
#include "sc_tick_internal.h"

namespace sctick {

namespace {

struct Slab { bool hit; float t; uint32_t axis; };

// intersectRayAABB, editor_core.cpp:438-470 (tmax starts at the ray's length instead of 1e30)
__device__ __forceinline__ Slab rayBox(const float o[3], const float dir[3], float maxDist, const float4& lo, const float4& hi)
{
  const float mn[3] = { lo.x, lo.y, lo.z }, mx[3] = { hi.x, hi.y, hi.z };
  Slab s; s.hit = true; s.t = 0.0f; s.axis = 3u;
  float tmin = 0.0f, tmax = maxDist;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    if (fabsf(dir[i]) < 1e-6f) {
      if (o[i] < mn[i] || o[i] > mx[i]) s.hit = false;
    } else {
      const float ood = 1.0f / dir[i];
      float t1 = (mn[i] - o[i]) * ood, t2 = (mx[i] - o[i]) * ood;
      if (t1 > t2) { const float k = t1; t1 = t2; t2 = k; }
      if (t1 > tmin) { tmin = t1; s.axis = (uint32_t)i; }          // std::max(tmin, t1)
      tmax = tmax < t2 ? tmax : t2;                                 // std::min(tmax, t2)
      if (tmin > tmax) s.hit = false;
    }
  }
  s.t = tmin;
  return s;
}

// One ray, cast by a whole wave (every lane must call): `valid` false = no segment (|dir|^2 <= 1e-6 or a negative length).
// The result is wave-uniform.  skipId: a box that never answers (0xFFFFFFFF = none) -- an agent's own box for its front ray.
struct WaveRay { bool valid, hit; float t; uint32_t id, axis, layer; float dir[3]; };

__device__ __forceinline__ WaveRay castRayWave(const DeviceState& d, const TickParams& p, const float o[3], const float dm[3], float maxDist,
                                               uint32_t rayMask, uint32_t skipId)
{
  const uint32_t lane = threadIdx.x & 63u;
  WaveRay w; w.valid = false; w.hit = false; w.t = 0.0f; w.id = 0xFFFFFFFFu; w.axis = 3u; w.layer = 0u; w.dir[0] = w.dir[1] = w.dir[2] = 0.0f;
  const float lenSq = dm[0] * dm[0] + dm[1] * dm[1] + dm[2] * dm[2];
  // (a NaN or non-positive length is no segment either)
  if (!(lenSq > 1e-6f) || !(maxDist >= 0.0f)) return w;
  w.valid = true;
  const float invLen = 1.0f / sqrtf(lenSq);
  const float dir[3] = { dm[0] * invLen, dm[1] * invLen, dm[2] * invLen };
  w.dir[0] = dir[0]; w.dir[1] = dir[1]; w.dir[2] = dir[2];
  const float ex = o[0] + dir[0] * maxDist, ez = o[2] + dir[2] * maxDist;

  // this lane's best so far
  float bt = INFINITY; uint32_t bid = 0xFFFFFFFFu, baxis = 3u, blayer = 0u;
  auto consider = [&](const float4& lo, const float4& hi) {
    const uint32_t lay = __float_as_uint(lo.w);
    // Bullet's needsCollision with the callback's group 0xFFFF: (proxy.group & mask) && (0xFFFF & proxy.mask)
    if (!((lay & 0xFFFFu) & rayMask) || !(lay >> 16)) return;
    const uint32_t id = __float_as_uint(hi.w) & ~kPrimary;
    if (id == skipId) return;
    const Slab s = rayBox(o, dir, maxDist, lo, hi);
    if (!s.hit) return;
    if (s.t < bt || (s.t == bt && id < bid)) { bt = s.t; bid = id; baxis = s.axis; blayer = lay & 0xFFFFu; }
  };

  // sectors under the segment's xz extent, clamped to the bin grid (boxes outside it are in the big list)
  const float fx0 = floorf((o[0] < ex ? o[0] : ex) * p.invSector) - p.binOx, fx1 = floorf((o[0] < ex ? ex : o[0]) * p.invSector) - p.binOx;
  const float fz0 = floorf((o[2] < ez ? o[2] : ez) * p.invSector) - p.binOz, fz1 = floorf((o[2] < ez ? ez : o[2]) * p.invSector) - p.binOz;
  const float gridX = (float)p.binSX - 1.0f, gridZ = (float)p.binSZ - 1.0f;
  bool anyOverflow = false;
  if (p.binSX && fx1 >= 0.0f && fz1 >= 0.0f && fx0 <= gridX && fz0 <= gridZ) {
    const uint32_t gx0 = (uint32_t)(fx0 < 0.0f ? 0.0f : fx0), gx1 = (uint32_t)(fx1 > gridX ? gridX : fx1);
    const uint32_t gz0 = (uint32_t)(fz0 < 0.0f ? 0.0f : fz0), gz1 = (uint32_t)(fz1 > gridZ ? gridZ : fz1);
    const float size = 1.0f / p.invSector;
    for (uint32_t gz = gz0; gz <= gz1; ++gz)
      for (uint32_t gx = gx0; gx <= gx1; ++gx) {
        if (gx1 - gx0 > 1u || gz1 - gz0 > 1u) {
          // long ray: skip sectors the segment cannot touch (the sector's square, grown by a metre, as a flat box)
          const float4 lo = make_float4(((float)gx + p.binOx) * size - 1.0f, -INFINITY, ((float)gz + p.binOz) * size - 1.0f, 0.0f);
          const float4 hi = make_float4(((float)gx + p.binOx + 1.0f) * size + 1.0f, INFINITY, ((float)gz + p.binOz + 1.0f) * size + 1.0f, 0.0f);
          if (!rayBox(o, dir, maxDist, lo, hi).hit) continue;
        }
        const uint32_t s = gz * p.binSX + gx;
        uint32_t n = d.binCount[s];
        if (n > kBinCap) { n = kBinCap; anyOverflow = true; }
        if (lane < n) {
          const float4* rec = d.bins + 2u * ((size_t)s * kBinCap + lane);
          consider(rec[0], rec[1]);
        }
      }
  }
  const uint32_t nbig = min(d.counters[kCtrPar + 8u * p.parity + kCtrBig], p.bigCap);
  for (uint32_t b = lane; b < nbig; b += 64u) consider(d.bigList[2u * (size_t)b], d.bigList[2u * (size_t)b + 1u]);
  // records that found their sector's bin full -- this tile's own boxes and a neighbour's border records (k_border_merge)
  // -- live only in the sector overflow list; they are this tile's to answer for like the records in the bins.  Duplicates
  // of a box that is also binned elsewhere are harmless: the same box gives the same distance and the id breaks the tie.
  // (Only a ray that crossed a sector holding more than its bin can meet one of them: a record of the list belongs to a
  //  sector whose counter passed 64.)
  if (anyOverflow) {
    const uint32_t nspill = min(d.counters[kCtrPar + 8u * p.parity + kCtrSpill], p.ovfCap);
    for (uint32_t e = lane; e < nspill; e += 64u) consider(d.spill[2u * (size_t)e], d.spill[2u * (size_t)e + 1u]);
  }

  // closest hit of the wave: distances are >= 0, so their bit patterns order like the values
  unsigned long long key = ((unsigned long long)__float_as_uint(bt) << 32) | bid;
  unsigned long long best = key;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const unsigned long long other = __shfl_xor(best, off, 64);
    best = other < best ? other : best;
  }
  const bool found = bid != 0xFFFFFFFFu && key == best;
  const unsigned long long winners = ballot64(found);
  if (!winners) return w;
  const int win = __ffsll((long long)winners) - 1;
  w.hit = true;
  w.t = __shfl(bt, win, 64); w.id = __shfl(bid, win, 64); w.axis = __shfl(baxis, win, 64); w.layer = __shfl(blayer, win, 64);
  return w;
}

__global__ __launch_bounds__(kTile) void k_ray_queries(const DeviceState d, const TickParams p, const RayQueryState q)
{
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t r = blockIdx.x * (kTile / 64u) + (threadIdx.x >> 6);
  if (r >= q.count) return;
  const float4 od = q.origin[r], dm = q.dir[r];
  const float o[3] = { od.x, od.y, od.z };
  const float dv[3] = { dm.x, dm.y, dm.z };
  RayHit48 out;
  out.hit = 0u; out.id = 0xFFFFFFFFu; out.distance = 0.0f;
  out.position[0] = out.position[1] = out.position[2] = 0.0f;
  out.normal[0] = 0.0f; out.normal[1] = 1.0f; out.normal[2] = 0.0f;           // RaycastHit{} (sc_physics.h:106-114)
  out.layer = 0u; out.pad = 0u; out.pad2 = 0u;
  const WaveRay w = castRayWave(d, p, o, dv, od.w, __float_as_uint(dm.w), 0xFFFFFFFFu);
  if (w.hit) {
    out.hit = 1u; out.id = w.id; out.distance = w.t; out.layer = w.layer;
    out.position[0] = o[0] + w.dir[0] * w.t; out.position[1] = o[1] + w.dir[1] * w.t; out.position[2] = o[2] + w.dir[2] * w.t;
    if (w.axis < 3u) {                                  // the face the ray entered through; a ray starting inside keeps (0,1,0)
      out.normal[0] = out.normal[1] = out.normal[2] = 0.0f;
      out.normal[w.axis] = w.dir[w.axis] > 0.0f ? -1.0f : 1.0f;
    }
  }
  if (lane == 0) q.hits[r] = out;
}

// ---- the traffic AI's obstacle ray (src/engine/traffic/sc_traffic_ai.cpp:300-345) --------------------------------------------
// Every agent of the OnRails tier casts one ray per step from 1.7 m ahead of its origin (0.6 m up) along its heading --
// forward = normalize(sin(yaw), 0, cos(yaw)) with the yaw's sin / cos as the entity's rotation streams hold them (host libm) --
// of TrafficSensors::frontRayLength (20 m) with mask 1, and brakes by clamp01((safe - d) / safe) when something other than
// itself is closer than TrafficSensors::safeDistance (10 m).  As for the ray queries above the candidates are the world AABBs
// of this tick's bins (own spec: Bullet is absent); the agent's own box never answers (Bullet does not report a convex shape a
// ray starts inside, and the reference discards a self hit: :322-325, :336).  The brake waits in aBrake[] for the on-rails
// step that produces the next frame.
__global__ __launch_bounds__(kTile) void k_list_onrails_agents(const DeviceState d, uint32_t n)
{
  const uint32_t i = blockIdx.x * kTile + threadIdx.x;
  const bool agent = i < n && d.moverKind[i] == kMoverTraffic && d.aMode[i] == kTierOnRails;
  const unsigned long long m = ballot64(agent);
  if (!m) return;
  const uint32_t lane = threadIdx.x & 63u;
  uint32_t base = 0;
  if (lane == 0) base = atomicAdd(d.agentCount, (uint32_t)__popcll(m));
  base = __shfl(base, 0, 64);
  if (agent) d.agentList[base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = i;
}

// Tiled world, in-order step: the rays are cast in the PAIR half, behind the border merge, so that an agent near a tile edge sees the
// neighbour tile's boxes too (the halo section of the border messages puts them in the ring bins).  The tick half only lists the agents
// and notes each one's ray as it stands -- the frame producer fused into the end-of-tick kernel moves the agents on before the pair
// half runs.  Same arithmetic as k_agent_front_rays.
__global__ __launch_bounds__(kTile) void k_list_onrails_agents_with_rays(const DeviceState d, uint32_t n)
{
  const uint32_t i = blockIdx.x * kTile + threadIdx.x;
  const bool agent = i < n && d.moverKind[i] == kMoverTraffic && d.aMode[i] == kTierOnRails;
  const unsigned long long m = ballot64(agent);
  if (!m) return;
  const uint32_t lane = threadIdx.x & 63u;
  uint32_t base = 0;
  if (lane == 0) base = atomicAdd(d.agentCount, (uint32_t)__popcll(m));
  base = __shfl(base, 0, 64);
  if (agent) {
    const uint32_t k = base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
    d.agentList[k] = i;
    float forward[3] = { d.rsy[i], 0.0f, d.rcy[i] };
    const float len = sqrtf(forward[0] * forward[0] + forward[1] * forward[1] + forward[2] * forward[2]);
    if (len > 1e-6f) { const float inv = 1.0f / len; forward[0] *= inv; forward[1] *= inv; forward[2] *= inv; }
    d.agentRays[2u * k] = make_float4(d.px[i] + forward[0] * 1.7f, d.py[i] + 0.6f, d.pz[i] + forward[2] * 1.7f, d.aRayLen[i]);
    d.agentRays[2u * k + 1u] = make_float4(forward[0], forward[2], d.aSafe[i], 0.0f);
  }
}

__global__ __launch_bounds__(kTile) void k_agent_front_rays_from_snapshot(const DeviceState d, const TickParams p)
{
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t count = *d.agentCount;
  const uint32_t waves = gridDim.x * (kTile / 64u);
  for (uint32_t k = blockIdx.x * (kTile / 64u) + (threadIdx.x >> 6); k < count; k += waves) {
    const uint32_t i = d.agentList[k];
    const float4 a = d.agentRays[2u * k], b = d.agentRays[2u * k + 1u];
    const float origin[3] = { a.x, a.y, a.z }, forward[3] = { b.x, 0.0f, b.y };
    const float rayLen = a.w, safe = b.z;
    const WaveRay w = castRayWave(d, p, origin, forward, rayLen, 1u, i | p.rankBits);
    float brake = 0.0f;
    if (w.hit && safe > 1e-3f && w.t < safe) {
      const float v = (safe - w.t) / safe;
      const float m = (1.0f < v) ? 1.0f : v;
      brake = (0.0f < m) ? m : 0.0f;
    }
    if (lane == 0) {
      d.aBrake[i] = brake;
      d.aHitDist[i] = w.hit ? w.t : rayLen;
      uint32_t kind = 0u;
      if (w.hit) {
        const bool own = (w.id & 0x7F000000u) == p.rankBits;
        const uint32_t mk = own ? d.moverKind[w.id & 0x00FFFFFFu] : 0u;
        kind = own ? ((mk == 1u || mk == kMoverTraffic) ? 2u : 3u) : ((w.layer & 1u) ? 2u : 3u);
      }
      d.aHitType[i] = kind;
    }
  }
}

__global__ __launch_bounds__(kTile) void k_fill_sensors(const DeviceState d, uint32_t first, uint32_t count, float rayLen, float safe)
{
  const uint32_t t = blockIdx.x * kTile + threadIdx.x;
  if (t < count) { d.aRayLen[first + t] = rayLen; d.aSafe[first + t] = safe; }
}

// (per-agent TrafficSensors: every agent casts with its own frontRayLength and brakes by its own safeDistance, :306-308; the
//  hit's distance and kind are left for the host as lastHitDistance / lastHitType, :339-345 -- Vehicle when the hit entity is a
//  vehicle by its mover kind (a traffic agent or a SynthWorld vehicle: what carries a VehicleComponent, :327), World otherwise;
//  a box that arrived from a neighbour tile has no mover kind here: Vehicle when its group has the dynamic bit; without a hit the
//  ray's length and None)
__global__ __launch_bounds__(kTile) void k_agent_front_rays(const DeviceState d, const TickParams p)
{
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t count = *d.agentCount;
  const uint32_t waves = gridDim.x * (kTile / 64u);
  for (uint32_t k = blockIdx.x * (kTile / 64u) + (threadIdx.x >> 6); k < count; k += waves) {
    const uint32_t i = d.agentList[k];
    float forward[3] = { d.rsy[i], 0.0f, d.rcy[i] };                 // { sin(currentYaw), 0, cos(currentYaw) }, :304
    const float len = sqrtf(forward[0] * forward[0] + forward[1] * forward[1] + forward[2] * forward[2]);      // normalize3, :38-48
    if (len > 1e-6f) { const float inv = 1.0f / len; forward[0] *= inv; forward[1] *= inv; forward[2] *= inv; }
    const float origin[3] = { d.px[i] + forward[0] * 1.7f, d.py[i] + 0.6f, d.pz[i] + forward[2] * 1.7f };       // :311-315
    const float rayLen = d.aRayLen[i], safe = d.aSafe[i];            // :306-308
    const WaveRay w = castRayWave(d, p, origin, forward, rayLen, 1u, i | p.rankBits);
    float brake = 0.0f;
    if (w.hit && safe > 1e-3f && w.t < safe) {
      const float v = (safe - w.t) / safe;                            // clamp01 = std::max(0, std::min(v, 1)), :16-24
      const float m = (1.0f < v) ? 1.0f : v;
      brake = (0.0f < m) ? m : 0.0f;
    }
    if (lane == 0) {
      d.aBrake[i] = brake;
      d.aHitDist[i] = w.hit ? w.t : rayLen;                           // :319-322, :341-345
      uint32_t kind = 0u;
      if (w.hit) {
        const bool own = (w.id & 0x7F000000u) == p.rankBits;
        const uint32_t mk = own ? d.moverKind[w.id & 0x00FFFFFFu] : 0u;
        kind = own ? ((mk == 1u || mk == kMoverTraffic) ? 2u : 3u) : ((w.layer & 1u) ? 2u : 3u);
      }
      d.aHitType[i] = kind;
    }
  }
}

// isOccupiedWorld (src/engine/traffic/sc_traffic_spawner.cpp:93-116): is any agent closer than `radius` to the point, in the
// xz plane, by Transform::localPos -- dx*dx + dz*dz < radius*radius, strictly.  The reference walks the TrafficAgent and
// VehicleComponent pools; here an entity counts when its collision group meets the query's mask.  A handful of
// queries per frame (spawn attempts) against every entity: one thread per entity, all queries from LDS.
__global__ __launch_bounds__(kTile) void k_occupancy(const DeviceState d, uint32_t n, const float4* __restrict__ q, uint32_t count,
                                                     uint32_t* __restrict__ blocked)
{
  __shared__ float4 sq[kMaxOccupancyQueries];
  for (uint32_t k = threadIdx.x; k < count; k += kTile) sq[k] = q[k];
  __syncthreads();
  const uint32_t i = blockIdx.x * kTile + threadIdx.x;
  if (i >= n) return;
  const uint32_t group = d.layers[i] & 0xFFFFu;
  if (!group) return;
  const float x = d.px[i], z = d.pz[i];
  for (uint32_t k = 0; k < count; ++k) {
    const float4 c = sq[k];                              // (x, z, radius, mask)
    if (!(group & __float_as_uint(c.w))) continue;
    const float dx = x - c.x, dz = z - c.y;
    if (dx * dx + dz * dz < c.z * c.z) atomicOr(&blocked[k >> 5], 1u << (k & 31u));
  }
}

} // namespace

void launchOccupancy(const DeviceState& d, uint32_t n, const float4* q, uint32_t count, uint32_t* blocked, hipStream_t s)
{
  if (!n || !count) return;
  hipLaunchKernelGGL(k_occupancy, dim3((n + kTile - 1) / kTile), dim3(kTile), 0, s, d, n, q, count, blocked);
}

void launchFillSensors(const DeviceState& d, uint32_t first, uint32_t count, float rayLen, float safe, hipStream_t s)
{
  if (!count || !d.aRayLen) return;
  hipLaunchKernelGGL(k_fill_sensors, dim3((count + kTile - 1) / kTile), dim3(kTile), 0, s, d, first, count, rayLen, safe);
}

void launchAgentFrontRays(const DeviceState& d, const TickParams& p, hipStream_t s)
{
  if (!p.n || !d.aLane || !d.aBrake) return;
  hipMemsetAsync(d.agentCount, 0, sizeof(uint32_t), s);
  hipLaunchKernelGGL(k_list_onrails_agents, dim3((p.n + kTile - 1) / kTile), dim3(kTile), 0, s, d, p.n);
  const uint32_t blocks = std::min((p.n + 3u) / 4u, 8192u);          // (a wave per agent, wave-strided over the list the kernel above wrote)
  hipLaunchKernelGGL(k_agent_front_rays, dim3(std::max(blocks, 1u)), dim3(kTile), 0, s, d, p);
}

void launchAgentRaySnapshot(const DeviceState& d, const TickParams& p, hipStream_t s)
{
  if (!p.n || !d.aLane || !d.aBrake || !d.agentRays) return;
  hipMemsetAsync(d.agentCount, 0, sizeof(uint32_t), s);
  hipLaunchKernelGGL(k_list_onrails_agents_with_rays, dim3((p.n + kTile - 1) / kTile), dim3(kTile), 0, s, d, p.n);
}
void launchAgentFrontRaysFromSnapshot(const DeviceState& d, const TickParams& p, hipStream_t s)
{
  if (!p.n || !d.aLane || !d.aBrake || !d.agentRays) return;
  const uint32_t blocks = std::min((p.n + 3u) / 4u, 8192u);
  hipLaunchKernelGGL(k_agent_front_rays_from_snapshot, dim3(std::max(blocks, 1u)), dim3(kTile), 0, s, d, p);
}

void launchRayQueries(const DeviceState& d, const TickParams& p, const RayQueryState& q, hipStream_t s)
{
  if (!q.count) return;
  const uint32_t perBlock = kTile / 64u;
  hipLaunchKernelGGL(k_ray_queries, dim3((q.count + perBlock - 1) / perBlock), dim3(kTile), 0, s, d, p, q);
}

} // namespace sctick
