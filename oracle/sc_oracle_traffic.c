/*
 * sc_oracle_traffic.c -- CPU oracle for the step BEFORE the tick path: the lane graph, the on-rails
 * branch of TrafficAISystem and TrafficLODSystem's tier selection.  TEST INFRASTRUCTURE ONLY (see sc_oracle.h).
 *
 * Plain-C restatement of
 *   src/engine/traffic/sc_traffic_lanes.cpp   :33-43 quantPos/quantDir, :65-91 addNode, :93-135 addSegment,
 *                                             :137-156 chooseNextSegment, :158-225 buildProceduralForSector,
 *                                             :291-352 advanceAlongLane, :392-400 laneSpeedLimit
 *   src/engine/traffic/sc_traffic_lanes.cpp   :240-279 queryNearestLane
 *   src/engine/traffic/sc_traffic_ai.cpp      :58-62 smoothExp, :72-75 yawFromDir, :264-299 the per-agent preamble
 *                                             (lane re-acquisition, lane validity, look-ahead point, the 1e-4 early-out,
 *                                             desired speed), :300-345 the obstacle ray and its brake, :434-460 the on-rails branch
 *   src/engine/traffic/sc_traffic_lod.cpp     :269-274 threshold repair, :303-307 distances, :323-353 hysteresis,
 *                                             :355-417 the physics / kinematic caps
 * PARITY UNPINNED: the reference holds no test or fixture for these, and the translation units need
 * sc_world_partition.h -> sc_assets.h -> <vulkan/vulkan.h>, so they do not build here without stand-ins.
 * Everything is compiled with -ffp-contract=off; libm calls are the float overloads the reference's std:: calls pick.
 */
#include "sc_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#define ORC_INVALID_LANE 0xFFFFFFFFu

typedef struct { float pos[3]; float dir[3]; float speedLimit; uint32_t* conn; uint32_t connLen, connCap; } OrcLaneNode;   /* sc_traffic_lanes.h:13-19 */
typedef struct { uint32_t startNode, endNode; float width; int32_t ownerX, ownerZ; float length; float dir[3]; uint8_t active; } OrcLaneSegment; /* :21-30 */
typedef struct { int32_t x, y, z; int16_t dx, dy, dz; uint32_t node; uint8_t used; } OrcNodeSlot;

struct OrcLaneGraph {
  OrcLaneNode* nodes; uint32_t nodeLen, nodeCap;
  OrcLaneSegment* segs; uint32_t segLen, segCap;
  OrcNodeSlot* table; uint32_t tableCap, tableUsed;       /* stands in for m_nodeLookup (an unordered_map keyed by LaneNodeKey) */
  float laneWidth, speedLimit;                             /* sc_traffic_lanes.h:92-93 */
};

static void* xr(void* p, size_t n) { void* q = realloc(p, n ? n : 1); if (!q) abort(); return q; }

OrcLaneGraph* orc_lanes_new(void)
{
  OrcLaneGraph* g = calloc(1, sizeof *g);
  if (!g) abort();
  g->laneWidth = 3.5f; g->speedLimit = 12.0f;
  return g;
}

void orc_lanes_free(OrcLaneGraph* g)
{
  if (!g) return;
  for (uint32_t i = 0; i < g->nodeLen; ++i) free(g->nodes[i].conn);
  free(g->nodes); free(g->segs); free(g->table); free(g);
}

static float length3(const float v[3]) { return sqrtf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]); }     /* :11-14 */
static float dot3(const float a[3], const float b[3]) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; } /* :28-31 */
static void normalize3(float v[3])                                                                       /* :16-26 */
{
  const float len = length3(v);
  if (len > 1e-6f) { const float inv = 1.0f / len; v[0] *= inv; v[1] *= inv; v[2] *= inv; }
}
static int32_t quantPos(float v) { const float s = v * 100.0f; return (int32_t)floorf(s + (s >= 0.0f ? 0.5f : -0.5f)); }   /* :33-37 */
static int16_t quantDir(float v) { const float s = v * 1000.0f; return (int16_t)floorf(s + (s >= 0.0f ? 0.5f : -0.5f)); }  /* :39-43 */

static uint64_t keyHash(const OrcNodeSlot* k)                                                            /* :46-55 */
{
  const uint64_t a = (uint32_t)k->x, ay = (uint32_t)k->y, b = (uint32_t)k->z;
  const uint64_t c = (uint16_t)k->dx, cy = (uint16_t)k->dy, d = (uint16_t)k->dz;
  return (a * 73856093ull) ^ (ay * 83492791ull) ^ (b * 19349663ull) ^ (c * 2654435761ull) ^ (cy * 97531ull) ^ (d * 4256249ull);
}
static int keyEq(const OrcNodeSlot* a, const OrcNodeSlot* b) { return a->x == b->x && a->y == b->y && a->z == b->z && a->dx == b->dx && a->dy == b->dy && a->dz == b->dz; }

static OrcNodeSlot* tableFind(OrcLaneGraph* g, const OrcNodeSlot* key)
{
  if (!g->tableCap) return NULL;
  uint32_t i = (uint32_t)(keyHash(key) % g->tableCap);
  while (g->table[i].used) { if (keyEq(&g->table[i], key)) return &g->table[i]; i = (i + 1u) % g->tableCap; }
  return NULL;
}
static void tableInsert(OrcLaneGraph* g, const OrcNodeSlot* key)
{
  if ((g->tableUsed + 1u) * 2u > g->tableCap) {
    OrcNodeSlot* old = g->table; const uint32_t oldCap = g->tableCap;
    g->tableCap = oldCap ? oldCap * 2u : 1024u;
    g->table = calloc(g->tableCap, sizeof *g->table);
    if (!g->table) abort();
    g->tableUsed = 0;
    for (uint32_t k = 0; k < oldCap; ++k) if (old[k].used) tableInsert(g, &old[k]);
    free(old);
  }
  uint32_t i = (uint32_t)(keyHash(key) % g->tableCap);
  while (g->table[i].used) i = (i + 1u) % g->tableCap;
  g->table[i] = *key; g->table[i].used = 1; g->tableUsed++;
}

static uint32_t addNode(OrcLaneGraph* g, const float pos[3], const float dir[3], float speedLimit)      /* :65-91 */
{
  OrcNodeSlot key; memset(&key, 0, sizeof key);
  key.x = quantPos(pos[0]); key.y = quantPos(pos[1]); key.z = quantPos(pos[2]);
  key.dx = quantDir(dir[0]); key.dy = quantDir(dir[1]); key.dz = quantDir(dir[2]);
  OrcNodeSlot* hit = tableFind(g, &key);
  if (hit) return hit->node;
  if (g->nodeLen == g->nodeCap) { g->nodeCap = g->nodeCap ? g->nodeCap * 2u : 64u; g->nodes = xr(g->nodes, (size_t)g->nodeCap * sizeof *g->nodes); }
  OrcLaneNode* n = &g->nodes[g->nodeLen];
  memset(n, 0, sizeof *n);
  memcpy(n->pos, pos, 12); memcpy(n->dir, dir, 12); n->speedLimit = speedLimit;
  key.node = g->nodeLen;
  tableInsert(g, &key);
  return g->nodeLen++;
}

static uint32_t addSegment(OrcLaneGraph* g, uint32_t startNode, uint32_t endNode, const float dir[3], int32_t ox, int32_t oz)  /* :93-135 */
{
  if (startNode >= g->nodeLen || endNode >= g->nodeLen) return ORC_INVALID_LANE;
  const OrcLaneNode* a = &g->nodes[startNode]; const OrcLaneNode* b = &g->nodes[endNode];
  float segDir[3] = { b->pos[0] - a->pos[0], b->pos[1] - a->pos[1], b->pos[2] - a->pos[2] };
  const float len = length3(segDir);
  if (len > 1e-6f) { const float inv = 1.0f / len; segDir[0] *= inv; segDir[1] *= inv; segDir[2] *= inv; }
  else { memcpy(segDir, dir, 12); normalize3(segDir); }
  if (g->segLen == g->segCap) { g->segCap = g->segCap ? g->segCap * 2u : 64u; g->segs = xr(g->segs, (size_t)g->segCap * sizeof *g->segs); }
  OrcLaneSegment* s = &g->segs[g->segLen];
  s->startNode = startNode; s->endNode = endNode; s->width = g->laneWidth; s->ownerX = ox; s->ownerZ = oz;
  s->length = len; memcpy(s->dir, segDir, 12); s->active = 1;
  OrcLaneNode* sn = &g->nodes[startNode];
  if (sn->connLen == sn->connCap) { sn->connCap = sn->connCap ? sn->connCap * 2u : 2u; sn->conn = xr(sn->conn, (size_t)sn->connCap * 4u); }
  sn->conn[sn->connLen++] = g->segLen;
  return g->segLen++;
}

/* buildProceduralForSector (:158-225) for a sector that has no lanes yet: two lanes per axis, lane width / 2 off the
 * centre lines; bounds = sectorBounds (sc_world_partition.cpp:277-287): min = coord * size, max = min + size */
void orc_lanes_build_sector(OrcLaneGraph* g, int32_t cx, int32_t cz, float sectorSize, uint32_t outSegs[4])
{
  const float minX = (float)cx * sectorSize, minZ = (float)cz * sectorSize;
  const float maxX = minX + sectorSize, maxZ = minZ + sectorSize;
  const float centerX = (minX + maxX) * 0.5f, centerZ = (minZ + maxZ) * 0.5f;
  const float y = 0.0f, offset = g->laneWidth * 0.5f;
  uint32_t k = 0;
  {
    const float dirPos[3] = { 1.0f, 0.0f, 0.0f }, dirNeg[3] = { -1.0f, 0.0f, 0.0f };
    float start[3] = { minX, y, centerZ - offset }, end[3] = { maxX, y, centerZ - offset };
    uint32_t n0 = addNode(g, start, dirPos, g->speedLimit), n1 = addNode(g, end, dirPos, g->speedLimit);
    outSegs[k++] = addSegment(g, n0, n1, dirPos, cx, cz);
    start[0] = maxX; start[2] = centerZ + offset; end[0] = minX; end[2] = centerZ + offset;
    n0 = addNode(g, start, dirNeg, g->speedLimit); n1 = addNode(g, end, dirNeg, g->speedLimit);
    outSegs[k++] = addSegment(g, n0, n1, dirNeg, cx, cz);
  }
  {
    const float dirPos[3] = { 0.0f, 0.0f, 1.0f }, dirNeg[3] = { 0.0f, 0.0f, -1.0f };
    float start[3] = { centerX + offset, y, minZ }, end[3] = { centerX + offset, y, maxZ };
    uint32_t n0 = addNode(g, start, dirPos, g->speedLimit), n1 = addNode(g, end, dirPos, g->speedLimit);
    outSegs[k++] = addSegment(g, n0, n1, dirPos, cx, cz);
    start[0] = centerX - offset; start[2] = maxZ; end[0] = centerX - offset; end[2] = minZ;
    n0 = addNode(g, start, dirNeg, g->speedLimit); n1 = addNode(g, end, dirNeg, g->speedLimit);
    outSegs[k++] = addSegment(g, n0, n1, dirNeg, cx, cz);
  }
}

void orc_lanes_set_active(OrcLaneGraph* g, uint32_t seg, int active) { if (seg < g->segLen) g->segs[seg].active = active ? 1 : 0; }   /* removeSector, :227-237 */
uint32_t orc_lanes_segment_count(const OrcLaneGraph* g) { return g->segLen; }
uint32_t orc_lanes_node_count(const OrcLaneGraph* g) { return g->nodeLen; }

/* flat views for tests (what a host hands to scTickSetLaneGraph) */
void orc_lanes_export(const OrcLaneGraph* g, float* segStart3, float* segDir3, float* segLength, uint8_t* segActive,
                      uint32_t* segEndNode, float* segSpeedLimit, float* nodePos3, uint32_t* nodeConnOffset, uint32_t* nodeConn)
{
  for (uint32_t i = 0; i < g->segLen; ++i) {
    const OrcLaneSegment* s = &g->segs[i];
    memcpy(segStart3 + 3u * i, g->nodes[s->startNode].pos, 12); memcpy(segDir3 + 3u * i, s->dir, 12);
    segLength[i] = s->length; segActive[i] = s->active; segEndNode[i] = s->endNode;
    segSpeedLimit[i] = g->nodes[s->startNode].speedLimit;                                                /* laneSpeedLimit, :392-400 */
  }
  uint32_t at = 0;
  for (uint32_t n = 0; n < g->nodeLen; ++n) {
    memcpy(nodePos3 + 3u * n, g->nodes[n].pos, 12);
    nodeConnOffset[n] = at;
    for (uint32_t k = 0; k < g->nodes[n].connLen; ++k) nodeConn[at++] = g->nodes[n].conn[k];
  }
  nodeConnOffset[g->nodeLen] = at;
}
uint32_t orc_lanes_connection_count(const OrcLaneGraph* g)
{
  uint32_t c = 0;
  for (uint32_t n = 0; n < g->nodeLen; ++n) c += g->nodes[n].connLen;
  return c;
}

static uint32_t chooseNextSegment(const OrcLaneGraph* g, const float dir[3], const OrcLaneNode* node)   /* :137-156 */
{
  uint32_t best = ORC_INVALID_LANE; float bestDot = -1.0f;
  for (uint32_t k = 0; k < node->connLen; ++k) {
    const uint32_t segId = node->conn[k];
    if (segId >= g->segLen) continue;
    const OrcLaneSegment* seg = &g->segs[segId];
    if (!seg->active) continue;
    const float d = dot3(dir, seg->dir);
    if (d > bestDot) { bestDot = d; best = segId; }
  }
  return best;
}

int orc_lanes_advance(const OrcLaneGraph* g, uint32_t* laneId, float* s, float distance, float outPos[3], float outDir[3])   /* :291-352 */
{
  if (*laneId == ORC_INVALID_LANE || *laneId >= g->segLen) return 0;
  float remaining = distance; uint32_t current = *laneId; float currentS = *s;
  for (uint32_t guard = 0; guard < 8; ++guard) {
    const OrcLaneSegment* seg = &g->segs[current];
    if (!seg->active) return 0;
    const float len = seg->length;
    if (len <= 1e-5f) return 0;
    const float available = len - currentS;
    if (remaining <= available) {
      currentS += remaining;
      const OrcLaneNode* a = &g->nodes[seg->startNode];
      outPos[0] = a->pos[0] + seg->dir[0] * currentS; outPos[1] = a->pos[1] + seg->dir[1] * currentS; outPos[2] = a->pos[2] + seg->dir[2] * currentS;
      memcpy(outDir, seg->dir, 12);
      *laneId = current; *s = currentS;
      return 1;
    }
    remaining -= available; currentS = 0.0f;
    const OrcLaneNode* endNode = &g->nodes[seg->endNode];
    const uint32_t next = chooseNextSegment(g, seg->dir, endNode);
    if (next == ORC_INVALID_LANE) {
      memcpy(outPos, endNode->pos, 12); memcpy(outDir, seg->dir, 12);
      *laneId = current; *s = len;
      return 1;
    }
    current = next;
  }
  return 0;
}

static float smoothExp(float current, float target, float response, float dt)      /* sc_traffic_ai.cpp:58-62 */
{
  const float t = 1.0f - expf(-response * dt);
  return current + (target - current) * t;
}

/* TrafficLaneGraph::queryNearestLane, sc_traffic_lanes.cpp:240-279: the closest point of every active segment, the first
 * segment wins among equals (strict <). */
int orc_lanes_query_nearest(const OrcLaneGraph* g, const float pos[3], uint32_t* laneOut, float* sOut)
{
  uint32_t best = ORC_INVALID_LANE; float bestS = 0.0f, bestDist = 0.0f; int hasBest = 0;
  for (uint32_t i = 0; i < g->segLen; ++i) {
    const OrcLaneSegment* seg = &g->segs[i];
    if (!seg->active || seg->length <= 1e-5f) continue;
    const OrcLaneNode* a = &g->nodes[seg->startNode];
    const float toP[3] = { pos[0] - a->pos[0], pos[1] - a->pos[1], pos[2] - a->pos[2] };
    const float proj = toP[0] * seg->dir[0] + toP[1] * seg->dir[1] + toP[2] * seg->dir[2];
    const float mn = (proj < seg->length) ? proj : seg->length;            /* std::min(seg.length, proj) */
    const float sv = (0.0f < mn) ? mn : 0.0f;                               /* std::max(0.0f, ...) */
    const float closest[3] = { a->pos[0] + seg->dir[0] * sv, a->pos[1] + seg->dir[1] * sv, a->pos[2] + seg->dir[2] * sv };
    const float dx = pos[0] - closest[0], dy = pos[1] - closest[1], dz = pos[2] - closest[2];
    const float distSq = dx * dx + dy * dy + dz * dz;
    if (!hasBest || distSq < bestDist) { hasBest = 1; bestDist = distSq; best = i; bestS = sv; }
  }
  *laneOut = best; *sOut = bestS;
  return best != ORC_INVALID_LANE;
}

static float clampf(float v, float lo, float hi)          /* std::max(lo, std::min(v, hi)), sc_traffic_ai.cpp:16-19 */
{
  const float m = (hi < v) ? hi : v;
  return (lo < m) ? m : lo;
}

/* The obstacle ray of one agent, sc_traffic_ai.cpp:300-345, against the world AABBs of the broadphase (own spec, as the ray
 * queries: Bullet is absent).  forward = normalize(sin(yaw), 0, cos(yaw)) with the yaw's sin / cos as the Transform holds them
 * (host libm), origin 1.7 m ahead and 0.6 m up, PhysicsWorld::raycast(origin, forward, rayLen, 1u).  The agent's own box
 * never answers (Bullet does not report a convex shape the ray starts inside, and the reference ignores a self hit anyway,
 * :322-325, :336).  brake = clamp01((safe - d) / safe) for a hit closer than `safe` (:336-339). */
/* the ray itself: 1 = something was hit; *tOut the entry distance, *idxOut the box (lowest index among equal distances) */
static int front_ray_cast(uint32_t n, const float* min3, const float* max3, const uint32_t* group, const uint32_t* mask,
                          uint32_t self, const float pos[3], float sinYaw, float cosYaw, float rayLen, float* tOut, uint32_t* idxOut)
{
  float forward[3] = { sinYaw, 0.0f, cosYaw };
  normalize3(forward);
  const float origin[3] = { pos[0] + forward[0] * 1.7f, pos[1] + 0.6f, pos[2] + forward[2] * 1.7f };
  const float lenSq = forward[0] * forward[0] + forward[1] * forward[1] + forward[2] * forward[2];
  if (!(lenSq > 1e-6f) || !(rayLen >= 0.0f)) return 0;
  const float invLen = 1.0f / sqrtf(lenSq);
  const float dir[3] = { forward[0] * invLen, forward[1] * invLen, forward[2] * invLen };
  float best = INFINITY; int hit = 0; uint32_t who = 0;
  for (uint32_t i = 0; i < n; ++i) {
    if (i == self) continue;
    if (!(min3[3 * i] <= max3[3 * i])) continue;
    const uint32_t gq = group[i] & 0xFFFFu, mq = mask[i] & 0xFFFFu;
    if (!(gq & 1u) || !mq) continue;
    float tmin = 0.0f, tmax = rayLen; int ok = 1;
    for (int k = 0; k < 3 && ok; ++k) {                                     /* intersectRayAABB, editor_core.cpp:438-470 */
      if (fabsf(dir[k]) < 1e-6f) { if (origin[k] < min3[3 * i + k] || origin[k] > max3[3 * i + k]) ok = 0; }
      else {
        const float ood = 1.0f / dir[k];
        float t1 = (min3[3 * i + k] - origin[k]) * ood, t2 = (max3[3 * i + k] - origin[k]) * ood;
        if (t1 > t2) { const float q = t1; t1 = t2; t2 = q; }
        if (t1 > tmin) tmin = t1;
        tmax = tmax < t2 ? tmax : t2;
        if (tmin > tmax) ok = 0;
      }
    }
    if (ok && tmin < best) { best = tmin; hit = 1; who = i; }
  }
  *tOut = best; *idxOut = who;
  return hit;
}

float orc_traffic_front_ray_brake(uint32_t n, const float* min3, const float* max3, const uint32_t* group, const uint32_t* mask,
                                  uint32_t self, const float pos[3], float sinYaw, float cosYaw, float rayLen, float safe)
{
  float t; uint32_t who;
  if (front_ray_cast(n, min3, max3, group, mask, self, pos, sinYaw, cosYaw, rayLen, &t, &who) && safe > 1e-3f && t < safe)
    return clampf((safe - t) / safe, 0.0f, 1.0f);
  return 0.0f;
}

/* Per-agent TrafficSensors (sc_traffic_common.h:46-53) and what the AI leaves in them (sc_traffic_ai.cpp:306-308, :317-345): every
 * OnRails agent casts with ITS frontRayLength and brakes by ITS safeDistance (rayLen / safe per entity; NULL = the defaults 20 / 10
 * of :307-308), and lastHitDistance / lastHitType are written back: the hit's distance and Vehicle (2) or World (3), or the ray's
 * length and None (0) without a hit.  "The hit entity carries a VehicleComponent or VehicleRuntime" (:327) is isVehicle[] here
 * (traffic agents and SynthWorld's vehicle movers); Self (1) cannot occur -- an agent's own box never answers (own spec, as the
 * rays themselves: Bullet is absent). */
void orc_traffic_front_ray_sensors(OrcWorld* w, const float* min3, const float* max3, const uint32_t* group, const uint32_t* mask,
                                   const uint8_t* isAgent, const uint8_t* mode, const uint8_t* isVehicle, const float* rayLen, const float* safe,
                                   float* brakeOut, float* hitDistOut, uint8_t* hitTypeOut)
{
  const OrcTransform* d = orc_transform_dense_data(w);
  const uint32_t n = orc_transform_count(w);
  for (uint32_t i = 0; i < n; ++i) {
    brakeOut[i] = 0.0f; hitDistOut[i] = 0.0f; hitTypeOut[i] = 0u;
    if (!isAgent[i] || mode[i] != 2u) continue;
    const float len = rayLen ? rayLen[i] : 20.0f, sf = safe ? safe[i] : 10.0f;
    const float sy = sinf(d[i].localRot[1]), cy = cosf(d[i].localRot[1]);
    float t; uint32_t who;
    hitDistOut[i] = len;
    if (front_ray_cast(n, min3, max3, group, mask, i, d[i].localPos, sy, cy, len, &t, &who)) {
      hitDistOut[i] = t;
      hitTypeOut[i] = isVehicle[who] ? 2u : 3u;
      if (sf > 1e-3f && t < sf) brakeOut[i] = clampf((sf - t) / sf, 0.0f, 1.0f);
    }
  }
}

/* Every OnRails agent's brake for one step (the rays of sc_traffic_ai.cpp:300-345 against the world as it stands): sin / cos of
 * Transform::localRot[1] with the float libm, as the reference's std::sin(currentYaw) / std::cos(currentYaw).  A box whose
 * AABB misses the ray segment's own bounding box (grown by a centimetre) cannot be hit and is skipped before the slab test. */
void orc_traffic_front_ray_brakes(OrcWorld* w, const float* min3, const float* max3, const uint32_t* group, const uint32_t* mask,
                                  const uint8_t* isAgent, const uint8_t* mode, float rayLen, float safe, float* brakeOut)
{
  const OrcTransform* d = orc_transform_dense_data(w);
  const uint32_t n = orc_transform_count(w);
  for (uint32_t i = 0; i < n; ++i) {
    brakeOut[i] = 0.0f;
    if (!isAgent[i] || mode[i] != 2u) continue;
    const float sy = sinf(d[i].localRot[1]), cy = cosf(d[i].localRot[1]);
    float f[3] = { sy, 0.0f, cy };
    normalize3(f);
    const float o[3] = { d[i].localPos[0] + f[0] * 1.7f, d[i].localPos[1] + 0.6f, d[i].localPos[2] + f[2] * 1.7f };
    float lo[3], hi[3];
    for (int k = 0; k < 3; ++k) { const float e = o[k] + f[k] * (rayLen * 1.001f); lo[k] = (o[k] < e ? o[k] : e) - 0.01f; hi[k] = (o[k] < e ? e : o[k]) + 0.01f; }
    /* gather the candidates, then the exact routine on them (indices kept: `self` must stay recognisable) */
    float best = 0.0f;
    {
      float brake = 0.0f;
      /* the exact test on the boxes that can be hit at all */
      static uint32_t* cand = NULL; static uint32_t candCap = 0;
      uint32_t m = 0;
      for (uint32_t j = 0; j < n; ++j) {
        if (max3[3 * j] < lo[0] || min3[3 * j] > hi[0] || max3[3 * j + 2] < lo[2] || min3[3 * j + 2] > hi[2] || max3[3 * j + 1] < lo[1] || min3[3 * j + 1] > hi[1]) continue;
        if (m == candCap) { candCap = candCap ? candCap * 2u : 256u; cand = xr(cand, (size_t)candCap * 4u); }
        cand[m++] = j;
      }
      if (m) {
        float* cmn = xr(NULL, (size_t)m * 12u); float* cmx = xr(NULL, (size_t)m * 12u); uint32_t* cg = xr(NULL, (size_t)m * 4u); uint32_t* cm = xr(NULL, (size_t)m * 4u);
        uint32_t self = 0xFFFFFFFFu;
        for (uint32_t k = 0; k < m; ++k) {
          const uint32_t j = cand[k];
          memcpy(cmn + 3 * k, min3 + 3 * j, 12); memcpy(cmx + 3 * k, max3 + 3 * j, 12); cg[k] = group[j]; cm[k] = mask[j];
          if (j == i) self = k;
        }
        brake = orc_traffic_front_ray_brake(m, cmn, cmx, cg, cm, self, d[i].localPos, sy, cy, rayLen, safe);
        free(cmn); free(cmx); free(cg); free(cm);
      }
      best = brake;
    }
    brakeOut[i] = best;
  }
}

/* TrafficAISystem for the agents of the OnRails tier (mode 2): sc_traffic_ai.cpp:264-299 preamble + :434-460.  Every agent
 * without a lane first takes the nearest active one (:264-272, whatever its tier).  obstacleBrake[i] (may be NULL: no
 * PhysicsWorld, brake 0) is the agent's brake from its front ray, :300-345 -- cast by the caller against the world as it
 * stood BEFORE this step, which is what Bullet's world is while TrafficAISystem walks the agents.  Arrays are in
 * Transform-pool dense order; isAgent[i] marks the entities that carry TrafficAgent + TrafficVehicle.  Physics / Kinematic
 * agents are not moved here (their transforms come from the physics sync, :351-433 / TrafficPhysicsSyncSystem). */
void orc_traffic_ai_onrails_braked(OrcWorld* w, const OrcLaneGraph* g, const uint8_t* isAgent, uint32_t* laneId, float* laneS,
                                   float* targetSpeed, const uint8_t* mode, const float* lookAheadDist, const float* obstacleBrakeIn,
                                   float speedMultiplier, float dt)
{
  OrcTransform* d = orc_transform_dense_data(w);
  const uint32_t n = orc_transform_count(w);
  for (uint32_t i = 0; i < n; ++i) {
    if (!isAgent[i]) continue;
    OrcTransform* tr = &d[i];
    if (laneId[i] == ORC_INVALID_LANE) {                                           /* :264-272 */
      uint32_t q; float qs;
      if (orc_lanes_query_nearest(g, tr->localPos, &q, &qs)) { laneId[i] = q; laneS[i] = qs; }
    }
    if (mode[i] != 2u) continue;
    if (laneId[i] == ORC_INVALID_LANE || laneId[i] >= g->segLen) continue;         /* getLane() == nullptr, :274-276 */
    if (!g->segs[laneId[i]].active) continue;
    float target[3] = { 0, 0, 0 }, tmpDir[3];
    { uint32_t id = laneId[i]; float ss = laneS[i];                                /* getLookAheadPoint, sc_traffic_lanes.cpp:281-289 */
      if (!orc_lanes_advance(g, &id, &ss, lookAheadDist[i], target, tmpDir)) continue; }
    float toTarget[3] = { target[0] - tr->localPos[0], 0.0f, target[2] - tr->localPos[2] };
    if (length3(toTarget) < 1e-4f) continue;                                       /* :283-284 */
    float desiredSpeed = g->nodes[g->segs[laneId[i]].startNode].speedLimit;        /* laneSpeedLimit, :296 */
    desiredSpeed *= speedMultiplier;                                               /* :297-298 (dbg->speedMultiplier, 1 by default) */
    desiredSpeed = (0.0f < desiredSpeed) ? desiredSpeed : 0.0f;                    /* std::max(0.0f, desiredSpeed), :299 */
    const float obstacleBrake = obstacleBrakeIn ? obstacleBrakeIn[i] : 0.0f;       /* :300-345 */
    const float desired = desiredSpeed * (1.0f - obstacleBrake);                   /* :436 */
    targetSpeed[i] = smoothExp(targetSpeed[i], desired, 2.5f, dt);                 /* :437 */
    const float travel = targetSpeed[i] * dt;                                      /* :439 */
    uint32_t id = laneId[i]; float ss = laneS[i]; float pos[3] = { 0, 0, 0 }, dir[3] = { 0, 0, 0 };
    if (orc_lanes_advance(g, &id, &ss, travel, pos, dir)) {                        /* :445-459 */
      pos[1] = tr->localPos[1];
      laneId[i] = id; laneS[i] = ss;
      tr->localPos[0] = pos[0]; tr->localPos[1] = pos[1]; tr->localPos[2] = pos[2];
      tr->localRot[0] = 0.0f; tr->localRot[1] = atan2f(dir[0], dir[2]); tr->localRot[2] = 0.0f;   /* yawFromDir, :72-75 */
      tr->dirty = 1;
    }
  }
}

void orc_traffic_ai_onrails(OrcWorld* w, const OrcLaneGraph* g, const uint8_t* isAgent, uint32_t* laneId, float* laneS,
                            float* targetSpeed, const uint8_t* mode, const float* lookAheadDist, float speedMultiplier, float dt)
{
  orc_traffic_ai_onrails_braked(w, g, isAgent, laneId, laneS, targetSpeed, mode, lookAheadDist, NULL, speedMultiplier, dt);
}

/* TrafficLODSystem's tier selection (sc_traffic_lod.cpp:269-274, :303-307, :323-417): desired tier per vehicle from its
 * xz distance to the player with hysteresis, then the physics / kinematic caps.  The caps sort by distance, descending,
 * with std::sort (order of equal distances unspecified); here equal distances keep their index order (stable), which is
 * one of the orders std::sort may produce.  The total cap / despawn (:419-465) is streaming and is not restated. */
void orc_traffic_lod_tiers(OrcWorld* w, const uint8_t* isAgent, const uint8_t* mode, const float playerPos[3],
                           float tierAEnter, float tierAExit, float tierBEnter, float tierBExit,
                           uint32_t maxPhysics, uint32_t maxKinematic, uint8_t* desiredOut, uint32_t counts[3])
{
  if (tierAExit < tierAEnter + 1.0f) tierAExit = tierAEnter + 1.0f;                /* :269-274 */
  if (tierBEnter < tierAExit + 1.0f) tierBEnter = tierAExit + 1.0f;
  if (tierBExit < tierBEnter + 1.0f) tierBExit = tierBEnter + 1.0f;
  const OrcTransform* d = orc_transform_dense_data(w);
  const uint32_t n = orc_transform_count(w);
  uint32_t m = 0;
  for (uint32_t i = 0; i < n; ++i) if (isAgent[i]) m++;
  uint32_t* idx = xr(NULL, (size_t)m * 4u); float* dist = xr(NULL, (size_t)m * 4u); uint8_t* des = xr(NULL, m);
  m = 0;
  for (uint32_t i = 0; i < n; ++i) {
    if (!isAgent[i]) continue;
    const float dx = d[i].localPos[0] - playerPos[0], dz = d[i].localPos[2] - playerPos[2];
    idx[m] = i; dist[m] = sqrtf(dx * dx + dz * dz);                                /* :303-307, distanceSq2d */
    const uint8_t cur = mode[i]; const float dd = dist[m]; uint8_t want;
    if (cur == 0u) want = (dd > tierAExit) ? ((dd < tierBEnter) ? 1u : 2u) : 0u;   /* :328-334 */
    else if (cur == 1u) want = (dd < tierAEnter) ? 0u : ((dd > tierBExit) ? 2u : 1u);   /* :335-343 */
    else want = (dd < tierAEnter) ? 0u : ((dd < tierBEnter) ? 1u : 2u);            /* :344-352 */
    des[m++] = want;
  }
  uint32_t physicsCount = 0, kinematicCount = 0, onRailsCount = 0;
  for (uint32_t k = 0; k < m; ++k) { if (des[k] == 0u) physicsCount++; else if (des[k] == 1u) kinematicCount++; else onRailsCount++; }
  uint32_t* sel = xr(NULL, (size_t)m * 4u);
  if (maxPhysics > 0 && physicsCount > maxPhysics) {                              /* :373-399 */
    uint32_t c = 0;
    for (uint32_t k = 0; k < m; ++k) if (des[k] == 0u) sel[c++] = k;
    for (uint32_t a = 1; a < c; ++a) {                                            /* stable insertion sort, distance descending */
      const uint32_t v = sel[a]; uint32_t b = a;
      while (b > 0 && dist[sel[b - 1]] < dist[v]) { sel[b] = sel[b - 1]; --b; }
      sel[b] = v;
    }
    for (uint32_t a = maxPhysics; a < c; ++a) {
      if (maxKinematic == 0 || kinematicCount < maxKinematic) { des[sel[a]] = 1u; kinematicCount++; }
      else { des[sel[a]] = 2u; onRailsCount++; }
      physicsCount--;
    }
  }
  if (maxKinematic > 0 && kinematicCount > maxKinematic) {                        /* :401-417 */
    uint32_t c = 0;
    for (uint32_t k = 0; k < m; ++k) if (des[k] == 1u) sel[c++] = k;
    for (uint32_t a = 1; a < c; ++a) {
      const uint32_t v = sel[a]; uint32_t b = a;
      while (b > 0 && dist[sel[b - 1]] < dist[v]) { sel[b] = sel[b - 1]; --b; }
      sel[b] = v;
    }
    for (uint32_t a = maxKinematic; a < c; ++a) { des[sel[a]] = 2u; kinematicCount--; onRailsCount++; }
  }
  for (uint32_t k = 0; k < m; ++k) desiredOut[idx[k]] = des[k];
  counts[0] = physicsCount; counts[1] = kinematicCount; counts[2] = onRailsCount;
  free(idx); free(dist); free(des); free(sel);
}

/* The total cap of TrafficLODSystem, sc_traffic_lod.cpp:419-465: with more vehicles than maxTotal the surplus is flagged for
 * despawning -- the OnRails bucket sorted by distance descending first, then the Kinematic one, then the Physics one (std::sort
 * there: order of equal distances unspecified; here they keep their pool order, one of the orders it may produce).  `mode` is
 * the tier of every vehicle after the caps (what `desired` holds at that point).  Returns how many are flagged; outIdx receives
 * their dense indices in flagging order. */
uint32_t orc_traffic_lod_despawns(OrcWorld* w, const uint8_t* isAgent, const uint8_t* mode, const float playerPos[3], uint32_t maxTotal, uint32_t* outIdx)
{
  const OrcTransform* d = orc_transform_dense_data(w);
  const uint32_t n = orc_transform_count(w);
  uint32_t m = 0;
  for (uint32_t i = 0; i < n; ++i) if (isAgent[i]) m++;
  if (maxTotal == 0 || m <= maxTotal) return 0;
  uint32_t toRemove = m - maxTotal, flagged = 0;
  uint32_t* idx = xr(NULL, (size_t)m * 4u); float* dist = xr(NULL, (size_t)m * 4u); uint32_t* sel = xr(NULL, (size_t)m * 4u);
  m = 0;
  for (uint32_t i = 0; i < n; ++i) {
    if (!isAgent[i]) continue;
    const float dx = d[i].localPos[0] - playerPos[0], dz = d[i].localPos[2] - playerPos[2];
    idx[m] = i; dist[m++] = sqrtf(dx * dx + dz * dz);
  }
  const uint8_t order[3] = { 2u, 1u, 0u };                                      /* OnRails, Kinematic, Physics */
  for (int b = 0; b < 3 && toRemove > 0; ++b) {
    uint32_t c = 0;
    for (uint32_t k = 0; k < m; ++k) if (mode[idx[k]] == order[b]) sel[c++] = k;
    /* stable merge sort by distance, descending (buckets hold up to every vehicle: no quadratic insertion here) */
    uint32_t* tmp = xr(NULL, (size_t)(c ? c : 1) * 4u);
    for (uint32_t width = 1; width < c; width *= 2u) {
      for (uint32_t lo = 0; lo < c; lo += 2u * width) {
        const uint32_t mid = lo + width < c ? lo + width : c, hi = lo + 2u * width < c ? lo + 2u * width : c;
        uint32_t a = lo, bb = mid, o = lo;
        while (a < mid && bb < hi) tmp[o++] = (dist[sel[bb]] > dist[sel[a]]) ? sel[bb++] : sel[a++];
        while (a < mid) tmp[o++] = sel[a++];
        while (bb < hi) tmp[o++] = sel[bb++];
      }
      memcpy(sel, tmp, (size_t)c * 4u);
    }
    free(tmp);
    for (uint32_t k = 0; k < c && toRemove > 0; ++k) { outIdx[flagged++] = idx[sel[k]]; toRemove--; }
  }
  free(idx); free(dist); free(sel);
  return flagged;
}
