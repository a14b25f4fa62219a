/*
 * sc_oracle.c -- CPU oracle for the world-tick path.  TEST INFRASTRUCTURE ONLY
 * (see sc_oracle.h for the pin status of every function).
 *
 * Build: gcc -std=c11 -O2 -ffp-contract=off -fno-fast-math -fPIC -shared -pthread
 * (-ffp-contract=off matters: the reference is /fp:precise, nothing may fuse into FMA).
 *
 * Everything here is written from the reference's behaviour, not its text; each block cites
 * the reference file:line (relative to the reference tree) it follows.
 */
#define _GNU_SOURCE
#include "sc_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdatomic.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

/* ------------------------------------------------------------------------------------------
 * Math -- src/core/src/sc_math.cpp.  Column-major: element(row r, col c) = m[c*4 + r].
 * ---------------------------------------------------------------------------------------- */

void orc_mat4_identity(float out[16])
{
  for (int i = 0; i < 16; ++i) out[i] = 0.0f;
  out[0] = out[5] = out[10] = out[15] = 1.0f;
}

/* sc_math.cpp:52-68 (the SSE variant, which is what x64 builds take): every output element is
 * ((a[0*4+r]*b[c*4+0] + a[1*4+r]*b[c*4+1]) + a[2*4+r]*b[c*4+2]) + a[3*4+r]*b[c*4+3], unfused,
 * starting from the first product (not from 0.0f as the scalar fallback at :70-83 does). */
void orc_mat4_mul(const float a[16], const float b[16], float out[16])
{
  float r[16];
  for (int c = 0; c < 4; ++c) {
    const float* bc = b + c * 4;
    for (int row = 0; row < 4; ++row) {
      float acc = a[row] * bc[0];
      acc = acc + a[4 + row] * bc[1];
      acc = acc + a[8 + row] * bc[2];
      acc = acc + a[12 + row] * bc[3];
      r[c * 4 + row] = acc;
    }
  }
  memcpy(out, r, sizeof r);
}

/* sc_math.cpp:100-128: three axis matrices from host libm cos/sin, combined as (Rz*Ry)*Rx. */
void orc_mat4_rotation_xyz(float rx, float ry, float rz, float out[16])
{
  const float cx = cosf(rx), sx = sinf(rx);
  const float cy = cosf(ry), sy = sinf(ry);
  const float cz = cosf(rz), sz = sinf(rz);

  float mx[16], my[16], mz[16], zy[16];
  orc_mat4_identity(mx);
  mx[5] = cx;  mx[6] = sx;  mx[9] = -sx; mx[10] = cx;
  orc_mat4_identity(my);
  my[0] = cy;  my[2] = -sy; my[8] = sy;  my[10] = cy;
  orc_mat4_identity(mz);
  mz[0] = cz;  mz[1] = sz;  mz[4] = -sz; mz[5] = cz;

  orc_mat4_mul(mz, my, zy);
  orc_mat4_mul(zy, mx, out);
}

/* sc_math.cpp:130-142: local = T * (R * S), each a full 4x4 product. */
void orc_mat4_trs(const float pos[3], const float rot[3], const float scale[3], float out[16])
{
  if (!pos || !rot || !scale) { orc_mat4_identity(out); return; }
  float t[16], r[16], s[16], rs[16];
  orc_mat4_identity(t);
  t[12] = pos[0]; t[13] = pos[1]; t[14] = pos[2];
  orc_mat4_rotation_xyz(rot[0], rot[1], rot[2], r);
  memset(s, 0, sizeof s);
  s[0] = scale[0]; s[5] = scale[1]; s[10] = scale[2]; s[15] = 1.0f;
  orc_mat4_mul(r, s, rs);
  orc_mat4_mul(t, rs, out);
}

/* sc_math.cpp:144-207: cofactor inverse.  Each adjugate entry is a left-to-right sum of six
 * signed triple products (m[i]*m[j])*m[k]; the table below lists, per output slot, the six
 * (sign,i,j,k) terms in the order the reference adds them (order fixes the fp32 rounding). */
typedef struct { int8_t s; uint8_t i, j, k; } OrcCofTerm;
static const struct { uint8_t slot; OrcCofTerm t[6]; } kCofactors[16] = {
  { 0, {{+1,5,10,15},{-1,5,11,14},{-1,9,6,15},{+1,9,7,14},{+1,13,6,11},{-1,13,7,10}}},
  { 4, {{-1,4,10,15},{+1,4,11,14},{+1,8,6,15},{-1,8,7,14},{-1,12,6,11},{+1,12,7,10}}},
  { 8, {{+1,4,9,15},{-1,4,11,13},{-1,8,5,15},{+1,8,7,13},{+1,12,5,11},{-1,12,7,9}}},
  {12, {{-1,4,9,14},{+1,4,10,13},{+1,8,5,14},{-1,8,6,13},{-1,12,5,10},{+1,12,6,9}}},
  { 1, {{-1,1,10,15},{+1,1,11,14},{+1,9,2,15},{-1,9,3,14},{-1,13,2,11},{+1,13,3,10}}},
  { 5, {{+1,0,10,15},{-1,0,11,14},{-1,8,2,15},{+1,8,3,14},{+1,12,2,11},{-1,12,3,10}}},
  { 9, {{-1,0,9,15},{+1,0,11,13},{+1,8,1,15},{-1,8,3,13},{-1,12,1,11},{+1,12,3,9}}},
  {13, {{+1,0,9,14},{-1,0,10,13},{-1,8,1,14},{+1,8,2,13},{+1,12,1,10},{-1,12,2,9}}},
  { 2, {{+1,1,6,15},{-1,1,7,14},{-1,5,2,15},{+1,5,3,14},{+1,13,2,7},{-1,13,3,6}}},
  { 6, {{-1,0,6,15},{+1,0,7,14},{+1,4,2,15},{-1,4,3,14},{-1,12,2,7},{+1,12,3,6}}},
  {10, {{+1,0,5,15},{-1,0,7,13},{-1,4,1,15},{+1,4,3,13},{+1,12,1,7},{-1,12,3,5}}},
  {14, {{-1,0,5,14},{+1,0,6,13},{+1,4,1,14},{-1,4,2,13},{-1,12,1,6},{+1,12,2,5}}},
  { 3, {{-1,1,6,11},{+1,1,7,10},{+1,5,2,11},{-1,5,3,10},{-1,9,2,7},{+1,9,3,6}}},
  { 7, {{+1,0,6,11},{-1,0,7,10},{-1,4,2,11},{+1,4,3,10},{+1,8,2,7},{-1,8,3,6}}},
  {11, {{-1,0,5,11},{+1,0,7,9},{+1,4,1,11},{-1,4,3,9},{-1,8,1,7},{+1,8,3,5}}},
  {15, {{+1,0,5,10},{-1,0,6,9},{-1,4,1,10},{+1,4,2,9},{+1,8,1,6},{-1,8,2,5}}},
};

void orc_mat4_inverse(const float a[16], float out[16])
{
  float o[16];
  for (int e = 0; e < 16; ++e) {
    float acc = 0.0f;
    for (int q = 0; q < 6; ++q) {
      const OrcCofTerm t = kCofactors[e].t[q];
      /* a leading "-m[i]*m[j]*m[k]" is ((-m[i])*m[j])*m[k] == -((m[i]*m[j])*m[k]) exactly, and
       * "x - y" == "x + (-y)" exactly, so applying the sign to the product is bit-identical. */
      float p = (a[t.i] * a[t.j]) * a[t.k];
      if (t.s < 0) p = -p;
      acc = (q == 0) ? p : acc + p;
    }
    o[kCofactors[e].slot] = acc;
  }
  float det = a[0] * o[0];
  det = det + a[1] * o[4];
  det = det + a[2] * o[8];
  det = det + a[3] * o[12];
  if (fabsf(det) <= 1e-6f) { orc_mat4_identity(out); return; }   /* sc_math.h:6 EPSILON, :199 */
  const float inv_det = 1.0f / det;
  for (int i = 0; i < 16; ++i) out[i] = o[i] * inv_det;
}

/* sc_math.cpp:209-232 */
void orc_mat4_perspective_rh_zo(float fovYRadians, float aspect, float zNear, float zFar,
                                int flipY, float out[16])
{
  const float eps = 1e-6f;
  if (fovYRadians <= eps || aspect <= eps || zNear <= eps || zFar <= zNear + eps) {
    orc_mat4_identity(out);
    return;
  }
  memset(out, 0, 16 * sizeof(float));
  const float f = 1.0f / tanf(fovYRadians * 0.5f);
  out[0] = f / aspect;
  out[5] = flipY ? -f : f;
  out[10] = zFar / (zNear - zFar);
  out[14] = (zFar * zNear) / (zNear - zFar);
  out[11] = -1.0f;
}

/* ------------------------------------------------------------------------------------------
 * ECS storage -- sc_ecs.h:14-34 (Entity), :42-58 + sc_ecs.cpp:11-56 (EntityManager),
 * sc_ecs.h:199-277 (sparse-set ComponentPool).
 * ---------------------------------------------------------------------------------------- */

#define ORC_INDEX_MASK 0x00FFFFFFu
static inline uint32_t ent_index(uint32_t e) { return e & ORC_INDEX_MASK; }
static inline uint32_t ent_generation(uint32_t e) { return e >> 24; }
static inline uint32_t ent_make(uint32_t idx, uint32_t gen) { return (gen << 24) | (idx & ORC_INDEX_MASK); }

typedef struct {
  uint32_t* dense;      /* m_denseEntities */
  uint8_t*  data;       /* m_data, elem bytes each */
  uint32_t* sparse;     /* m_sparse: dense index + 1, 0 = absent */
  uint32_t  size, cap, sparseLen, sparseCap;
  size_t    elem;
} OrcPool;

struct OrcWorld {
  uint32_t* generations; uint32_t genLen, genCap;
  uint32_t* freeList;    uint32_t freeLen, freeCap;
  uint32_t  alive;
  OrcPool transforms, cameras, meshes, bounds;
};

static void* xrealloc(void* p, size_t n)
{
  void* q = realloc(p, n ? n : 1);
  if (!q) abort();
  return q;
}

static void pool_init(OrcPool* p, size_t elem) { memset(p, 0, sizeof *p); p->elem = elem; }
static void pool_free(OrcPool* p) { free(p->dense); free(p->data); free(p->sparse); }

static int pool_has(const OrcPool* p, uint32_t e)
{
  const uint32_t idx = ent_index(e);
  return idx < p->sparseLen && p->sparse[idx] != 0;
}

static void* pool_get(OrcPool* p, uint32_t e)
{
  const uint32_t idx = ent_index(e);
  if (idx >= p->sparseLen) return NULL;
  const uint32_t slot = p->sparse[idx];
  return slot ? p->data + (size_t)(slot - 1u) * p->elem : NULL;
}

/* sc_ecs.h:203-218.  Returns the slot; *fresh tells whether it was appended. */
static void* pool_add(OrcPool* p, uint32_t e, int* fresh)
{
  const uint32_t idx = ent_index(e);
  if (idx >= p->sparseLen) {
    if (idx + 1u > p->sparseCap) {
      uint32_t nc = p->sparseCap ? p->sparseCap : 64u;
      while (nc < idx + 1u) nc *= 2u;
      p->sparse = xrealloc(p->sparse, (size_t)nc * sizeof(uint32_t));
      p->sparseCap = nc;
    }
    memset(p->sparse + p->sparseLen, 0, (size_t)(idx + 1u - p->sparseLen) * sizeof(uint32_t));
    p->sparseLen = idx + 1u;
  }
  const uint32_t slot = p->sparse[idx];
  if (slot) { *fresh = 0; return p->data + (size_t)(slot - 1u) * p->elem; }
  if (p->size == p->cap) {
    const uint32_t nc = p->cap ? p->cap * 2u : 64u;
    p->dense = xrealloc(p->dense, (size_t)nc * sizeof(uint32_t));
    p->data = xrealloc(p->data, (size_t)nc * p->elem);
    p->cap = nc;
  }
  p->dense[p->size] = e;
  p->sparse[idx] = p->size + 1u;
  *fresh = 1;
  return p->data + (size_t)(p->size++) * p->elem;
}

/* sc_ecs.h:240-262: the last element is swapped into the hole, so dense order changes. */
static void pool_remove(OrcPool* p, uint32_t e)
{
  const uint32_t idx = ent_index(e);
  if (idx >= p->sparseLen) return;
  const uint32_t slot = p->sparse[idx];
  if (!slot) return;
  const uint32_t di = slot - 1u, last = p->size - 1u;
  if (di != last) {
    p->dense[di] = p->dense[last];
    memcpy(p->data + (size_t)di * p->elem, p->data + (size_t)last * p->elem, p->elem);
    p->sparse[ent_index(p->dense[di])] = di + 1u;
  }
  p->size--;
  p->sparse[idx] = 0;
}

OrcWorld* orc_world_new(void)
{
  OrcWorld* w = calloc(1, sizeof *w);
  if (!w) abort();
  pool_init(&w->transforms, sizeof(OrcTransform));
  pool_init(&w->cameras, sizeof(OrcCamera));
  pool_init(&w->meshes, sizeof(OrcRenderMesh));
  pool_init(&w->bounds, sizeof(OrcBounds));
  return w;
}

void orc_world_free(OrcWorld* w)
{
  if (!w) return;
  pool_free(&w->transforms); pool_free(&w->cameras); pool_free(&w->meshes); pool_free(&w->bounds);
  free(w->generations); free(w->freeList); free(w);
}

uint32_t orc_entity_create(OrcWorld* w)
{
  if (w->freeLen) {
    const uint32_t idx = w->freeList[--w->freeLen];
    w->alive++;
    return ent_make(idx, w->generations[idx]);
  }
  if (w->genLen == w->genCap) {
    w->genCap = w->genCap ? w->genCap * 2u : 64u;
    w->generations = xrealloc(w->generations, (size_t)w->genCap * sizeof(uint32_t));
  }
  const uint32_t idx = w->genLen++;
  w->generations[idx] = 0;
  w->alive++;
  return ent_make(idx, 0);
}

int orc_entity_alive(const OrcWorld* w, uint32_t e)
{
  const uint32_t idx = ent_index(e);
  return idx < w->genLen && w->generations[idx] == ent_generation(e);
}

int orc_entity_destroy(OrcWorld* w, uint32_t e)
{
  const uint32_t idx = ent_index(e);
  if (idx >= w->genLen) return 0;
  const uint32_t gen = w->generations[idx];
  if (gen != ent_generation(e)) return 0;
  w->generations[idx] = gen + 1u;     /* kept as a full u32, as the reference does */
  if (w->freeLen == w->freeCap) {
    w->freeCap = w->freeCap ? w->freeCap * 2u : 64u;
    w->freeList = xrealloc(w->freeList, (size_t)w->freeCap * sizeof(uint32_t));
  }
  w->freeList[w->freeLen++] = idx;
  if (w->alive) w->alive--;
  pool_remove(&w->transforms, e); pool_remove(&w->cameras, e);
  pool_remove(&w->meshes, e);     pool_remove(&w->bounds, e);
  return 1;
}

static void transform_default(OrcTransform* t)
{
  memset(t, 0, sizeof *t);
  t->parent = ORC_INVALID_ENTITY;
  t->localScale[0] = t->localScale[1] = t->localScale[2] = 1.0f;
  orc_mat4_identity(t->worldMatrix);
  t->dirty = 1;
}

/* World::add<T>(e) assigns T{} even when the component already exists (sc_ecs.h:292-299). */
OrcTransform* orc_add_transform(OrcWorld* w, uint32_t e)
{
  int fresh; OrcTransform* t = pool_add(&w->transforms, e, &fresh);
  transform_default(t);
  return t;
}
OrcTransform* orc_get_transform(OrcWorld* w, uint32_t e) { return pool_get(&w->transforms, e); }

OrcCamera* orc_add_camera(OrcWorld* w, uint32_t e)
{
  int fresh; OrcCamera* c = pool_add(&w->cameras, e, &fresh);
  memset(c, 0, sizeof *c);
  c->fovY = 60.0f; c->nearZ = 0.1f; c->farZ = 1000.0f; c->aspect = 16.0f / 9.0f; c->active = 0;
  return c;
}
OrcRenderMesh* orc_add_render_mesh(OrcWorld* w, uint32_t e)
{
  int fresh; OrcRenderMesh* m = pool_add(&w->meshes, e, &fresh);
  m->meshId = 0; m->materialId = 0;
  return m;
}
OrcBounds* orc_add_bounds(OrcWorld* w, uint32_t e)
{
  int fresh; OrcBounds* b = pool_add(&w->bounds, e, &fresh);
  memset(b, 0, sizeof *b);
  return b;
}
int orc_has_bounds(const OrcWorld* w, uint32_t e) { return pool_has(&w->bounds, e); }
int orc_has_render_mesh(const OrcWorld* w, uint32_t e) { return pool_has(&w->meshes, e); }

uint32_t orc_transform_count(const OrcWorld* w) { return w->transforms.size; }
const uint32_t* orc_transform_dense_entities(const OrcWorld* w) { return w->transforms.dense; }
OrcTransform* orc_transform_dense_data(OrcWorld* w) { return (OrcTransform*)w->transforms.data; }

int orc_world_build(OrcWorld* w, uint32_t n,
                    const float* pos3, const float* rot3, const float* scale3,
                    const int32_t* parent_index,
                    const uint8_t* has_mesh, const uint32_t* mesh_id, const uint32_t* material_id,
                    const uint8_t* has_bounds, const float* bmin3, const float* bmax3)
{
  if (w->genLen != 0) return 0;                 /* only on an empty world: entity i == index i */
  for (uint32_t i = 0; i < n; ++i) {
    const uint32_t e = orc_entity_create(w);
    OrcTransform* t = orc_add_transform(w, e);
    for (int k = 0; k < 3; ++k) {               /* setLocal, sc_ecs.h:78-84 */
      t->localPos[k] = pos3[3 * i + k];
      t->localRot[k] = rot3[3 * i + k];
      t->localScale[k] = scale3[3 * i + k];
    }
    t->dirty = 1;
    if (parent_index && parent_index[i] >= 0) t->parent = ent_make((uint32_t)parent_index[i], 0);
    if (!has_mesh || has_mesh[i]) {
      OrcRenderMesh* m = orc_add_render_mesh(w, e);
      if (mesh_id) m->meshId = mesh_id[i];
      if (material_id) m->materialId = material_id[i];
    }
    if (!has_bounds || has_bounds[i]) {
      OrcBounds* b = orc_add_bounds(w, e);
      for (int k = 0; k < 3; ++k) { b->min[k] = bmin3[3 * i + k]; b->max[k] = bmax3[3 * i + k]; }
    }
  }
  return 1;
}

void orc_set_local_positions(OrcWorld* w, uint32_t n, const uint32_t* entities, const float* pos3)
{
  for (uint32_t i = 0; i < n; ++i) {
    OrcTransform* t = orc_get_transform(w, entities[i]);
    if (!t) continue;
    t->localPos[0] = pos3[3 * i]; t->localPos[1] = pos3[3 * i + 1]; t->localPos[2] = pos3[3 * i + 2];
    t->dirty = 1;                               /* setLocalPosition, sc_ecs.h:92-96 */
  }
}

void orc_nudge_roots_x(OrcWorld* w, float dx)
{
  OrcTransform* d = (OrcTransform*)w->transforms.data;
  for (uint32_t i = 0; i < w->transforms.size; ++i) {
    if (d[i].parent == ORC_INVALID_ENTITY) { d[i].localPos[0] = d[i].localPos[0] + dx; d[i].dirty = 1; }
  }
}

void orc_mark_dirty(OrcWorld* w, uint32_t n, const uint32_t* entities)
{
  for (uint32_t i = 0; i < n; ++i) {
    OrcTransform* t = orc_get_transform(w, entities[i]);
    if (t) t->dirty = 1;
  }
}

void orc_read_world_matrices(OrcWorld* w, float* out)
{
  const OrcTransform* d = (const OrcTransform*)w->transforms.data;
  for (uint32_t i = 0; i < w->transforms.size; ++i) memcpy(out + 16u * (size_t)i, d[i].worldMatrix, 64);
}
void orc_read_dirty(OrcWorld* w, uint8_t* out)
{
  const OrcTransform* d = (const OrcTransform*)w->transforms.data;
  for (uint32_t i = 0; i < w->transforms.size; ++i) out[i] = d[i].dirty;
}
void orc_read_parents(OrcWorld* w, uint32_t* out)
{
  const OrcTransform* d = (const OrcTransform*)w->transforms.data;
  for (uint32_t i = 0; i < w->transforms.size; ++i) out[i] = d[i].parent;
}
void orc_read_local_scales(OrcWorld* w, float* out)
{
  const OrcTransform* d = (const OrcTransform*)w->transforms.data;
  for (uint32_t i = 0; i < w->transforms.size; ++i) memcpy(out + 3u * (size_t)i, d[i].localScale, 12);
}

/* ------------------------------------------------------------------------------------------
 * Job system as CullingSystem uses it -- JobSystem::Dispatch(count, groupSize, f) + Wait
 * (src/core/include/sc_jobs.h:70-134, src/core/src/sc_jobs.cpp), restated:
 *   - one bounded MPMC ring of kQueueSize = 1024 cells per worker, sequence-numbered cells
 *     (sc_jobs.cpp:12-101);
 *   - Dispatch cuts [0, count) into ceil(count / groupSize) jobs and enqueues them one by one
 *     (sc_jobs.h:88-129); enqueue picks a ring round-robin (m_rr), falls back to the first ring
 *     with room, and when every ring is full runs the job on the calling thread
 *     (sc_jobs.cpp:247-288) -- with 8192 jobs at 1M entities and 1024 slots per worker that
 *     overflow path is the common one on few workers;
 *   - a worker pops its own ring, else steals from the others in index order, else sleeps until
 *     jobs are queued (runOne :290-352, workerMain :354-372);
 *   - Wait: the caller steals from any ring while the fence counts down, else naps 200 us on the
 *     fence (:202-218).
 * Not restated: the per-frame payload arena (2 MB; a job's payload is 24 bytes, so 87k jobs fit:
 * never the limit below 11M candidates), the fence pool (one Dispatch is in flight at a time
 * here), telemetry counters and scoped timers.
 * ---------------------------------------------------------------------------------------- */

typedef void (*OrcRangeFn)(uint32_t start, uint32_t end, void* user);

#define ORC_QUEUE_SIZE 1024u                       /* kQueueSize, sc_jobs.cpp:12 */

typedef struct { uint32_t start, end; OrcRangeFn fn; void* user; } OrcJobItem;
typedef struct { atomic_uint seq; OrcJobItem job; } OrcCell;
typedef struct { OrcCell* buffer; uint32_t mask; atomic_uint enqueuePos, dequeuePos; } OrcRing;

static int ring_enqueue(OrcRing* q, const OrcJobItem* job)               /* :45-72 */
{
  OrcCell* cell;
  uint32_t pos = atomic_load_explicit(&q->enqueuePos, memory_order_relaxed);
  for (;;) {
    cell = &q->buffer[pos & q->mask];
    const uint32_t seq = atomic_load_explicit(&cell->seq, memory_order_acquire);
    const int32_t diff = (int32_t)seq - (int32_t)pos;
    if (diff == 0) { if (atomic_compare_exchange_weak_explicit(&q->enqueuePos, &pos, pos + 1u, memory_order_relaxed, memory_order_relaxed)) break; }
    else if (diff < 0) return 0;                                          /* full */
    else pos = atomic_load_explicit(&q->enqueuePos, memory_order_relaxed);
  }
  cell->job = *job;
  atomic_store_explicit(&cell->seq, pos + 1u, memory_order_release);
  return 1;
}

static int ring_dequeue(OrcRing* q, OrcJobItem* out)                      /* :74-100 */
{
  OrcCell* cell;
  uint32_t pos = atomic_load_explicit(&q->dequeuePos, memory_order_relaxed);
  for (;;) {
    cell = &q->buffer[pos & q->mask];
    const uint32_t seq = atomic_load_explicit(&cell->seq, memory_order_acquire);
    const int32_t diff = (int32_t)seq - (int32_t)(pos + 1u);
    if (diff == 0) { if (atomic_compare_exchange_weak_explicit(&q->dequeuePos, &pos, pos + 1u, memory_order_relaxed, memory_order_relaxed)) break; }
    else if (diff < 0) return 0;                                          /* empty */
    else pos = atomic_load_explicit(&q->dequeuePos, memory_order_relaxed);
  }
  *out = cell->job;
  atomic_store_explicit(&cell->seq, pos + q->mask + 1u, memory_order_release);
  return 1;
}

static struct {
  pthread_t* threads; OrcRing* rings; uint32_t workers; atomic_int shutdown;
  pthread_mutex_t wakeMu; pthread_cond_t wakeCv;                          /* m_wakeMutex / m_wakeCv */
  atomic_uint rr; atomic_int jobsQueued;                                  /* m_rr, m_jobsQueued */
  atomic_int fenceCount; pthread_mutex_t fenceMu; pthread_cond_t fenceCv; /* one JobFence (sc_jobs.h:34-40) */
  atomic_ulong ranInline;                                                 /* jobs the dispatcher had to run itself (all rings full) */
} gJobs = { .wakeMu = PTHREAD_MUTEX_INITIALIZER, .wakeCv = PTHREAD_COND_INITIALIZER,
            .fenceMu = PTHREAD_MUTEX_INITIALIZER, .fenceCv = PTHREAD_COND_INITIALIZER };

static void job_finish(const OrcJobItem* job)
{
  job->fn(job->start, job->end, job->user);
  if (atomic_fetch_sub_explicit(&gJobs.fenceCount, 1, memory_order_acq_rel) - 1 == 0) {
    pthread_mutex_lock(&gJobs.fenceMu);
    pthread_cond_broadcast(&gJobs.fenceCv);
    pthread_mutex_unlock(&gJobs.fenceMu);
  }
}

static int jobs_run_one(uint32_t workerIndex)                             /* runOne, :290-352 */
{
  if (gJobs.workers == 0) return 0;
  OrcJobItem job; int found = 0;
  if (workerIndex < gJobs.workers) {
    found = ring_dequeue(&gJobs.rings[workerIndex], &job);
    for (uint32_t i = 0; i < gJobs.workers && !found; ++i) { if (i != workerIndex) found = ring_dequeue(&gJobs.rings[i], &job); }   /* steal */
  } else {
    for (uint32_t i = 0; i < gJobs.workers && !found; ++i) found = ring_dequeue(&gJobs.rings[i], &job);                           /* main thread helps */
  }
  if (!found) return 0;
  atomic_fetch_sub_explicit(&gJobs.jobsQueued, 1, memory_order_relaxed);
  job_finish(&job);
  return 1;
}

static void* jobs_worker(void* arg)                                        /* workerMain, :354-372 */
{
  const uint32_t index = (uint32_t)(uintptr_t)arg;
  while (!atomic_load_explicit(&gJobs.shutdown, memory_order_relaxed)) {
    if (jobs_run_one(index)) continue;
    pthread_mutex_lock(&gJobs.wakeMu);
    while (!atomic_load_explicit(&gJobs.shutdown, memory_order_relaxed) && atomic_load_explicit(&gJobs.jobsQueued, memory_order_relaxed) <= 0)
      pthread_cond_wait(&gJobs.wakeCv, &gJobs.wakeMu);
    pthread_mutex_unlock(&gJobs.wakeMu);
  }
  return NULL;
}

int orc_jobs_init(uint32_t workers)
{
  orc_jobs_shutdown();
  if (workers == 0) return 1;
  gJobs.threads = calloc(workers, sizeof(pthread_t));
  gJobs.rings = calloc(workers, sizeof(OrcRing));
  if (!gJobs.threads || !gJobs.rings) abort();
  for (uint32_t i = 0; i < workers; ++i) {
    OrcRing* q = &gJobs.rings[i];
    q->buffer = calloc(ORC_QUEUE_SIZE, sizeof(OrcCell));
    if (!q->buffer) abort();
    q->mask = ORC_QUEUE_SIZE - 1u;
    for (uint32_t k = 0; k < ORC_QUEUE_SIZE; ++k) atomic_store_explicit(&q->buffer[k].seq, k, memory_order_relaxed);
    atomic_store(&q->enqueuePos, 0u); atomic_store(&q->dequeuePos, 0u);
  }
  atomic_store(&gJobs.shutdown, 0); atomic_store(&gJobs.rr, 0u); atomic_store(&gJobs.jobsQueued, 0); atomic_store(&gJobs.fenceCount, 0);
  gJobs.workers = workers;
  for (uint32_t i = 0; i < workers; ++i) {
    if (pthread_create(&gJobs.threads[i], NULL, jobs_worker, (void*)(uintptr_t)i) != 0) { gJobs.workers = i; orc_jobs_shutdown(); return 0; }
  }
  return 1;
}

void orc_jobs_shutdown(void)
{
  if (!gJobs.threads) { gJobs.workers = 0; return; }
  atomic_store(&gJobs.shutdown, 1);
  pthread_mutex_lock(&gJobs.wakeMu);
  pthread_cond_broadcast(&gJobs.wakeCv);
  pthread_mutex_unlock(&gJobs.wakeMu);
  for (uint32_t i = 0; i < gJobs.workers; ++i) pthread_join(gJobs.threads[i], NULL);
  for (uint32_t i = 0; gJobs.rings && i < gJobs.workers; ++i) free(gJobs.rings[i].buffer);
  free(gJobs.threads); free(gJobs.rings); gJobs.threads = NULL; gJobs.rings = NULL; gJobs.workers = 0;
}

uint32_t orc_jobs_workers(void) { return gJobs.workers; }
unsigned long orc_jobs_ran_inline(void) { return atomic_load(&gJobs.ranInline); }

static void jobs_enqueue(const OrcJobItem* job)                            /* enqueue, :247-288 */
{
  const uint32_t idx = atomic_fetch_add_explicit(&gJobs.rr, 1u, memory_order_relaxed) % gJobs.workers;
  int placed = ring_enqueue(&gJobs.rings[idx], job);
  for (uint32_t i = 0; i < gJobs.workers && !placed; ++i) placed = ring_enqueue(&gJobs.rings[i], job);   /* full: linear search */
  if (placed) {
    atomic_fetch_add_explicit(&gJobs.jobsQueued, 1, memory_order_relaxed);
    pthread_mutex_lock(&gJobs.wakeMu);                 /* (notify_one under the mutex: no lost wake-up against the predicate above) */
    pthread_cond_signal(&gJobs.wakeCv);
    pthread_mutex_unlock(&gJobs.wakeMu);
    return;
  }
  atomic_fetch_add(&gJobs.ranInline, 1ul);
  job_finish(job);                                     /* every ring full: the caller runs it, nothing is lost */
}

static void jobs_dispatch_wait(uint32_t count, uint32_t groupSize, OrcRangeFn fn, void* user)
{
  if (count == 0 || groupSize == 0) return;
  const uint32_t groups = (count + groupSize - 1u) / groupSize;
  if (gJobs.workers == 0) {                            /* no job system: the ranges run in order on the caller */
    for (uint32_t g = 0; g < groups; ++g) {
      const uint32_t s = g * groupSize, e = (s + groupSize > count) ? count : s + groupSize;
      fn(s, e, user);
    }
    return;
  }
  atomic_store_explicit(&gJobs.fenceCount, (int)groups, memory_order_release);          /* allocFence(groupCount) */
  for (uint32_t g = 0; g < groups; ++g) {                                                 /* Dispatch, sc_jobs.h:88-129 */
    OrcJobItem job;
    job.start = g * groupSize; job.end = (job.start + groupSize > count) ? count : job.start + groupSize;
    job.fn = fn; job.user = user;
    jobs_enqueue(&job);
  }
  while (atomic_load_explicit(&gJobs.fenceCount, memory_order_acquire) > 0) {            /* Wait, sc_jobs.cpp:202-218 */
    if (jobs_run_one(gJobs.workers)) continue;
    struct timespec ts;
    clock_gettime(CLOCK_REALTIME, &ts);
    ts.tv_nsec += 200000; if (ts.tv_nsec >= 1000000000L) { ts.tv_nsec -= 1000000000L; ts.tv_sec += 1; }
    pthread_mutex_lock(&gJobs.fenceMu);
    if (atomic_load_explicit(&gJobs.fenceCount, memory_order_acquire) > 0) pthread_cond_timedwait(&gJobs.fenceCv, &gJobs.fenceMu, &ts);
    pthread_mutex_unlock(&gJobs.fenceMu);
  }
}

/* ------------------------------------------------------------------------------------------
 * TransformSystem -- sc_ecs.cpp:118-211.
 * Kept structurally faithful on purpose (per-tick entity gather, one growable child list per
 * entity index allocated every tick, explicit-stack DFS) because this is also the CPU baseline.
 * ---------------------------------------------------------------------------------------- */

typedef struct { uint32_t* p; uint32_t n, cap; } OrcEntVec;

static void entvec_push(OrcEntVec* v, uint32_t e)
{
  if (v->n == v->cap) {
    v->cap = v->cap ? v->cap * 2u : 1u;          /* std::vector doubling from 1 */
    v->p = xrealloc(v->p, (size_t)v->cap * sizeof(uint32_t));
  }
  v->p[v->n++] = e;
}

void orc_transform_system(OrcWorld* w)
{
  OrcPool* tp = &w->transforms;
  if (tp->size == 0) return;

  /* :122-126 gather in pool dense order */
  OrcEntVec entities = {0};
  for (uint32_t i = 0; i < tp->size; ++i) entvec_push(&entities, tp->dense[i]);

  /* :131-137 */
  uint32_t maxIndex = 0;
  for (uint32_t i = 0; i < entities.n; ++i) {
    const uint32_t ix = ent_index(entities.p[i]);
    if (ix > maxIndex) maxIndex = ix;
  }
  OrcEntVec* children = calloc((size_t)maxIndex + 1u, sizeof(OrcEntVec));
  uint32_t* roots = xrealloc(NULL, (size_t)entities.n * sizeof(uint32_t));
  uint32_t rootCount = 0;
  if (!children) abort();

  /* :139-165 */
  for (uint32_t i = 0; i < entities.n; ++i) {
    const uint32_t e = entities.p[i];
    OrcTransform* t = pool_get(tp, e);

    if (t->localScale[0] == 0.0f && t->localScale[1] == 0.0f && t->localScale[2] == 0.0f) {
      t->localScale[0] = t->localScale[1] = t->localScale[2] = 1.0f;
      t->dirty = 1;
    }

    const uint32_t p = t->parent;
    const int validParent = p != ORC_INVALID_ENTITY && p != e && orc_entity_alive(w, p) && pool_has(tp, p);
    if (!validParent) {
      if (p != ORC_INVALID_ENTITY) t->dirty = 1;
      t->parent = ORC_INVALID_ENTITY;
      roots[rootCount++] = e;
    } else {
      entvec_push(&children[ent_index(p)], e);
    }
  }

  /* :167-210 explicit-stack DFS; an entity in a parent cycle has no root above it and is never visited */
  typedef struct { uint32_t e; uint8_t parentDirty; } Item;
  Item* stack = xrealloc(NULL, (size_t)entities.n * sizeof(Item));
  uint32_t sp = 0;
  for (uint32_t i = 0; i < rootCount; ++i) { stack[sp].e = roots[i]; stack[sp].parentDirty = 0; sp++; }

  while (sp) {
    const Item it = stack[--sp];
    OrcTransform* t = pool_get(tp, it.e);
    const int nodeDirty = t->dirty || it.parentDirty;
    if (nodeDirty) {
      float local[16];
      orc_mat4_trs(t->localPos, t->localRot, t->localScale, local);
      if (t->parent != ORC_INVALID_ENTITY) {
        OrcTransform* pt = pool_get(tp, t->parent);
        if (pt) orc_mat4_mul(pt->worldMatrix, local, t->worldMatrix);
        else memcpy(t->worldMatrix, local, sizeof local);
      } else {
        memcpy(t->worldMatrix, local, sizeof local);
      }
      t->dirty = 0;
    }
    const uint32_t ix = ent_index(it.e);
    if (ix <= maxIndex) {
      const OrcEntVec* ch = &children[ix];
      for (uint32_t c = 0; c < ch->n; ++c) { stack[sp].e = ch->p[c]; stack[sp].parentDirty = (uint8_t)nodeDirty; sp++; }
    }
  }

  for (uint32_t i = 0; i <= maxIndex; ++i) free(children[i].p);
  free(children); free(roots); free(stack); free(entities.p);
}

/* ------------------------------------------------------------------------------------------
 * CameraSystem -- sc_ecs.cpp:213-272.  ForEach<Camera, Transform> is driven by the Camera pool.
 * ---------------------------------------------------------------------------------------- */
void orc_camera_system(OrcWorld* w, OrcCameraState* st)
{
  if (!st) return;
  OrcCamera *active = NULL, *fallback = NULL;
  OrcTransform *activeT = NULL, *fallbackT = NULL;
  uint32_t activeE = ORC_INVALID_ENTITY, fallbackE = ORC_INVALID_ENTITY;

  for (uint32_t i = 0; i < w->cameras.size; ++i) {
    const uint32_t e = w->cameras.dense[i];
    OrcTransform* t = pool_get(&w->transforms, e);
    if (!t) continue;
    OrcCamera* c = (OrcCamera*)w->cameras.data + i;
    if (!fallback) { fallback = c; fallbackT = t; fallbackE = e; }
    if (!active && c->active) { active = c; activeT = t; activeE = e; }
  }
  if (!active && fallback) { active = fallback; activeT = fallbackT; activeE = fallbackE; }
  if (!active) { orc_mat4_identity(st->viewProj); st->activeCamera = ORC_INVALID_ENTITY; return; }

  active->aspect = (st->aspect > 0.0f) ? st->aspect : active->aspect;
  const float fovRad = active->fovY * 3.1415926535f / 180.0f;
  float proj[16], view[16];
  orc_mat4_perspective_rh_zo(fovRad, active->aspect, active->nearZ, active->farZ, 1, proj);
  orc_mat4_inverse(activeT->worldMatrix, view);
  orc_mat4_mul(proj, view, st->viewProj);
  st->activeCamera = activeE;
}

/* ------------------------------------------------------------------------------------------
 * Culling -- sc_world_partition.cpp:1071-1144, 1199-1284.
 * ---------------------------------------------------------------------------------------- */

/* :1071-1103.  Rows of the column-major viewProj; the "near" plane is r3 + r2 (the OpenGL form,
 * kept although depth is 0..1 -- a reference quirk that must be reproduced). */
void orc_frustum_from_viewproj(const float m[16], OrcFrustum* out)
{
  const float row[4][4] = {
    { m[0], m[4], m[8],  m[12] },
    { m[1], m[5], m[9],  m[13] },
    { m[2], m[6], m[10], m[14] },
    { m[3], m[7], m[11], m[15] },
  };
  memset(out, 0, sizeof *out);
  for (int p = 0; p < 6; ++p) {
    const int axis = p >> 1;
    const int minus = p & 1;           /* 0: r3 + r_axis, 1: r3 - r_axis */
    float v[4];
    for (int k = 0; k < 4; ++k) v[k] = minus ? row[3][k] - row[axis][k] : row[3][k] + row[axis][k];
    const float lenSq = v[0] * v[0] + v[1] * v[1] + v[2] * v[2];
    if (lenSq > 1e-8f) {
      const float invLen = 1.0f / sqrtf(lenSq);
      out->planes[p].n[0] = v[0] * invLen;
      out->planes[p].n[1] = v[1] * invLen;
      out->planes[p].n[2] = v[2] * invLen;
      out->planes[p].d = v[3] * invLen;
    }
  }
  out->valid = 1;
}

/* :1105-1117 */
int orc_sphere_in_frustum(const OrcFrustum* f, const float c[3], float radius)
{
  if (!f->valid) return 1;
  for (int p = 0; p < 6; ++p) {
    const OrcPlane* pl = &f->planes[p];
    const float d = pl->n[0] * c[0] + pl->n[1] * c[1] + pl->n[2] * c[2] + pl->d;
    if (d < -radius) return 0;
  }
  return 1;
}

/* :1119-1144 */
void orc_world_bounds_sphere(const float m[16], const OrcBounds* b, float outCenter[3], float* outRadius)
{
  const float cx = (b->min[0] + b->max[0]) * 0.5f, cy = (b->min[1] + b->max[1]) * 0.5f, cz = (b->min[2] + b->max[2]) * 0.5f;
  const float ex = (b->max[0] - b->min[0]) * 0.5f, ey = (b->max[1] - b->min[1]) * 0.5f, ez = (b->max[2] - b->min[2]) * 0.5f;
  outCenter[0] = m[0] * cx + m[4] * cy + m[8] * cz + m[12];
  outCenter[1] = m[1] * cx + m[5] * cy + m[9] * cz + m[13];
  outCenter[2] = m[2] * cx + m[6] * cy + m[10] * cz + m[14];
  const float sx = sqrtf(m[0] * m[0] + m[1] * m[1] + m[2] * m[2]);
  const float sy = sqrtf(m[4] * m[4] + m[5] * m[5] + m[6] * m[6]);
  const float sz = sqrtf(m[8] * m[8] + m[9] * m[9] + m[10] * m[10]);
  const float syz = (sy < sz) ? sz : sy;          /* std::max(a,b) == (a<b)?b:a */
  const float maxScale = (sx < syz) ? syz : sx;
  const float localRadius = sqrtf(ex * ex + ey * ey + ez * ez);
  *outRadius = localRadius * maxScale;
}

OrcCullingState* orc_culling_state_new(void) { OrcCullingState* s = calloc(1, sizeof *s); if (!s) abort(); return s; }
void orc_culling_state_free(OrcCullingState* s)
{
  if (!s) return;
  free(s->candidates); free(s->visible); free(s->culled); free(s->visibilityMask); free(s);
}

static void u32vec_reserve(uint32_t** p, uint32_t* cap, uint32_t need)
{
  if (*cap >= need) return;
  *p = xrealloc(*p, (size_t)need * sizeof(uint32_t));
  *cap = need;
}

typedef struct { OrcWorld* w; OrcCullingState* s; OrcFrustum fr; } CullJob;

static void cull_range(uint32_t start, uint32_t end, void* user)
{
  CullJob* j = user;
  for (uint32_t i = start; i < end; ++i) {                 /* :1242-1269 */
    const uint32_t e = j->s->candidates[i];
    const OrcTransform* t = pool_get(&j->w->transforms, e);
    if (!t) { j->s->visibilityMask[i] = 0; continue; }
    if (!pool_has(&j->w->bounds, e)) { j->s->visibilityMask[i] = 1; continue; }
    const OrcBounds* b = pool_get(&j->w->bounds, e);
    if (!b) { j->s->visibilityMask[i] = 1; continue; }
    float c[3], r;
    orc_world_bounds_sphere(t->worldMatrix, b, c, &r);
    j->s->visibilityMask[i] = orc_sphere_in_frustum(&j->fr, c, r) ? 1u : 0u;
  }
}

void orc_culling_system(OrcWorld* w, OrcCullingState* s, const float viewProj[16])
{
  if (!s || !viewProj) return;

  /* :1206-1210 candidates = Transform-pool dense order filtered by RenderMesh */
  s->candidatesLen = 0;
  u32vec_reserve(&s->candidates, &s->candidatesCap, w->transforms.size);
  for (uint32_t i = 0; i < w->transforms.size; ++i) {
    const uint32_t e = w->transforms.dense[i];
    if (pool_has(&w->meshes, e)) s->candidates[s->candidatesLen++] = e;
  }
  const uint32_t total = s->candidatesLen;
  s->renderablesTotal = total;
  s->visibleLen = s->culledLen = 0;
  u32vec_reserve(&s->visible, &s->visibleCap, total);
  u32vec_reserve(&s->culled, &s->culledCap, total);
  if (total == 0) { s->visibleCount = s->culledCount = 0; return; }

  if (s->freezeCulling) {                                   /* :1227-1233 */
    memcpy(s->visible, s->candidates, (size_t)total * sizeof(uint32_t));
    s->visibleLen = total; s->visibleCount = total; s->culledCount = 0;
    return;
  }

  orc_frustum_from_viewproj(viewProj, &s->frustum);
  if (s->maskLen < total) { s->visibilityMask = xrealloc(s->visibilityMask, total); s->maskLen = total; }

  CullJob job = { w, s, s->frustum };
  jobs_dispatch_wait(total, 128u, cull_range, &job);        /* :1240-1271 */

  for (uint32_t i = 0; i < total; ++i) {                    /* :1273-1280 serial stable compaction */
    const uint32_t e = s->candidates[i];
    if (s->visibilityMask[i]) s->visible[s->visibleLen++] = e;
    else s->culled[s->culledLen++] = e;
  }
  s->visibleCount = s->visibleLen;
  s->culledCount = s->culledLen;
}

/* :1286-1359, draw emission only (asset touching / eviction are renderer-side and out of scope). */
uint32_t orc_render_prep_streaming(OrcWorld* w, const OrcCullingState* s, uint32_t maxDraws,
                                   OrcDrawItem* out, uint32_t outCap, uint32_t* droppedOut)
{
  uint32_t emitted = 0, dropped = 0;
  for (uint32_t i = 0; i < s->visibleLen; ++i) {
    const uint32_t e = s->visible[i];
    const OrcTransform* t = pool_get(&w->transforms, e);
    const OrcRenderMesh* rm = pool_get(&w->meshes, e);
    if (!t || !rm) continue;
    if (maxDraws > 0 && emitted >= maxDraws) { dropped++; continue; }
    if (emitted < outCap) {
      OrcDrawItem* d = &out[emitted];
      d->entity = e; d->meshId = rm->meshId; d->materialId = rm->materialId; d->_pad = 0;
      memcpy(d->model, t->worldMatrix, 64);
    }
    emitted++;
  }
  if (droppedOut) *droppedOut = dropped;
  return emitted;
}

/* The renderer's draw order, src/engine/src/sc_vk.cpp:1842-1864: draws with meshId >= meshCount or without a
 * Material are skipped (:1846-1851), the rest ordered by (pipelineId of the material, materialId, meshId) (:1854-1864).
 * pipelineOfMaterial[h] = 0xFF means getMaterial(h) == nullptr.  std::sort does not define the order of equal
 * keys; this restatement uses a stable insertion order (equal keys keep RenderFrameData::draws order), one of the
 * orders the reference can produce.  Writes indices into `items`; returns how many draws are left. */
static int draw_before(const OrcDrawItem* a, const OrcDrawItem* b, const uint8_t* pipe)
{
  const uint32_t pa = pipe[a->materialId], pb = pipe[b->materialId];
  if (pa != pb) return pa < pb;
  if (a->materialId != b->materialId) return a->materialId < b->materialId;
  return a->meshId < b->meshId;
}
uint32_t orc_renderer_draw_order(const OrcDrawItem* items, uint32_t n, const uint8_t* pipelineOfMaterial,
                                 uint32_t materialCount, uint32_t meshCount, uint32_t* order)
{
  uint32_t kept = 0;
  for (uint32_t i = 0; i < n; ++i) {
    if (items[i].meshId >= meshCount) continue;
    if (items[i].materialId >= materialCount || pipelineOfMaterial[items[i].materialId] == 0xFF) continue;
    /* stable insertion from the back: stop at the first element that is not after the new one */
    uint32_t at = kept;
    while (at > 0 && draw_before(&items[i], &items[order[at - 1]], pipelineOfMaterial)) { order[at] = order[at - 1]; --at; }
    order[at] = i;
    ++kept;
  }
  return kept;
}

/* ---- ray queries (own spec, include/sc_tick.h "ray queries"): brute force over every box --------------------
 * direction handling and result fields after PhysicsWorld::raycast (sc_physics.cpp:740-777), the ray-box test is
 * intersectRayAABB (tools/world_editor/editor_core/editor_core.cpp:438-470) with the far limit at maxDist, closest
 * hit wins, equal distances go to the lower index (PickEntity, :487-491). */
static int ray_box(const float o[3], const float dir[3], float maxDist, const float* mn, const float* mx, float* tOut, int* axisOut)
{
  float tmin = 0.0f, tmax = maxDist;
  int axis = 3;
  for (int i = 0; i < 3; ++i) {
    if (fabsf(dir[i]) < 1e-6f) {
      if (o[i] < mn[i] || o[i] > mx[i]) return 0;
    } else {
      const float ood = 1.0f / dir[i];
      float t1 = (mn[i] - o[i]) * ood, t2 = (mx[i] - o[i]) * ood;
      if (t1 > t2) { const float k = t1; t1 = t2; t2 = k; }
      if (t1 > tmin) { tmin = t1; axis = i; }
      tmax = tmax < t2 ? tmax : t2;
      if (tmin > tmax) return 0;
    }
  }
  *tOut = tmin; *axisOut = axis;
  return 1;
}

/* the slab test alone, for pinning it against the reference's own intersectRayAABB (tests/golden/ray_aabb_ref.npz): tmax = 1e30f there */
int orc_ray_box_probe(const float origin[3], const float dir[3], float tmax, const float mn[3], const float mx[3], float* tOut)
{
  int axis; float t = 0.0f;
  const int hit = ray_box(origin, dir, tmax, mn, mx, &t, &axis);
  if (hit && tOut) *tOut = t;
  return hit;
}

void orc_raycast_boxes(uint32_t n, const float* min3, const float* max3, const uint32_t* group, const uint32_t* mask,
                       uint32_t rays, const float* origin3, const float* dir3, const float* maxDist, const uint32_t* rayMask,
                       OrcRayHit* out)
{
  for (uint32_t r = 0; r < rays; ++r) {
    OrcRayHit h; memset(&h, 0, sizeof h);
    h.id = 0xFFFFFFFFu; h.normal[1] = 1.0f;
    const float* o = origin3 + 3 * r; const float* dv = dir3 + 3 * r;
    const float lenSq = dv[0] * dv[0] + dv[1] * dv[1] + dv[2] * dv[2];
    if (lenSq > 1e-6f && maxDist[r] >= 0.0f) {
      const float invLen = 1.0f / sqrtf(lenSq);
      const float dir[3] = { dv[0] * invLen, dv[1] * invLen, dv[2] * invLen };
      float best = INFINITY; int bestAxis = 3;
      for (uint32_t i = 0; i < n; ++i) {
        if (!(min3[3 * i] <= max3[3 * i])) continue;                      /* no Bounds (or NaN): no box */
        const uint32_t g = group[i] & 0xFFFFu, m = mask[i] & 0xFFFFu;
        if (!(g & rayMask[r]) || !m) continue;
        float t; int axis;
        if (!ray_box(o, dir, maxDist[r], min3 + 3 * i, max3 + 3 * i, &t, &axis)) continue;
        if (t < best) { best = t; bestAxis = axis; h.hit = 1; h.id = i; h.layer = g; }
      }
      if (h.hit) {
        h.distance = best;
        for (int k = 0; k < 3; ++k) h.position[k] = o[k] + dir[k] * best;
        if (bestAxis < 3) { h.normal[0] = h.normal[1] = h.normal[2] = 0.0f; h.normal[bestAxis] = dir[bestAxis] > 0.0f ? -1.0f : 1.0f; }
      }
    }
    out[r] = h;
  }
}

/* isOccupiedWorld, src/engine/traffic/sc_traffic_spawner.cpp:93-116 (distanceSq2d :66-71): some agent with
 * dx*dx + dz*dz < radius*radius on Transform::localPos.  The reference walks its TrafficAgent and VehicleComponent pools;
 * agents are given here as a flag per Transform-pool entity. */
int orc_is_occupied(OrcWorld* w, const uint8_t* isAgent, const float pos[3], float radius)
{
  const OrcTransform* d = (const OrcTransform*)w->transforms.data;
  const float r2 = radius * radius;
  for (uint32_t i = 0; i < w->transforms.size; ++i) {
    if (!isAgent[i]) continue;
    const float dx = d[i].localPos[0] - pos[0], dz = d[i].localPos[2] - pos[2];
    if (dx * dx + dz * dz < r2) return 1;
  }
  return 0;
}

/* sc_world_partition.cpp:268-275 */
void orc_world_to_sector(float sectorSize, float x, float z, int32_t* sx, int32_t* sz)
{
  const float inv = 1.0f / sectorSize;
  *sx = (int32_t)floorf(x * inv);
  *sz = (int32_t)floorf(z * inv);
}

/* Upstream movers: pos += vel*dt (two roundings), vehicles (1) wrap inside [lo,hi), peds (2) reflect. */
void orc_advance_movers(OrcWorld* w, const uint8_t* kind, float* vel, const float* lo, const float* hi, float dt)
{
  OrcTransform* d = (OrcTransform*)w->transforms.data;
  for (uint32_t i = 0; i < w->transforms.size; ++i) {
    if (!kind[i]) continue;
    float vx = vel[2 * i], vz = vel[2 * i + 1];
    const float lox = lo[2 * i], loz = lo[2 * i + 1], hix = hi[2 * i], hiz = hi[2 * i + 1];
    float x = d[i].localPos[0] + vx * dt, z = d[i].localPos[2] + vz * dt;
    if (kind[i] == 1) {
      if (x >= hix) x = lox + (x - hix); else if (x < lox) x = hix - (lox - x);
      if (z >= hiz) z = loz + (z - hiz); else if (z < loz) z = hiz - (loz - z);
    } else {
      if (x > hix) { x = hix - (x - hix); vx = -vx; } else if (x < lox) { x = lox + (lox - x); vx = -vx; }
      if (z > hiz) { z = hiz - (z - hiz); vz = -vz; } else if (z < loz) { z = loz + (loz - z); vz = -vz; }
      vel[2 * i] = vx; vel[2 * i + 1] = vz;
    }
    d[i].localPos[0] = x; d[i].localPos[2] = z;
    d[i].dirty = 1;
  }
}

void orc_tick(OrcWorld* w, OrcCameraState* cam, OrcCullingState* cull)
{
  orc_transform_system(w);
  orc_camera_system(w, cam);
  orc_culling_system(w, cull, cam->viewProj);
}

/* ------------------------------------------------------------------------------------------
 * Broadphase -- this build's own spec (the reference hands the job to Bullet 3.25's
 * btDbvtBroadphase, sc_physics.cpp:218-225, whose source is not in the tree).
 *   world AABB: centre as computeWorldBoundsSphere's centre; half extent h_r = |M[r,0]|*ex +
 *   |M[r,1]|*ey + |M[r,2]|*ez; min = c - h, max = c + h (fp32).
 *   pair (i<j) iff closed intervals overlap on all three axes and Bullet's default filter
 *   (gi & mj) && (gj & mi) passes (sc_physics.cpp:372-379 group/mask rules).
 * ---------------------------------------------------------------------------------------- */
void orc_world_aabb(const float m[16], const OrcBounds* b, float outMin[3], float outMax[3])
{
  const float cx = (b->min[0] + b->max[0]) * 0.5f, cy = (b->min[1] + b->max[1]) * 0.5f, cz = (b->min[2] + b->max[2]) * 0.5f;
  const float ex = (b->max[0] - b->min[0]) * 0.5f, ey = (b->max[1] - b->min[1]) * 0.5f, ez = (b->max[2] - b->min[2]) * 0.5f;
  for (int r = 0; r < 3; ++r) {
    const float c = m[r] * cx + m[4 + r] * cy + m[8 + r] * cz + m[12 + r];
    const float h = fabsf(m[r]) * ex + fabsf(m[4 + r]) * ey + fabsf(m[8 + r]) * ez;
    outMin[r] = c - h;
    outMax[r] = c + h;
  }
}

void orc_read_world_aabbs(OrcWorld* w, float* min3n, float* max3n)
{
  const OrcTransform* d = (const OrcTransform*)w->transforms.data;
  for (uint32_t i = 0; i < w->transforms.size; ++i) {
    const OrcBounds* b = pool_get(&w->bounds, w->transforms.dense[i]);
    if (b) orc_world_aabb(d[i].worldMatrix, b, min3n + 3u * (size_t)i, max3n + 3u * (size_t)i);
    else for (int k = 0; k < 3; ++k) { min3n[3u * (size_t)i + k] = INFINITY; max3n[3u * (size_t)i + k] = -INFINITY; }
  }
}

static inline int aabb_overlap(const float* amin, const float* amax, const float* bmin, const float* bmax)
{
  return amin[0] <= bmax[0] && bmin[0] <= amax[0] &&
         amin[1] <= bmax[1] && bmin[1] <= amax[1] &&
         amin[2] <= bmax[2] && bmin[2] <= amax[2];
}
static inline int filter_pass(uint32_t gi, uint32_t mi, uint32_t gj, uint32_t mj)
{
  return (gi & mj) != 0 && (gj & mi) != 0;
}

uint64_t orc_broadphase_bruteforce(uint32_t n, const float* mn, const float* mx,
                                   const uint32_t* group, const uint32_t* mask,
                                   uint32_t* pairs, uint64_t cap)
{
  uint64_t count = 0;
  for (uint32_t i = 0; i < n; ++i) {
    for (uint32_t j = i + 1u; j < n; ++j) {
      if (!aabb_overlap(mn + 3u * (size_t)i, mx + 3u * (size_t)i, mn + 3u * (size_t)j, mx + 3u * (size_t)j)) continue;
      if (!filter_pass(group[i], mask[i], group[j], mask[j])) continue;
      if (count < cap) { pairs[2 * count] = i; pairs[2 * count + 1] = j; }
      count++;
    }
  }
  return count;
}

typedef struct { uint64_t key; uint32_t obj; } GridEntry;
static int grid_entry_cmp(const void* a, const void* b)
{
  const GridEntry *x = a, *y = b;
  if (x->key != y->key) return x->key < y->key ? -1 : 1;
  return (x->obj > y->obj) - (x->obj < y->obj);
}
static int u64_cmp(const void* a, const void* b)
{
  const uint64_t x = *(const uint64_t*)a, y = *(const uint64_t*)b;
  return (x > y) - (x < y);
}
static inline int32_t cell_of(float v, float inv) { return (int32_t)floorf(v * inv); }   /* worldToSector arithmetic */
static inline uint64_t cell_key(int32_t cx, int32_t cz) { return ((uint64_t)(uint32_t)cz << 32) | (uint32_t)cx; }

uint64_t orc_broadphase_grid(uint32_t n, const float* mn, const float* mx,
                             const uint32_t* group, const uint32_t* mask, float cellSize,
                             uint32_t* pairs, uint64_t cap)
{
  const float inv = 1.0f / cellSize;
  /* every object is entered into each xz cell its AABB touches; a pair is reported only from the
   * cell holding the low corner of the xz intersection, so no pair is produced twice.  A box that
   * would cover more than kBigCells cells is kept out of the grid and tested against everything. */
  const int64_t kBigCells = 4096;
  uint8_t* big = calloc(n ? n : 1, 1);
  size_t entries = 0;
  for (uint32_t i = 0; i < n; ++i) {
    if (!(mn[3u * (size_t)i] <= mx[3u * (size_t)i])) continue;      /* empty (no Bounds) */
    const float wx = (mx[3u * (size_t)i] - mn[3u * (size_t)i]) * inv, wz = (mx[3u * (size_t)i + 2] - mn[3u * (size_t)i + 2]) * inv;
    if (!(wx < 60.0f && wz < 60.0f)) { big[i] = 1; continue; }
    const int64_t nx = (int64_t)cell_of(mx[3u * (size_t)i], inv) - cell_of(mn[3u * (size_t)i], inv) + 1;
    const int64_t nz = (int64_t)cell_of(mx[3u * (size_t)i + 2], inv) - cell_of(mn[3u * (size_t)i + 2], inv) + 1;
    if (nx * nz > kBigCells) { big[i] = 1; continue; }
    entries += (size_t)(nx * nz);
  }
  GridEntry* g = xrealloc(NULL, entries * sizeof(GridEntry));
  size_t k = 0;
  for (uint32_t i = 0; i < n; ++i) {
    if (big[i] || !(mn[3u * (size_t)i] <= mx[3u * (size_t)i])) continue;
    const int32_t x0 = cell_of(mn[3u * (size_t)i], inv), x1 = cell_of(mx[3u * (size_t)i], inv);
    const int32_t z0 = cell_of(mn[3u * (size_t)i + 2], inv), z1 = cell_of(mx[3u * (size_t)i + 2], inv);
    for (int32_t z = z0; z <= z1; ++z) for (int32_t x = x0; x <= x1; ++x) { g[k].key = cell_key(x, z); g[k].obj = i; k++; }
  }
  qsort(g, entries, sizeof(GridEntry), grid_entry_cmp);

  uint64_t* found = NULL; size_t foundLen = 0, foundCap = 0;
  for (size_t s = 0; s < entries;) {
    size_t e = s;
    while (e < entries && g[e].key == g[s].key) e++;
    const int32_t ccx = (int32_t)(uint32_t)(g[s].key & 0xFFFFFFFFu), ccz = (int32_t)(uint32_t)(g[s].key >> 32);
    for (size_t a = s; a < e; ++a) for (size_t b = a + 1; b < e; ++b) {
      const uint32_t i = g[a].obj, j = g[b].obj;          /* i < j by the sort */
      const float* imn = mn + 3u * (size_t)i; const float* imx = mx + 3u * (size_t)i;
      const float* jmn = mn + 3u * (size_t)j; const float* jmx = mx + 3u * (size_t)j;
      if (!aabb_overlap(imn, imx, jmn, jmx)) continue;
      const float lx = imn[0] > jmn[0] ? imn[0] : jmn[0], lz = imn[2] > jmn[2] ? imn[2] : jmn[2];
      if (cell_of(lx, inv) != ccx || cell_of(lz, inv) != ccz) continue;
      if (!filter_pass(group[i], mask[i], group[j], mask[j])) continue;
      if (foundLen == foundCap) { foundCap = foundCap ? foundCap * 2 : 1024; found = xrealloc(found, foundCap * sizeof(uint64_t)); }
      found[foundLen++] = ((uint64_t)i << 32) | j;
    }
    s = e;
  }
  free(g);
  for (uint32_t b = 0; b < n; ++b) {                       /* oversized boxes: against everything, once */
    if (!big[b]) continue;
    for (uint32_t j = 0; j < n; ++j) {
      if (j == b || (big[j] && j < b)) continue;
      const uint32_t lo = b < j ? b : j, hi = b < j ? j : b;
      if (!aabb_overlap(mn + 3u * (size_t)lo, mx + 3u * (size_t)lo, mn + 3u * (size_t)hi, mx + 3u * (size_t)hi)) continue;
      if (!filter_pass(group[lo], mask[lo], group[hi], mask[hi])) continue;
      if (foundLen == foundCap) { foundCap = foundCap ? foundCap * 2 : 1024; found = xrealloc(found, foundCap * sizeof(uint64_t)); }
      found[foundLen++] = ((uint64_t)lo << 32) | hi;
    }
  }
  free(big);
  qsort(found, foundLen, sizeof(uint64_t), u64_cmp);
  for (size_t q = 0; q < foundLen && q < cap; ++q) { pairs[2 * q] = (uint32_t)(found[q] >> 32); pairs[2 * q + 1] = (uint32_t)found[q]; }
  free(found);
  return foundLen;
}
