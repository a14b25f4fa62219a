/*
 * sc_oracle.h -- CPU oracle for the world-tick path.  TEST INFRASTRUCTURE ONLY.
 *
 * This is a from-scratch C restatement of the reference's algorithm for the
 * RenderPrep chain (TransformSystem -> CameraSystem -> CullingSystem ->
 * RenderPrepStreamingSystem) plus the broadphase spec this build defines.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * it; the product path (libsc_tick.so) never does.
 *
 * Pin status (see DESIGN.md "Oracle"):
 *   PINNED   orc_mat4_*            bit-for-bit against the reference's own sc_math.cpp
 *                                  compiled unmodified (oracle/_ref, tests/golden/sc_math_ref.npz)
 *   PINNED   pool / entity rules   against the reference's header-only ComponentPool<T>/Entity
 *                                  (tests/golden/sc_ecs_pool_ref.json)
 *   UNPINNED orc_transform_system, orc_camera_system, orc_frustum_from_viewproj,
 *            orc_sphere_in_frustum, orc_world_bounds_sphere, orc_culling_system,
 *            orc_render_prep_streaming: "parity unpinned" -- the reference holds no test,
 *            fixture or golden vector for them and their translation units do not build
 *            here without stand-ins (strncpy_s, <windows.h>, <vulkan/vulkan.h>), so they are
 *            restated from the source text and cross-checked only by a second independent
 *            numpy restatement (oracle/oracle_np.py).
 *   OWN SPEC orc_broadphase_*      the reference delegates to Bullet 3.25 (absent); the spec
 *                                  is this build's, pinned by brute force vs grid.
 *
 * All file:line citations are relative to the reference tree.
 */
#ifndef SC_ORACLE_H
#define SC_ORACLE_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_INVALID_ENTITY 0xFFFFFFFFu

/* ---- math: src/core/src/sc_math.cpp ---- */
void orc_mat4_identity(float out[16]);
void orc_mat4_mul(const float a[16], const float b[16], float out[16]);              /* :11-85  */
void orc_mat4_rotation_xyz(float rx, float ry, float rz, float out[16]);             /* :100-128 */
void orc_mat4_trs(const float pos[3], const float rot[3], const float scale[3], float out[16]); /* :130-142 */
void orc_mat4_inverse(const float a[16], float out[16]);                             /* :144-207 */
void orc_mat4_perspective_rh_zo(float fovYRadians, float aspect, float zNear, float zFar,
                                int flipY, float out[16]);                           /* :209-232 */

/* ---- ECS storage: src/core/include/sc_ecs.h ---- */
typedef struct OrcTransform {           /* sc_ecs.h:63-71, sizeof == 128 */
  uint32_t parent;                      /* Entity value, 0xFFFFFFFF = none */
  float localPos[3];
  float localRot[3];
  float localScale[3];
  uint32_t _pad0[2];
  float worldMatrix[16];                /* offset 48, column-major m[col*4+row] */
  uint8_t dirty;                        /* offset 112 */
  uint8_t _pad1[15];
} OrcTransform;

typedef struct OrcCamera {              /* sc_ecs.h:98-105 */
  float fovY, nearZ, farZ, aspect;
  uint8_t active;
} OrcCamera;

typedef struct OrcRenderMesh { uint32_t meshId, materialId; } OrcRenderMesh;   /* sc_ecs.h:107-111 */
typedef struct OrcBounds { float min[3]; float max[3]; } OrcBounds;            /* sc_world_partition.h:27-31,298-301 */
typedef struct OrcDrawItem {            /* sc_ecs.h:159-165, sizeof == 80 */
  uint32_t entity, meshId, materialId, _pad;
  float model[16];
} OrcDrawItem;

typedef struct OrcWorld OrcWorld;

OrcWorld* orc_world_new(void);
void      orc_world_free(OrcWorld* w);
uint32_t  orc_entity_create(OrcWorld* w);                 /* sc_ecs.cpp:11-26 */
int       orc_entity_destroy(OrcWorld* w, uint32_t e);    /* sc_ecs.cpp:28-42, 80-90 */
int       orc_entity_alive(const OrcWorld* w, uint32_t e);/* sc_ecs.cpp:44-50 */

OrcTransform*  orc_add_transform(OrcWorld* w, uint32_t e);
OrcTransform*  orc_get_transform(OrcWorld* w, uint32_t e);
OrcCamera*     orc_add_camera(OrcWorld* w, uint32_t e);
OrcRenderMesh* orc_add_render_mesh(OrcWorld* w, uint32_t e);
OrcBounds*     orc_add_bounds(OrcWorld* w, uint32_t e);
int            orc_has_bounds(const OrcWorld* w, uint32_t e);
int            orc_has_render_mesh(const OrcWorld* w, uint32_t e);

uint32_t orc_transform_count(const OrcWorld* w);
/* dense-order views of the Transform pool */
const uint32_t* orc_transform_dense_entities(const OrcWorld* w);
OrcTransform*   orc_transform_dense_data(OrcWorld* w);

/* bulk construction from flat arrays (entity i gets index i, generation 0; parent_index -1 = none) */
int orc_world_build(OrcWorld* w, uint32_t n,
                    const float* pos3, const float* rot3, const float* scale3,
                    const int32_t* parent_index,
                    const uint8_t* has_mesh, const uint32_t* mesh_id, const uint32_t* material_id,
                    const uint8_t* has_bounds, const float* bmin3, const float* bmax3);
/* bulk mutators (sc_ecs.h:73-96 semantics: every setter marks dirty) */
void orc_set_local_positions(OrcWorld* w, uint32_t n, const uint32_t* entities, const float* pos3);
void orc_nudge_roots_x(OrcWorld* w, float dx);   /* SynthWorld dirty regime (ii): localPos.x += dx on every root */
void orc_mark_dirty(OrcWorld* w, uint32_t n, const uint32_t* entities);
/* bulk read-back in Transform-pool dense order */
void orc_read_world_matrices(OrcWorld* w, float* out16n);
void orc_read_dirty(OrcWorld* w, uint8_t* outn);
void orc_read_parents(OrcWorld* w, uint32_t* outn);
void orc_read_local_scales(OrcWorld* w, float* out3n);

/* ---- JobSystem::Dispatch / Wait as used by culling (sc_jobs.h:70-134, sc_jobs.cpp:12-101, :202-218, :247-372):
 *      per-worker 1024-slot rings, round-robin enqueue with linear fallback, inline execution when every ring is full,
 *      own-ring-then-steal workers, a waiting caller that helps ---- */
int  orc_jobs_init(uint32_t workers);   /* workers = threads besides the caller; 0 = no job system (ranges run in order on the caller) */
void orc_jobs_shutdown(void);
uint32_t orc_jobs_workers(void);
unsigned long orc_jobs_ran_inline(void); /* jobs the dispatcher ran itself because every ring was full, since start */

/* ---- systems ---- */
void orc_transform_system(OrcWorld* w);                                    /* sc_ecs.cpp:118-211 */

typedef struct OrcCameraState {          /* sc_ecs.h:445-450 + RenderFrameData::viewProj */
  float viewProj[16];
  uint32_t activeCamera;
  float aspect;
} OrcCameraState;
void orc_camera_system(OrcWorld* w, OrcCameraState* st);                   /* sc_ecs.cpp:213-272 */

typedef struct OrcPlane { float n[3]; float d; } OrcPlane;
typedef struct OrcFrustum { OrcPlane planes[6]; uint8_t valid; } OrcFrustum; /* sc_world_partition.h:33-43 */
void orc_frustum_from_viewproj(const float viewProj[16], OrcFrustum* out); /* sc_world_partition.cpp:1071-1103 */
int  orc_sphere_in_frustum(const OrcFrustum* f, const float center[3], float radius); /* :1105-1117 */
void orc_world_bounds_sphere(const float worldMatrix[16], const OrcBounds* b,
                             float outCenter[3], float* outRadius);        /* :1119-1144 */

typedef struct OrcCullingState {         /* sc_world_partition.h:334-351 */
  int freezeCulling;
  OrcFrustum frustum;
  uint32_t renderablesTotal, visibleCount, culledCount;
  uint32_t* candidates; uint32_t candidatesLen, candidatesCap;
  uint32_t* visible;    uint32_t visibleLen,    visibleCap;
  uint32_t* culled;     uint32_t culledLen,     culledCap;
  uint8_t*  visibilityMask; uint32_t maskLen;
} OrcCullingState;
OrcCullingState* orc_culling_state_new(void);
void orc_culling_state_free(OrcCullingState* s);
void orc_culling_system(OrcWorld* w, OrcCullingState* s, const float viewProj[16]); /* :1199-1284 */

/* RenderPrepStreamingSystem, draw emission only (:1286-1359); returns emitted, *dropped out */
uint32_t orc_render_prep_streaming(OrcWorld* w, const OrcCullingState* s, uint32_t maxDraws,
                                   OrcDrawItem* out, uint32_t outCap, uint32_t* dropped);

/* the renderer's filter + sort of RenderFrameData::draws (src/engine/src/sc_vk.cpp:1842-1864), stable; UNPINNED like
 * the other systems (sc_vk.cpp needs Vulkan); order[] receives indices into items, returns the number kept */
uint32_t orc_renderer_draw_order(const OrcDrawItem* items, uint32_t n, const uint8_t* pipelineOfMaterial,
                                 uint32_t materialCount, uint32_t meshCount, uint32_t* order);

/* ---- sector binning: sc_world_partition.cpp:268-275 ---- */
void orc_world_to_sector(float sectorSize, float x, float z, int32_t* sx, int32_t* sz);

/* ---- broadphase (own spec, DESIGN.md "Broadphase spec") ---- */
/* world AABB of Bounds under worldMatrix: centre = M*c, half = |M3x3|*e; min = c-h, max = c+h */
void orc_world_aabb(const float worldMatrix[16], const OrcBounds* b, float outMin[3], float outMax[3]);
/* AABBs for all Transform-pool entities (dense order); entities without Bounds get min=+inf,max=-inf (never overlap) */
void orc_read_world_aabbs(OrcWorld* w, float* min3n, float* max3n);
/* brute force O(n^2): pairs (i<j), closed-interval overlap on 3 axes && (gi&mj)&&(gj&mi); returns total count,
   writes at most cap pairs (i,j) in lexicographic order */
uint64_t orc_broadphase_bruteforce(uint32_t n, const float* min3, const float* max3,
                                   const uint32_t* group, const uint32_t* mask,
                                   uint32_t* pairs2, uint64_t cap);
/* uniform-grid restatement (cell size cellSize, cells keyed like worldToSector on the AABB min/max);
   same output contract, lexicographically sorted */
uint64_t orc_broadphase_grid(uint32_t n, const float* min3, const float* max3,
                             const uint32_t* group, const uint32_t* mask, float cellSize,
                             uint32_t* pairs2, uint64_t cap);

/* ---- ray queries (own spec; see sc_oracle.c): brute force over all boxes, group/mask as uploaded (low 16 bits) ---- */
typedef struct OrcRayHit { uint32_t hit, id; float distance; float position[3]; float normal[3]; uint32_t layer; uint32_t pad[2]; } OrcRayHit;
int  orc_ray_box_probe(const float origin[3], const float dir[3], float tmax, const float mn[3], const float mx[3], float* tOut);
void orc_raycast_boxes(uint32_t n, const float* min3, const float* max3, const uint32_t* group, const uint32_t* mask,
                       uint32_t rays, const float* origin3, const float* dir3, const float* maxDist, const uint32_t* rayMask,
                       OrcRayHit* out);

/* isOccupiedWorld (sc_traffic_spawner.cpp:93-116); isAgent flags the entities of the TrafficAgent / VehicleComponent pools */
int orc_is_occupied(OrcWorld* w, const uint8_t* isAgent, const float pos[3], float radius);

/* ---- upstream movers (own spec, include/sc_tick.h "upstream movers"; model after the on-rails tier,
 *      sc_traffic_ai.cpp:434-460): dense-order arrays; vel is updated in place for reflecting peds ---- */
void orc_advance_movers(OrcWorld* w, const uint8_t* kind, float* vel_xz2, const float* lo_xz2, const float* hi_xz2, float dt);

/* ---- the step before the path, sc_oracle_traffic.c: lane graph, on-rails traffic advance, tier selection.
 *      PARITY UNPINNED (no reference test or fixture; the sources need Vulkan headers to compile). ---- */
typedef struct OrcLaneGraph OrcLaneGraph;
OrcLaneGraph* orc_lanes_new(void);
void     orc_lanes_free(OrcLaneGraph* g);
void     orc_lanes_build_sector(OrcLaneGraph* g, int32_t cx, int32_t cz, float sectorSize, uint32_t outSegs[4]);  /* sc_traffic_lanes.cpp:158-225 */
void     orc_lanes_set_active(OrcLaneGraph* g, uint32_t seg, int active);                                        /* :227-237 */
uint32_t orc_lanes_segment_count(const OrcLaneGraph* g);
uint32_t orc_lanes_node_count(const OrcLaneGraph* g);
uint32_t orc_lanes_connection_count(const OrcLaneGraph* g);
void     orc_lanes_export(const OrcLaneGraph* g, float* segStart3, float* segDir3, float* segLength, uint8_t* segActive,
                          uint32_t* segEndNode, float* segSpeedLimit, float* nodePos3, uint32_t* nodeConnOffset, uint32_t* nodeConn);
int      orc_lanes_advance(const OrcLaneGraph* g, uint32_t* laneId, float* s, float distance, float outPos[3], float outDir[3]); /* :291-352 */
/* dense-order arrays; laneId / laneS / targetSpeed are updated in place (sc_traffic_ai.cpp:264-299, :434-460) */
void     orc_traffic_ai_onrails(OrcWorld* w, const OrcLaneGraph* g, const uint8_t* isAgent, uint32_t* laneId, float* laneS,
                                float* targetSpeed, const uint8_t* mode, const float* lookAheadDist, float speedMultiplier, float dt);
/* the same with the obstacle brake of every agent (sc_traffic_ai.cpp:300-345, :436; NULL = none) */
void     orc_traffic_ai_onrails_braked(OrcWorld* w, const OrcLaneGraph* g, const uint8_t* isAgent, uint32_t* laneId, float* laneS,
                                       float* targetSpeed, const uint8_t* mode, const float* lookAheadDist, const float* obstacleBrake,
                                       float speedMultiplier, float dt);
int      orc_lanes_query_nearest(const OrcLaneGraph* g, const float pos[3], uint32_t* laneOut, float* sOut);   /* sc_traffic_lanes.cpp:240-279 */
/* one agent's front ray against the world AABBs (own spec) and the brake it yields, sc_traffic_ai.cpp:300-345 */
float    orc_traffic_front_ray_brake(uint32_t n, const float* min3, const float* max3, const uint32_t* group, const uint32_t* mask,
                                     uint32_t self, const float pos[3], float sinYaw, float cosYaw, float rayLen, float safe);
void     orc_traffic_front_ray_sensors(OrcWorld* w, const float* min3, const float* max3, const uint32_t* group, const uint32_t* mask,
                                       const uint8_t* isAgent, const uint8_t* mode, const uint8_t* isVehicle, const float* rayLen, const float* safe,
                                       float* brakeOut, float* hitDistOut, uint8_t* hitTypeOut);
void     orc_traffic_front_ray_brakes(OrcWorld* w, const float* min3, const float* max3, const uint32_t* group, const uint32_t* mask,
                                      const uint8_t* isAgent, const uint8_t* mode, float rayLen, float safe, float* brakeOut);
/* mode / desired: 0 Physics, 1 Kinematic, 2 OnRails (sc_traffic_common.h:11-16); counts[3] after the caps (sc_traffic_lod.cpp:323-417) */
void     orc_traffic_lod_tiers(OrcWorld* w, const uint8_t* isAgent, const uint8_t* mode, const float playerPos[3],
                               float tierAEnter, float tierAExit, float tierBEnter, float tierBExit,
                               uint32_t maxPhysics, uint32_t maxKinematic, uint8_t* desiredOut, uint32_t counts[3]);

/* the total cap behind it (sc_traffic_lod.cpp:419-465): dense indices flagged for despawning, in flagging order; returns their number */
uint32_t orc_traffic_lod_despawns(OrcWorld* w, const uint8_t* isAgent, const uint8_t* mode, const float playerPos[3], uint32_t maxTotal, uint32_t* outIdx);

/* ---- whole-tick convenience for the cpu_baseline leg: Transform + Camera + Culling ---- */
void orc_tick(OrcWorld* w, OrcCameraState* cam, OrcCullingState* cull);

#ifdef __cplusplus
}
#endif
#endif
