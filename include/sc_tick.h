/*
 * sc_tick.h -- C ABI of the MI355X world-tick library (libsc_tick.so).
 *
 * One context owns the device-resident SoA state of one world tile on one GPU and runs the
 * RenderPrep hot path of SandboxCityEngine on it:
 *
 *   TransformSystem            src/core/src/sc_ecs.cpp:118-211
 *   (CameraSystem stays on the host: sc_ecs.cpp:213-272; its product, viewProj, is an input here)
 *   CullingSystem              src/engine/world/sc_world_partition.cpp:1199-1284
 *   RenderPrepStreamingSystem  src/engine/world/sc_world_partition.cpp:1286-1359 (draw emission)
 *   AABB broadphase            Bullet btDbvtBroadphase behind src/engine/physics/sc_physics.cpp:218-225, :289, :296-299
 *
 * The reference has no FFI on this path: systems are C++ functions `void(World&, float, void*)`
 * (src/core/include/sc_scheduler.h:38).  This ABI is what the C++ adapter systems in
 * sc_gameengine_amd/host/sc_tick_systems.cpp (same signature, same state structs) call; a cgo /
 * JNI / ctypes binding would bind exactly these symbols.  Conventions follow the reference's own
 * C ABI, src/engine/include/sc_engine_render.h:130-163 and src/engine/src/sc_engine_render.cpp:88-177:
 * extern "C", opaque context from paired create/destroy, `int` 1 = ok / 0 = failed, every entry
 * point tolerates NULL, POD structs with fixed-size arrays, the caller owns every buffer it passes
 * and the callee copies, and a GetApiVersion().
 *
 * Entities are addressed by their DENSE index in the Transform pool (ComponentPool<Transform>::
 * denseEntities order, src/core/include/sc_ecs.h:199-277): that order defines the order of the
 * candidate / visible / culled lists (sc_world_partition.cpp:1206-1210, :1273-1280).
 *
 * Threading: a context may be used from any thread, one call at a time (systems may run on any job
 * worker: src/core/src/sc_scheduler.cpp:117-126); every call binds the device itself.
 * All device work is queued on the context's own stream; calls that return data synchronise it.
 */
#ifndef SC_TICK_H
#define SC_TICK_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SC_TICK_API_VERSION 7u
#define SC_TICK_MAX_ENTITIES ((1u << 24) - 1u)   /* Entity::INDEX_BITS = 24 (sc_ecs.h:18-20); index 0xFFFFFF is the "no parent" value */
#define SC_TICK_NO_PARENT (-1)

typedef struct ScTickContext ScTickContext;

/* scTickRun flags: which stages of the tick to execute */
enum {
  SC_TICK_XFORM       = 1u << 0,   /* TransformSystem */
  SC_TICK_CULL        = 1u << 1,   /* CullingSystem: visibility bits + ordered visible list */
  SC_TICK_BROADPHASE  = 1u << 2,   /* world AABBs + grid pair search */
  SC_TICK_CULLED_LIST = 1u << 3,   /* also build CullingState::culled */
  SC_TICK_DRAWS       = 1u << 4,   /* RenderPrepStreamingSystem draw list from the visible list */
  SC_TICK_DENSE_AABBS = 1u << 5,   /* with BROADPHASE: also keep per-entity world AABBs for scTickReadWorldAabbs */
  SC_TICK_SPLIT_PAIRS = 1u << 6,   /* with BROADPHASE on a multi-GPU tile: stop after filling the bins and packing the
                                      border messages; the caller exchanges them and calls scTickRunPairs */
  SC_TICK_SORT_DRAWS  = 1u << 7,   /* with DRAWS: the list comes out in the renderer's bind order (scTickSetDrawSortTable) */
  SC_TICK_RAYS        = 1u << 8,   /* with BROADPHASE: answer the ray queries set by scTickSetRayQueries against this tick's boxes */
  SC_TICK_PRODUCE_NEXT = 1u << 9,  /* with XFORM: the frame producer (scTickSetFrameProducer) is applied at the END of this run, inside
                                      the end-of-tick kernel, as the producer of the NEXT frame -- instead of at the start of the next
                                      scTickRun.  Results of this run are unaffected; positions read back are already the next frame's. */
  SC_TICK_FULL        = SC_TICK_XFORM | SC_TICK_CULL | SC_TICK_BROADPHASE
};

typedef struct ScTickContextDesc
{
  int32_t  device_ordinal;     /* HIP device */
  uint32_t capacity;           /* max entities in this context, <= SC_TICK_MAX_ENTITIES */
  /* broadphase grid: the rectangle of sectors this context's tile covers, in worldToSector
   * coordinates (sc_world_partition.cpp:268-275).  sectors_x == 0 disables the broadphase. */
  int32_t  tile_origin_x;
  int32_t  tile_origin_z;
  uint32_t tile_sectors_x;
  uint32_t tile_sectors_z;
  float    sector_size;        /* WorldPartitionConfig::sectorSizeMeters, 64 m (sc_world_partition.h:151) */
  uint32_t max_pairs;          /* capacity of the pair list (0 = capacity entities * 4) */
  uint32_t max_draws_budget;   /* WorldStreamingBudgets::maxDrawsBudget (sc_world_partition.h:309); 0 = unlimited */
  uint32_t reserved;
} ScTickContextDesc;

typedef struct ScTickCounts    /* CullingStats (sc_world_partition.h:334-339) + pair / draw counts */
{
  uint32_t entities;
  uint32_t renderables_total;
  uint32_t visible;
  uint32_t culled;
  uint32_t pairs;              /* pairs found (may exceed the pair capacity; the list is then truncated) */
  uint32_t pairs_truncated;    /* 1 if pairs > capacity */
  uint32_t draws_emitted;      /* RenderPrepStats (sc_world_partition.h:353-357) */
  uint32_t draws_dropped;
  uint32_t max_depth;          /* deepest hierarchy level after scTickSetTopology */
  uint32_t unreachable;        /* entities in or below a parent cycle (never updated, sc_ecs.cpp:173-210) */
  uint32_t bin_overflow;       /* bin records that found their sector's bin (64 records) full: they sit in the sector overflow list */
  uint32_t big_boxes;          /* boxes in the big list: larger than 2x2 sectors or outside the tile rectangle */
  uint32_t draws_sorted;       /* with SC_TICK_SORT_DRAWS: draws left after the renderer's mesh / material handle checks */
  uint32_t border_lost;        /* records the fixed capacities could not carry: border messages that ran out of room
                                  (scTickSetBorderCapacity), big boxes reaching beyond the eight neighbouring tiles, a sector with more
                                  than 64 + 1024 boxes; non-zero means pairs may be missing */
  uint32_t relinks;            /* whole-world hierarchy re-links (O(entities) on the host) this context has done so far */
  uint32_t vocabulary_violations;  /* with scTickSetWorldLayers(known): border records and big boxes that ARRIVED from a neighbour with group or
                                      mask bits outside the declared vocabulary this tick -- the neighbour broke the contract, pairs in bins
                                      this tile leaves unwritten may be missing; 0 otherwise.  (This tile's own layers cannot leave the
                                      vocabulary: scTickUploadLayers / scTickAppendEntities refuse them.) */
} ScTickCounts;

typedef struct ScTickDrawItem  /* DrawItem, sc_ecs.h:159-165: 80 bytes, model at offset 16, column-major */
{
  uint32_t dense_index;        /* the host adapter maps it to the Entity handle */
  uint32_t mesh_id;
  uint32_t material_id;
  uint32_t pad;
  float    model[16];
} ScTickDrawItem;

/* kernels whose per-launch durations scTickGetKernelTimes reports */
enum { SC_TICK_K_XFORM_CULL = 0, SC_TICK_K_COMPACT = 1, SC_TICK_K_PAIRS = 2, SC_TICK_K_NUDGE = 3 /* producer: nudge or movers */, SC_TICK_K_COUNT = 4 };

uint32_t       scTickGetApiVersion(void);
ScTickContext* scTickCreateContext(const ScTickContextDesc* desc);
void           scTickDestroyContext(ScTickContext* ctx);
/* text of the last failure on this context (or of the last failed create when ctx is NULL) */
const char*    scTickGetLastError(const ScTickContext* ctx);

/* ---- entity state upload (host -> device SoA).  [first, first+count) are dense indices. ---- */
int scTickSetEntityCount(ScTickContext* ctx, uint32_t count);
/* setLocal (sc_ecs.h:78-84): position, XYZ Euler radians, scale; marks the range dirty.  sin/cos of
 * the angles are taken here with the host libm, as mat4_rotation_xyz does (sc_math.cpp:102-107), so
 * device matrices equal the host's bit for bit.  An all-zero scale is stored as (1,1,1), the repair
 * TransformSystem applies (sc_ecs.cpp:143-149); repaired[i] (nullable) reports it. */
int scTickUploadLocals(ScTickContext* ctx, uint32_t first, uint32_t count,
                       const float* pos3, const float* rot3, const float* scale3, uint8_t* repaired);
/* setLocalPosition (sc_ecs.h:92-96) */
int scTickUploadPositions(ScTickContext* ctx, uint32_t first, uint32_t count, const float* pos3);
/* Bounds::localAabb (sc_world_partition.h:298-301); has_bounds NULL = every entity has one */
int scTickUploadBounds(ScTickContext* ctx, uint32_t first, uint32_t count,
                       const float* min3, const float* max3, const uint8_t* has_bounds);
/* RenderMesh (sc_ecs.h:107-111); has_mesh NULL = every entity is a culling candidate */
int scTickUploadRenderMeshes(ScTickContext* ctx, uint32_t first, uint32_t count,
                             const uint8_t* has_mesh, const uint32_t* mesh_id, const uint32_t* material_id);
/* collision filter group / mask (sc_physics.cpp:372-379); low 16 bits are kept, 0xFFFFFFFF = all */
int scTickUploadLayers(ScTickContext* ctx, uint32_t first, uint32_t count,
                       const uint32_t* group, const uint32_t* mask);
/* Transform::parent for every entity as a dense index (SC_TICK_NO_PARENT = root).  A parent that is
 * out of range or the entity itself is detached and the entity marked dirty (sc_ecs.cpp:151-160).
 * Computes hierarchy depth; entities in or below a parent cycle are flagged unreachable. */
int scTickSetTopology(ScTickContext* ctx, const int32_t* parent_dense_index, uint32_t count);
/* markDirty (sc_ecs.h:73-76) */
int scTickMarkDirty(ScTickContext* ctx, uint32_t first, uint32_t count);
int scTickMarkDirtyIndices(ScTickContext* ctx, const uint32_t* dense_indices, uint32_t count);
/* set Transform::dirty of a range to exactly these values (1 = dirty); used when a host ECS re-syncs
 * its whole state after the pool's dense order changed (sc_ecs.h:240-262 swap-remove) */
int scTickSetDirtyFlags(ScTickContext* ctx, uint32_t first, uint32_t count, const uint8_t* dirty);
/* seed Transform::worldMatrix (column-major Mat4, must be affine: row 3 == 0,0,0,1) */
int scTickUploadWorldMatrices(ScTickContext* ctx, uint32_t first, uint32_t count, const float* mat16);

/* ---- per-frame inputs ---- */
/* RenderFrameData::viewProj (sc_ecs.h:167-173); the six planes are derived as frustumFromViewProj
 * does (sc_world_partition.cpp:1071-1103) */
int scTickSetViewProj(ScTickContext* ctx, const float view_proj[16]);
/* or the planes directly: 6 x (nx, ny, nz, d), and Frustum::valid */
int scTickSetFrustumPlanes(ScTickContext* ctx, const float planes24[24], int valid);
int scTickGetFrustumPlanes(ScTickContext* ctx, float planes24[24], int* valid);
/* CullingState::freezeCulling (sc_world_partition.cpp:1227-1233) */
int scTickSetFreezeCulling(ScTickContext* ctx, int freeze);
/* WorldStreamingBudgets::maxDrawsBudget for the next SC_TICK_DRAWS (overrides the create-time value; 0 = unlimited) */
int scTickSetDrawBudget(ScTickContext* ctx, uint32_t max_draws);

/* ---- the tick ---- */
int scTickRun(ScTickContext* ctx, uint32_t flags);       /* queues the stages; returns at once */
int scTickSynchronize(ScTickContext* ctx);
/* upstream producer of SynthWorld's dirty regime (ii): localPos.x += dx on every root, marked dirty
 * (the device-side analogue of PhysicsSyncSystem's transform writes, sc_physics.cpp:1167-1186) */
int scTickNudgeRootsX(ScTickContext* ctx, float dx);

/* ---- multi-GPU tiles (one context per GPU, one process per GPU) ----
 * The world is cut into equal rectangular tiles of sectors; each context owns one (ScTickContextDesc
 * tile_*).  Transform and culling need no exchange.  The broadphase needs the boxes that reach over a
 * tile edge: after scTickRun(... | SC_TICK_BROADPHASE | SC_TICK_SPLIT_PAIRS) the message for each
 * existing neighbour sits in the bound send buffer; the caller moves send[d] of this rank into
 * recv[7-d] of the neighbour in direction d (RCCL send/recv, or a peer copy) on the context's stream
 * and then calls scTickRunPairs.  Directions: d = 0..7 for (dx,dz) = (-1,-1) (0,-1) (1,-1) (-1,0)
 * (1,0) (-1,1) (0,1) (1,1).  Pair ids are rank << 24 | dense index. */
int scTickSetTile(ScTickContext* ctx, uint32_t rank, uint32_t neighbour_mask);
/* This tile's place (tile_x, tile_z) in the grid of tiles_x x tiles_z equal tiles; sets the neighbour mask from it.
 * With the grid known the border messages also carry the big boxes (wider than 2x2 sectors or outside the tile's
 * rectangle) that reach a neighbour's region, so pairs with them are found across tile
 * borders too: a tile reports a box-vs-big pair when it owns the sector of the box's primary copy, and a big-vs-big
 * pair when it owns the sector holding the low corner of the intersection (sectors outside the world belong to the
 * nearest tile).  A big box may reach its own tile and the eight around it; one that reaches further, or a message
 * that runs out of room, is counted in ScTickCounts::border_lost. */
int scTickSetTileGrid(ScTickContext* ctx, uint32_t tile_x, uint32_t tile_z, uint32_t tiles_x, uint32_t tiles_z);
/* Capacity of the border messages: `records_per_ring_sector` records per sector of a ring side ON AVERAGE (the side's
 * sectors share the message: a crowded ring sector may take more than its share), never less than one full sector (its
 * 64-record bin + 1024 overflow records).  Default 16 -- SynthWorld's ring sectors hold 0-2.  A world authored at the engine's
 * streaming budget (200 entities per sector, src/sandbox/src/main.cpp:92-99) with districts on tile edges wants more; what a
 * message cannot hold is counted in ScTickCounts::border_lost, never dropped silently.  Every tile of a world must use the
 * same value (the two sides of an edge agree on the message size).  Call before scTickCommInit / before binding buffers:
 * bound buffers are unbound, scTickBorderBytes changes. */
int scTickSetBorderCapacity(ScTickContext* ctx, uint32_t records_per_ring_sector);
/* size in bytes of the fixed-capacity border message of direction d (same on both sides of an edge) */
uint32_t scTickBorderBytes(ScTickContext* ctx, uint32_t direction);
/* caller-owned device buffers (e.g. torch tensors) of at least scTickBorderBytes(d) bytes each */
int scTickBindBorderBuffers(ScTickContext* ctx, uint32_t direction, void* send_device_ptr, void* recv_device_ptr);
int scTickRunPairs(ScTickContext* ctx);
/* Pipelined tiles.  With a pairs stream set (hipStream_t; NULL switches it off), scTickRunPairs queues the merge, the ray
 * queries and the pair search of tick t on THAT stream, and the next scTickRun may start its fused kernel while they run:
 * everything the two halves share -- bins, big list, spill list, border messages -- exists `depth` times (3 unless
 * scTickSetPipelined chose otherwise, at most 4), selected by tick parity t mod depth (counters, big-box bits and the pair
 * output already are).  The library orders the halves with events: scTickRun makes the pairs stream wait for the pack of its
 * tick, and tick t+depth-1, whose end-of-tick kernel clears the counters tick t+depth fills again, first waits for the pair half
 * of tick t.  With depth d the pair half (exchange latency included) may take up to d-2 ticks before it holds anything up;
 * until that clearing, the parity keeps tick t's results (counts, pair shard counters) for the host to read.  The caller issues the exchange
 * of tick t on the pairs stream after scTickRun (border buffers of parity t mod depth, counted from the moment the pairs stream
 * was set: scTickBindBorderBuffersParity), then calls scTickRunPairs -- e.g. the pairs
 * stream is torch's current stream, where its RCCL operations go, and the tick runs on the context's own stream.
 * Read the results of tick t (pairs, ray hits, counts) after its scTickRunPairs and before the next scTickRun, as always.
 * With graph replay on (scTickSetGraphMode) each half of a pipelined step is a graph of its own, on its own stream.  Switching the pairs stream on or off synchronises and clears the per-parity broadphase
 * state (the two flows clear it differently): the previous tick's pairs / counts are no longer readable afterwards. */
int scTickSetPairsStream(ScTickContext* ctx, void* hip_stream);
int scTickBindBorderBuffersParity(ScTickContext* ctx, uint32_t parity, uint32_t direction, void* send_device_ptr, void* recv_device_ptr);
/* the device buffer bound for (tick parity, direction): recv == 0 the outgoing message, != 0 the incoming one; NULL = none.
 * Parity 0 is also the in-order flows' only set. */
void* scTickGetBorderBuffer(ScTickContext* ctx, uint32_t parity, uint32_t direction, int recv);
/* ---- the exchange itself, owned by the library (north_star: "Host code stays C++ ... RCCL over xGMI exchanging only
 * tile-border AABBs").  The reference has no counterpart: it is a single process ("Distributed
 * communication backend: none"); the tile sharding is this build's, its unit is the sector grid of
 * src/engine/world/sc_world_partition.cpp:268-287.
 * One RCCL communicator per context = per GPU = per process.  Rank 0 asks for a unique id, the host hands its 128 bytes
 * to every rank over whatever channel it has (a file, a socket, MPI, torch.distributed: bench.py), and every rank calls
 * scTickCommInit after scTickSetTileGrid.  The neighbour in direction d is the rank of tile (tile_x+dx, tile_z+dz) in
 * row-major tile order unless scTickCommSetPeers says otherwise (-1 = none).  The library then owns the border message
 * buffers of both tick parities (RCCL is opened with dlopen on first use; a single-GPU host never needs it).
 * scTickTileStep(flags) is one whole step of a tile in one call: scTickRun(flags | SC_TICK_SPLIT_PAIRS), the exchange as
 * ONE group of ncclSend / ncclRecv on the stream the pair half runs on (the pairs stream when pipelined), scTickRunPairs.
 * Nothing in it waits on the host.  A tile without neighbours (1x1 grid) runs scTickRun(flags).  A tile WITH neighbours
 * and no communicator fails: the exchange is never skipped silently.
 * scTickSetPipelined(n) = scTickSetPairsStream with a second stream of the library's own: n = 0 off, 1 = on with the default
 * depth (4 copies of the per-tick broadphase state), 2..4 = on with that depth (2 leaves the pair half no tick to hide under). */
#define SC_TICK_COMM_ID_BYTES 128
int scTickCommGetUniqueId(uint8_t id[SC_TICK_COMM_ID_BYTES]);
int scTickCommInit(ScTickContext* ctx, const uint8_t id[SC_TICK_COMM_ID_BYTES], uint32_t world_size, uint32_t rank);
int scTickCommSetPeers(ScTickContext* ctx, const int32_t peer_rank[8]);
int scTickCommDestroy(ScTickContext* ctx);
int scTickSetPipelined(ScTickContext* ctx, int enable);
/* What a multi-GPU run needs in order to be read afterwards: the communicator as ncclCommInitRank saw it, the exchange's
 * shape, and the host time scTickTileStep spends issuing each half of a step (averages since the last reset; an in-order
 * captured step counts as one tick half).  Valid without a communicator too (a 1x1 grid: zeros). */
typedef struct ScTickCommInfo
{
  uint32_t has_communicator, world_size, rank;
  uint32_t rccl_version;               /* ncclGetVersion of the library the context bound at run time */
  uint32_t neighbour_mask;
  int32_t  peer_rank[8];               /* -1 = no neighbour in that direction */
  uint32_t operations_per_group;       /* ncclSend + ncclRecv calls inside the one group of a step */
  uint32_t pipeline_depth;             /* 0 = in order */
  uint32_t border_records_per_sector;  /* scTickSetBorderCapacity */
  uint64_t bytes_sent_per_step;        /* fixed-size messages: what the group moves out of this rank per step */
  uint64_t host_steps;
  double   host_tick_half_us, host_pair_half_us;
} ScTickCommInfo;
int scTickGetCommInfo(ScTickContext* ctx, ScTickCommInfo* out);
/* Result assembly at N > 1: the global visible list is the concatenation of the tiles' lists in rank order --
 * entities are created tile-major, so that IS the order of the reference's serial compaction over the whole world
 * (src/engine/world/sc_world_partition.cpp:1273-1280) -- and each rank copies its slice to the host at its own offset.  This call
 * is the one tiny collective that needs: an all-gather of the ranks' visible counts of the last tick over the context's
 * communicator (ncclAllGather of one uint32 per rank; every rank must call it, after the same tick).  counts_out (may be NULL)
 * receives min(world_size, capacity) counts; *offset_out = the sum of the counts of the ranks before this one = where this rank's
 * scTickReadVisible slice starts in the global list (its ids are dense indices of the tile: add rank * entities-per-tile for
 * global dense indices); *total_out = the length of the global list.  Without a communicator (a 1x1 grid) the tile is the world:
 * offset 0, total = its own count.  Synchronises the context (a read-back call, like scTickReadVisible). */
int scTickGatherVisibleCounts(ScTickContext* ctx, uint32_t* counts_out, uint32_t capacity, uint64_t* offset_out, uint64_t* total_out);

/* Broadphase bins, how they are filled (diagnostics; the pair set never depends on any of it).  Records keep the bin slot they
 * reserved at the last "learn" tick while their box stays in its sector (no reservation, i.e. no atomic, on the ticks in
 * between; SC_TICK_HOME_PERIOD ticks apart, default 64; SC_TICK_VARIANT bit 1 switches the slots off), and the slots of a bin
 * whose own records cannot pass the group/mask filter against each other -- static props only -- are not even written until a
 * record from elsewhere needs them (the pair search then rebuilds them; SC_TICK_VARIANT bit 5 switches that off; ticks with
 * ray queries or traffic sensors, and pipelined tiles, write every record).
 * stats[0] remembered slots, [1] of those written on every tick, [2] bit 0: the last tick was allowed to leave the other slots
 * unwritten, bit 1: it left records of entities whose matrix was not rebuilt as they were, bit 2: no two of the uploaded layer words
 * (nor, on a tile, of the declared world vocabulary) admit a pair, so the pair role was launched as a sweep over the bins' counters
 * (a quarter of the workgroups; a launch shape, never a shortcut of the search), [3] learn ticks so far.  Reads the slots back (a few MB): not for the frame loop. */
int scTickGetBinStats(ScTickContext* ctx, uint32_t stats[4]);
/* stats[3] of the above alone: learn ticks so far.  Host-side, no read-back, no synchronisation (what a timed loop may ask). */
int scTickGetLearnTicks(ScTickContext* ctx, uint32_t* learn_ticks);
/* The layer VOCABULARY of the tiled world: the OR of the group words and the OR of the mask words of every collider that exists on
 * ANY tile, now or later (until the next call; bits 0..15, or 0xFFFFFFFF = all, as scTickUploadLayers).  With it a pipelined tile
 * (scTickSetPipelined / scTickSetPairsStream) leaves the bins unwritten whose own records can meet nothing the world contains --
 * static props in a world without dynamic bodies, say -- because nothing will ever read them; without it (known = 0, the default)
 * a pipelined tile writes every record on every tick.  A contract, and a checked one: while a vocabulary is declared,
 * scTickUploadLayers / scTickAppendEntities FAIL for a group or mask word outside it (declare the wider vocabulary first), a call
 * that narrows the vocabulary below the layers already uploaded fails too, and a record or big box that arrives from a neighbour
 * with bits outside it is counted in ScTickCounts::vocabulary_violations (bench.py's N > 1 gate requires 0).  In-order flows do not
 * need the declaration (they rebuild unwritten bins on demand). */
int scTickSetWorldLayers(ScTickContext* ctx, uint32_t group_or, uint32_t mask_or, int known);
int scTickResetHostTimes(ScTickContext* ctx);
int scTickTileStep(ScTickContext* ctx, uint32_t flags);
/* the middle third of scTickTileStep on its own, for hosts that interleave other work: after scTickRun(... | SC_TICK_SPLIT_PAIRS) */
int scTickExchangeBorders(ScTickContext* ctx);
/* external != 0: run all device work of this context on the caller's stream `hip_stream` (hipStream_t;
 * NULL is the legacy default stream), e.g. the stream its RCCL calls are ordered on.
 * external == 0: return to the context's own stream. */
int scTickSetStream(ScTickContext* ctx, void* hip_stream, int external);

/* ---- upstream movers (the step before the path) ----
 * The engine's on-rails traffic tier writes Transform::localPos every fixed step and marks it dirty
 * (src/engine/traffic/sc_traffic_ai.cpp:434-460, speed 12 m/s src/engine/traffic/sc_traffic_lanes.h:17).
 * SynthWorld's movers are that model reduced to straight segments inside the agent's sector:
 *   kind 1 (vehicle): pos += vel*dt, wrapping inside [lo, hi) on x and z
 *   kind 2 (ped):     pos += vel*dt, reflecting at lo / hi (the velocity component flips)
 * Only localPos.x / .z change; the moved entities are marked dirty, exactly as setLocalPosition does.
 * These two kinds are SynthWorld's stand-ins (peds have no reference counterpart); the engine's own mover, the on-rails
 * traffic agent, is below ("on-rails traffic") and advances in the same call. */
int scTickUploadMovers(ScTickContext* ctx, uint32_t first, uint32_t count, const uint8_t* kind,
                       const float* vel_xz2, const float* lo_xz2, const float* hi_xz2);
int scTickAdvanceMovers(ScTickContext* ctx, float dt);
int scTickReadMoverVelocities(ScTickContext* ctx, uint32_t first, uint32_t count, float* vel_xz2);
/* Make a producer part of the frame: every scTickRun then starts with it, so the whole frame (producer +
 * tick) is one stream sequence and, in graph mode, one captured hipGraph.  kind 0 = none,
 * 1 = scTickNudgeRootsX(param), 2 = scTickAdvanceMovers(param). */
int scTickSetFrameProducer(ScTickContext* ctx, uint32_t kind, float param);

/* ---- on-rails traffic: the engine's own upstream mover ----
 * TrafficAISystem moves every agent of the OnRails tier along the lane graph each fixed step and writes its Transform
 * (src/engine/traffic/sc_traffic_ai.cpp:264-299 preamble, :434-460 on-rails branch): targetSpeed is smoothed towards the
 * lane's speed limit (smoothExp, :58-62, response 2.5), the agent advances targetSpeed * dt along its lane
 * (TrafficLaneGraph::advanceAlongLane, src/engine/traffic/sc_traffic_lanes.cpp:291-352, crossing into the best-aligned
 * connected segment, :137-156, parking on a dead end), localPos.x/z follow the lane, localRot = (0, atan2(dir.x, dir.z), 0),
 * dirty = true.  An agent without a lane (lane id 0xFFFFFFFF) first takes the nearest active one (:264-272,
 * TrafficLaneGraph::queryNearestLane, sc_traffic_lanes.cpp:240-279).  The obstacle ray and its brake (:300-345) are
 * scTickSetTrafficSensors below; without it obstacleBrake = 0, as in the reference when TrafficAIState::physics is null.
 * The Physics / Kinematic tiers are Bullet's (absent): agents in those modes are left
 * alone, their transforms arrive through the upload calls like any physics-synced body's.
 * The lane graph is handed over flat: per segment the start node's position, the direction, length, end node, active flag
 * and the speed limit of its start node (laneSpeedLimit, sc_traffic_lanes.cpp:392-400); per node its position and the
 * segments that start there (LaneNode::connections, CSR).  sin / cos of each segment's yaw are taken here with the host
 * libm, so an agent's world matrix equals the host's bit for bit: the device never evaluates a trigonometric function.
 * scTickUploadTrafficAgents gives entities a TrafficAgent + TrafficVehicle (sc_traffic_common.h:26-44): lane id
 * (0xFFFFFFFF = none: the next step looks for the nearest lane), laneS, targetSpeed, mode 0 Physics /
 * 1 Kinematic / 2 OnRails, lookAheadDist (NULL = 12).  Agents then move with scTickAdvanceMovers(dt) or as the frame
 * producer kind 2, next to SynthWorld's straight-line movers. */
typedef struct ScTickLaneGraph
{
  uint32_t segments, nodes, connections;
  const float* seg_start3;          /* [segments][3] m_nodes[startNode].pos */
  const float* seg_dir3;            /* [segments][3] LaneSegment::dir (unit) */
  const float* seg_length;          /* [segments] */
  const uint8_t* seg_active;        /* [segments], NULL = all active */
  const uint32_t* seg_end_node;     /* [segments] */
  const float* seg_speed_limit;     /* [segments] m_nodes[startNode].speedLimit */
  const float* node_pos3;           /* [nodes][3] */
  const uint32_t* node_conn_offset; /* [nodes + 1] */
  const uint32_t* node_conn;        /* [connections] segment ids, in LaneNode::connections order */
} ScTickLaneGraph;
int scTickSetLaneGraph(ScTickContext* ctx, const ScTickLaneGraph* graph);
/* TrafficLaneGraph::removeSector / re-activation (sc_traffic_lanes.cpp:164-171, :227-237) */
int scTickSetLaneActive(ScTickContext* ctx, const uint32_t* segment_ids, uint32_t count, int active);
int scTickUploadTrafficAgents(ScTickContext* ctx, uint32_t first, uint32_t count, const uint8_t* is_agent, const uint32_t* lane_id,
                              const float* lane_s, const float* target_speed, const uint8_t* mode, const float* look_ahead_dist);
int scTickReadTrafficAgents(ScTickContext* ctx, uint32_t first, uint32_t count, uint32_t* lane_id, float* lane_s,
                            float* target_speed, uint8_t* mode);
/* The traffic AI's obstacle ray (sc_traffic_ai.cpp:300-345, the branch the reference takes when TrafficAIState::physics is
 * set).  With sensors enabled every scTickRun that includes SC_TICK_BROADPHASE casts, right after the boxes of the tick are
 * binned, one ray per OnRails agent: from 1.7 m ahead of the agent's origin and 0.6 m above it, along
 * normalize(sin(yaw), 0, cos(yaw)) with the yaw's sin / cos as the entity holds them, front_ray_length long (TrafficSensors::
 * frontRayLength, 20 m), mask 1; a hit other than the agent's own box closer than safe_distance (TrafficSensors::safeDistance,
 * 10 m) leaves obstacleBrake = clamp01((safe - d) / safe) for the agent, and the next on-rails step -- scTickAdvanceMovers or
 * the frame producer, SC_TICK_PRODUCE_NEXT included -- scales its desired speed by 1 - obstacleBrake (:436).  As for the ray
 * queries below the candidates are the WORLD AABBS of the broadphase with the collision layers as uploaded (own spec: Bullet
 * is absent; an agent's own box never answers).  Order in a frame: rays from the poses of frame t against the boxes of frame
 * t, then the step to frame t+1 -- what the reference does (the ray sees Bullet's world as the last physics step left it).
 * On a TILED world the reference's ray would see the whole world, so with sensors on (switch them on before the border buffers exist:
 * scTickBindBorderBuffers* / scTickCommInit, on every tile) the border messages carry a halo section -- the sender's core-edge
 * records, a full bin per cell in fixed slots -- that lands in the receiver's ring bins, and an in-order step (scTickTileStep without
 * scTickSetPipelined, or the caller-owned split flow) casts the rays in the PAIR half, behind the merge: an agent within a ray's
 * length (at most one sector) of the tile edge brakes for a vehicle on the neighbour tile, exactly as the whole world's agents would.
 * With the step fused into the end-of-tick kernel (SC_TICK_PRODUCE_NEXT) the next frame is produced in the tick half, BEFORE these
 * rays: the brake then acts one tick later -- the reference's own ordering against Bullet's last step.  A PIPELINED tile keeps
 * the rays in the tick half and sees its own boxes only (its pair half runs under the next tick).  The brake of a run
 * without SC_TICK_BROADPHASE is the last one computed.  front_ray_length / safe_distance here are every agent's values until
 * scTickUploadTrafficSensors gives some agents their own (the reference's per-entity TrafficSensors component falls back to exactly
 * these defaults).  Needs agents (scTickUploadTrafficAgents) and a tile rectangle. */
int scTickSetTrafficSensors(ScTickContext* ctx, int enable, float front_ray_length, float safe_distance);
int scTickReadTrafficBrakes(ScTickContext* ctx, uint32_t first, uint32_t count, float* obstacle_brake);
/* Per-agent TrafficSensors (src/engine/traffic/sc_traffic_common.h:46-53; the AI reads frontRayLength / safeDistance of the agent's own
 * component, sc_traffic_ai.cpp:306-308): scTickSetTrafficSensors gives every entity the two defaults, scTickUploadTrafficSensors
 * overrides them for a range (they travel with their entity through scTickRemoveEntities).  What the AI leaves in the component
 * (:339-345, read by the debug state :420-421, :479-480) comes back through scTickReadTrafficSensors: lastHitDistance -- the hit's
 * distance, or the agent's ray length without a hit -- and lastHitType -- 0 None, 2 Vehicle, 3 World; own spec like the rays:
 * Vehicle = the hit entity is a vehicle by its mover kind (a traffic agent, or a kind-1 mover: what carries a VehicleComponent,
 * :327); a box that arrived from a neighbour tile has no mover kind on this tile and counts as a vehicle when its group has the
 * dynamic bit; Self (1) does not occur because an agent's own box never answers.  Values of agents that are not OnRails, or
 * before the first ray tick, are 0. */
int scTickUploadTrafficSensors(ScTickContext* ctx, uint32_t first, uint32_t count, const float* front_ray_length, const float* safe_distance);
int scTickReadTrafficSensors(ScTickContext* ctx, uint32_t first, uint32_t count, float* last_hit_distance, uint8_t* last_hit_type);
/* TrafficDebugState::speedMultiplier (sc_traffic_ai.cpp:297-298); 1 by default */
int scTickSetTrafficSpeedMultiplier(ScTickContext* ctx, float multiplier);
/* TrafficLODSystem's tier selection (src/engine/traffic/sc_traffic_lod.cpp:269-274 threshold repair, :303-307 xz distance to
 * the player, :323-353 hysteresis, :355-417 the physics / kinematic caps -- candidates sorted by distance, descending, equal
 * distances in pool order, everything past the cap demoted): every agent's TrafficVehicle::mode becomes its desired tier.
 * The total cap (:419-465) is scTickSelectTrafficDespawns below; the despawn itself stays with the caller. */
typedef struct ScTickTierParams     /* TrafficDebugState, sc_traffic_common.h:67-75 */
{
  float tier_a_enter, tier_a_exit, tier_b_enter, tier_b_exit;   /* 50 / 70 / 110 / 150 m */
  uint32_t max_physics, max_kinematic;                          /* 24 / 64; 0 = no cap */
} ScTickTierParams;
typedef struct ScTickTierCounts { uint32_t physics, kinematic, on_rails, total; } ScTickTierCounts;   /* tierPhysics / tierKinematic / tierOnRails / totalVehicles */
int scTickSelectTrafficTiers(ScTickContext* ctx, const float player_pos[3], const ScTickTierParams* params, ScTickTierCounts* out);
/* The total cap behind the tier selection (sc_traffic_lod.cpp:419-465): with more vehicles than max_total
 * (TrafficDebugState::maxTrafficVehiclesTotal; 0 = no cap) the surplus is picked for despawning -- vehicles of the OnRails tier
 * first, then Kinematic, then Physics (the tiers as scTickSelectTrafficTiers left them), the farthest from the player first inside
 * a tier, equal distances in pool order.  *count = how many must go; their dense indices come back in that order (as many as
 * capacity holds).  The despawn itself is the caller's: scTickRemoveEntities. */
int scTickSelectTrafficDespawns(ScTickContext* ctx, const float player_pos[3], uint32_t max_total,
                                uint32_t* dense_indices, uint32_t capacity, uint32_t* count);

/* The renderer's draw order (VkRenderer::recordCommandBuffer, src/engine/src/sc_vk.cpp:1842-1864): draws whose
 * mesh handle is >= mesh_count or whose material handle has no Material are skipped, the rest sorted by
 * (Material::pipelineId, material handle, mesh handle).  pipeline_of_material[h] = pipelineId (< 128; PipelineId has
 * two values, sc_assets.h:22-26) of material handle h, 0xFF = no such material; both counts <= 2^24.  With
 * SC_TICK_DRAWS | SC_TICK_SORT_DRAWS the list scTickReadDraws returns is that sorted list.  Equal keys keep their
 * visible-list order (std::sort leaves it unspecified).  Call again when materials or meshes are created. */
int scTickSetDrawSortTable(ScTickContext* ctx, const uint8_t* pipeline_of_material, uint32_t material_count, uint32_t mesh_count);

/* ---- results (each synchronises the stream) ---- */
int scTickGetCounts(ScTickContext* ctx, ScTickCounts* out);
int scTickReadVisible(ScTickContext* ctx, uint32_t* dense_indices, uint32_t capacity, uint32_t* count);
int scTickReadCulled(ScTickContext* ctx, uint32_t* dense_indices, uint32_t capacity, uint32_t* count);
/* one bit per dense index, bit i of word i/64; set = visible candidate */
int scTickReadVisibilityBits(ScTickContext* ctx, uint64_t* words, uint32_t word_capacity);
int scTickReadWorldMatrices(ScTickContext* ctx, uint32_t first, uint32_t count, float* mat16);
int scTickReadWorldMatricesIndexed(ScTickContext* ctx, const uint32_t* dense_indices, uint32_t count, float* mat16);
int scTickReadDirty(ScTickContext* ctx, uint32_t first, uint32_t count, uint8_t* dirty);
int scTickReadPositions(ScTickContext* ctx, uint32_t first, uint32_t count, float* pos3);
int scTickReadWorldAabbs(ScTickContext* ctx, uint32_t first, uint32_t count, float* min3, float* max3);
/* pairs (a, b), a < b, unordered list; *count = pairs found.  The device keeps the list in 64 equal
 * segments (max_pairs / 64 each); at most that many pairs per segment are kept, pairs_truncated tells. */
int scTickReadPairs(ScTickContext* ctx, uint32_t* pairs2, uint32_t capacity, uint32_t* count);
int scTickReadDraws(ScTickContext* ctx, ScTickDrawItem* items, uint32_t capacity, uint32_t* count);

/* ---- per-frame read-back, overlapped with the next tick ----
 * In resident mode the engine still needs, every frame, what the renderer consumes: the visible list
 * (CullingState::visible, sc_world_partition.h:341-351), the draw items (RenderFrameData::draws, sc_ecs.h:167-173) and
 * the stats.  scTickSetFrameReadback(max_visible, max_draws) makes every scTickRun end with a small staging kernel (the
 * frame's counts, the first max_visible visible indices, the first max_draws draw items when SC_TICK_DRAWS ran, in ONE
 * device block) and ONE device-to-host copy of that block into pinned host memory, on a copy stream of the library's own:
 * the copy of frame t runs under the kernels of tick t+1.  Blocks and host buffers are double-buffered.
 * scTickAcquireFrame(frames_back) waits for the copy of ONE frame -- 0: the most recently queued, 1: the one before, which
 * is how a host overlaps: queue tick t+1, then take frame t -- and for nothing queued after it, and hands out pointers into
 * the pinned buffer; they stay valid until the second scTickRun after the one that produced the frame.
 * (0, 0) switches it off.  Not combinable with graph replay. */
typedef struct ScTickFrame
{
  uint64_t tick;                   /* index of the scTickRun that produced it (counted from the first one after switching on) */
  uint32_t renderables_total, visible, culled;      /* CullingStats */
  uint32_t draws_emitted, draws_dropped, draws_sorted;
  uint32_t visible_in_buffer;      /* min(visible, max_visible) */
  uint32_t draws_in_buffer;        /* min(draws, max_draws); 0 when the run had no SC_TICK_DRAWS */
  const uint32_t* visible_indices;         /* pinned host memory */
  const ScTickDrawItem* draws;             /* pinned host memory */
} ScTickFrame;
int scTickSetFrameReadback(ScTickContext* ctx, uint32_t max_visible, uint32_t max_draws);
int scTickAcquireFrame(ScTickContext* ctx, uint32_t frames_back, ScTickFrame* out);

/* ---- host-side helpers (no GPU work) ----
 * CameraSystem stays on the host (O(#cameras), sc_ecs.cpp:213-272).  These restate the four sc_math
 * functions it and the editor use (sc_math.h:31-58) with the reference's libm calls and operation
 * order, for callers that do not link the engine's own sc_math.  Column-major Mat4, m[c*4+r]. */
int scTickHostMat4Mul(const float a[16], const float b[16], float out[16]);                        /* sc_math.cpp:11-85 */
int scTickHostMat4Trs(const float pos[3], const float rot[3], const float scale[3], float out[16]); /* :130-142 */
int scTickHostMat4Inverse(const float a[16], float out[16]);                                        /* :144-207 */
int scTickHostMat4PerspectiveRhZo(float fov_y_radians, float aspect, float z_near, float z_far,
                                  int flip_y, float out[16]);                                       /* :209-232 */
/* viewProj = perspective(fovY*pi/180, aspect, near, far, flipY) * inverse(cameraWorld), sc_ecs.cpp:261-270 */
int scTickHostCameraViewProj(const float camera_world[16], float fov_y_degrees, float aspect,
                             float z_near, float z_far, float out_view_proj[16]);

/* ---- sector data (.scsector) and residency: the callers either side of the tick ----
 * Host-side reader of the editor/streamer sector format, tools/shared/world_format.cpp:185-338 (ReadSectorFile;
 * record layout as written by WriteSectorFile :76-181), straight into SoA arrays the upload calls take.
 * All versions the reference reads are read the same way: the INST record size is derived from the chunk size,
 * names / texture overrides are present when the record is long enough (:219-230), extra record bytes are
 * skipped, unknown chunks are skipped by their size, zero-sized chunk headers are ignored, and known chunks are
 * parsed by their own counts.  A file that ends early leaves the remaining fields at the reference's defaults
 * (id 0, scale 1, empty name), as its ifstream reads do; info->truncated tells. */
typedef struct ScTickSectorInfo
{
  uint32_t version;                /* SectorFile::version (kSectorVersion = 4, world_format.h:13) */
  int32_t  sector_x, sector_z;     /* SectorFile::sector */
  uint32_t instances;              /* records in the file (may exceed what was stored: see capacity) */
  uint32_t lanes, lane_points, spawners, colliders;
  uint32_t truncated;              /* 1 = the data ended inside a header, count or record */
} ScTickSectorInfo;

typedef struct ScTickSectorInstances
{
  uint32_t capacity;               /* records each non-NULL array below can hold; the first `capacity` are stored */
  uint64_t* id;                    /* Instance::id */
  uint64_t* model_id;              /* 0 for version < 4 */
  uint64_t* mesh_id;               /* asset ids (HashAssetPath), resolved to handles by the caller (:746-792 of sc_world_partition.cpp) */
  uint64_t* material_id;
  uint64_t* albedo_texture_id;     /* 0 unless the record carries overrides */
  uint32_t* material_flags;
  uint32_t* tags;
  float* pos3; float* rot3; float* scale3;     /* Instance::transform, [capacity][3] each: what setLocal takes */
  char* name64;                    /* [capacity][64], NUL-terminated (kInstanceNameMax) */
} ScTickSectorInstances;

/* 0 = not a sector file (bad magic / shorter than the magic) or NULL arguments; `out` may be NULL to only fill info */
int scTickSectorParse(const void* data, uint64_t size, ScTickSectorInfo* info, const ScTickSectorInstances* out);
int scTickSectorReadFile(const char* path, ScTickSectorInfo* info, const ScTickSectorInstances* out);
/* AssetId of a path: FNV-1a 64 (with the reference's non-standard starting value) over the lexically normalised, '/'-separated, lower-cased path (world_format.cpp:52-74); 0 for NULL */
uint64_t scTickHashAssetPath(const char* path);
/* "<root>/sectors/sector_<x>_<z>.scsector" (world_format.cpp:382-389); returns the length needed (excluding NUL) */
uint32_t scTickSectorPath(const char* world_root, int32_t x, int32_t z, char* out, uint32_t capacity);

/* Sector activation (WorldPartition::pumpCompletedLoads, sc_world_partition.cpp:916-958): `count` entities are
 * created at the END of the Transform pool's dense order (ComponentPool::add, sc_ecs.h:203-221) with setLocal
 * (dirty), a RenderMesh, Bounds (NULL min/max = the unit cube kUnitCubeBounds, :27) and collision layers (NULL =
 * group/mask all).  parent = NULL: all roots, nothing else is touched (O(count) work); otherwise dense indices
 * (earlier entities or this batch) and the hierarchy is re-linked.  *first_index = dense index of the first one. */
int scTickAppendEntities(ScTickContext* ctx, uint32_t count, const float* pos3, const float* rot3, const float* scale3,
                         const float* bounds_min3, const float* bounds_max3,
                         const uint32_t* mesh_id, const uint32_t* material_id,
                         const uint32_t* group, const uint32_t* mask, const int32_t* parent, uint32_t* first_index);
/* Despawn (WorldPartition::pumpUnloadQueue -> World::destroy, sc_world_partition.cpp:963-990, sc_ecs.cpp:28-42):
 * `count` distinct entities, named by their dense indices AT THE TIME OF THE CALL, are removed one after the other
 * with the pool's swap-remove (the last entity moves into the hole, sc_ecs.h:240-262), so the dense order afterwards
 * is the reference's.  Children of a removed entity become dirty roots (sc_ecs.cpp:151-160).  The net relocations
 * are returned: entity that was at moved_from[k] is now at moved_to[k] (arrays of `count`; may be NULL).
 * Device cost is O(count) unless a removed or relocated entity has children (then the hierarchy is re-linked). */
int scTickRemoveEntities(ScTickContext* ctx, const uint32_t* dense_indices, uint32_t count,
                         uint32_t* moved_from, uint32_t* moved_to, uint32_t* moved_count);

/* ---- ray queries over the broadphase bins ----
 * A batch of rays answered inside the tick, after the bins are filled and before the pair search consumes them --
 * what the traffic AI asks once per agent and frame (sc_traffic_ai.cpp:319, :644) and the vehicle camera once
 * (sc_vehicle.cpp:600).  Shaped like PhysicsWorld::raycast (src/engine/physics/sc_physics.cpp:740-777): the
 * direction is normalised the same way (no hit when |dir|^2 <= 1e-6), the segment runs from origin to
 * origin + ndir * max_dist, a box takes part when (group & mask) != 0 and its own mask is not empty (Bullet's default
 * filter with the callback's group 0xFFFF), the closest hit wins.  The reference tests Bullet's exact shapes; here the
 * candidates are the world AABBs of the broadphase, tested with the reference's own slab arithmetic
 * (intersectRayAABB, tools/world_editor/editor_core/editor_core.cpp:438-470); equal distances go to the lower id.
 * On a tiled world a context answers for the boxes registered in its own sectors, which includes the neighbours' boxes
 * that reach into them (the queries run after the border merge); the part of a ray beyond the tile is the neighbour's. */
typedef struct ScTickRayHit     /* RaycastHit, sc_physics.h:106-114 */
{
  uint32_t hit;                /* 0 / 1 */
  uint32_t id;                 /* rank << 24 | dense index of the box that was hit (0xFFFFFFFF: none) */
  float    distance;           /* along the normalised direction */
  float    position[3];        /* origin + ndir * distance */
  float    normal[3];          /* axis normal of the face the ray entered through; (0,1,0) when it starts inside the box */
  uint32_t layer;              /* the box's collision group */
  uint32_t pad[2];
} ScTickRayHit;
/* origin3 / dir3: [count][3]; max_dist, mask: [count].  The set stays until it is replaced (count 0 clears it). */
int scTickSetRayQueries(ScTickContext* ctx, uint32_t count, const float* origin3, const float* dir3,
                        const float* max_dist, const uint32_t* mask);
/* results of the last scTickRun(... | SC_TICK_BROADPHASE | SC_TICK_RAYS) (after scTickRunPairs on a tiled world) */
int scTickReadRayHits(ScTickContext* ctx, ScTickRayHit* hits, uint32_t capacity, uint32_t* count);

/* isOccupiedWorld (src/engine/traffic/sc_traffic_spawner.cpp:93-116), for a batch of at most 256 points: blocked[k] = 1
 * when some entity whose collision group meets mask[k] has dx*dx + dz*dz < radius[k]*radius[k] to point k, measured on
 * Transform::localPos in the xz plane like the reference (which walks its TrafficAgent and VehicleComponent pools; the
 * group mask selects the same entities here).  Answers from the positions as they are on the device now; synchronises. */
int scTickQueryOccupied(ScTickContext* ctx, uint32_t count, const float* pos3, const float* radius, const uint32_t* mask,
                        uint8_t* blocked);

/* ---- measurement ---- */
/* record HIP events around the kernel launches (on the context's stream) of every `enable`-th tick
 * from now on (1 = every tick; event records cost host time, so long runs sample); 0 = stop */
int scTickSetProfiling(ScTickContext* ctx, int enable);
/* ... and only around the kernels of `mask` (bit SC_TICK_K_*; 0 = all, the default).  Timing a launch by events costs ~6 us of gap on the
 * queue: a run that is itself being timed samples the dominant kernel alone. */
int scTickSetProfilingKernels(ScTickContext* ctx, uint32_t mask);
/* durations (ms) of the launches of `kernel` recorded since profiling was enabled; synchronises */
int scTickGetKernelTimes(ScTickContext* ctx, uint32_t kernel, float* ms, uint32_t capacity, uint32_t* count);
/* capture the current stage sequence into a hipGraph and replay it on scTickRun (0 = eager launches).  scTickTileStep on
 * an in-order tile replays the whole step (RCCL group included) as one graph; on a pipelined tile each half of the step is
 * a graph of its own, on its own stream.  Not combinable with the frame read-back. */
int scTickSetGraphMode(ScTickContext* ctx, int enable);
/* native stream handle (hipStream_t) for callers that order their own work against the tick */
void* scTickGetStream(ScTickContext* ctx);

#ifdef __cplusplus
}
#endif
#endif /* SC_TICK_H */
