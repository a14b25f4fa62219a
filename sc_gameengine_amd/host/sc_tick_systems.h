// sc_tick_systems.h -- drop-in replacements for the engine's RenderPrep systems, running on MI355X.
//
// Each function has the engine's system signature  void(World&, float dt, void* user)
// (src/core/include/sc_scheduler.h:38) and leaves the engine's own state exactly as the system it
// replaces does:
//   sc_amd::TransformSystem            <-> sc::TransformSystem            src/core/src/sc_ecs.cpp:118-211
//   sc_amd::CullingSystem              <-> sc::CullingSystem              src/engine/world/sc_world_partition.cpp:1199-1284
//   sc_amd::RenderPrepStreamingSystem  <-> sc::RenderPrepStreamingSystem  src/engine/world/sc_world_partition.cpp:1286-1359
// `user` is a sc_amd::TickAdapter* (it carries the engine's original state pointers), see INTEGRATION.md.
//
// Include the engine's sc_ecs.h + sc_world_partition.h (or sc_api_mirror.h) before this header.
#pragma once
#include <cstdint>
#include <vector>

#include "sc_tick.h"

namespace sc_amd
{
  struct TickAdapter
  {
    ScTickContext* ctx = nullptr;
    uint32_t capacity = 0;

    // the engine's own state, untouched in type and ownership (src/sandbox/src/main.cpp:101-115)
    sc::CullingState* culling = nullptr;
    sc::RenderPrepStreamingState* renderPrep = nullptr;

    // when false the engine keeps its CPU TransformSystem and only culling runs on the GPU:
    // world matrices are then uploaded every frame
    bool transformsOnDevice = true;
    // called per emitted draw, in order (the engine's asset touches, sc_world_partition.cpp:1322-1326)
    void (*onDraw)(void* user, const sc::DrawItem& item) = nullptr;
    void* onDrawUser = nullptr;

    // ---- adapter-private mirror of what the device holds ----
    std::vector<sc::Entity> dense;             // Transform pool dense order at the last sync
    std::vector<int32_t> parentIndex;
    std::vector<uint8_t> compFlags;            // bit0 RenderMesh, bit1 Bounds
    std::vector<uint32_t> indexToDense;        // entity index -> dense slot + 1
    std::vector<float> localsCache;            // pos, rot, scale per dense slot as last pushed
    bool deviceValid = false;
    uint64_t frames = 0;
    char lastError[256] = {};
  };

  // device_ordinal: HIP device; capacity: max entities with a Transform.  nullptr on failure
  // (scTickGetLastError(nullptr) has the reason).
  TickAdapter* CreateTickAdapter(int device_ordinal, uint32_t capacity);
  void DestroyTickAdapter(TickAdapter* a);

  void TransformSystem(sc::World& world, float dt, void* user);
  void CullingSystem(sc::World& world, float dt, void* user);
  void RenderPrepStreamingSystem(sc::World& world, float dt, void* user);
}
