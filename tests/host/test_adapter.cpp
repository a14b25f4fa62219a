// test_adapter.cpp -- drives the C++ adapter systems (sc_amd::*, engine signature) on a World built
// with the API mirror, frame by frame, next to the oracle's restatement of the engine's own systems,
// and compares every piece of engine state the originals would have produced.
// TEST INFRASTRUCTURE (links oracle/liboracle.so as the checker).  Needs a GPU.  Exit code 0 = pass.
#include "sc_api_mirror.h"
#include "sc_tick_systems.h"
#include "sc_oracle.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

static int gFail = 0;
#define CHECK(cond, ...) do { if (!(cond)) { if (gFail < 20) { std::printf("FAIL %s:%d: ", __FILE__, __LINE__); std::printf(__VA_ARGS__); std::printf("\n"); } ++gFail; } } while (0)

static bool sameValues(const float* a, const float* b, int n)     // IEEE equality: +0 == -0
{
  for (int i = 0; i < n; ++i) if (!(a[i] == b[i])) return false;
  return true;
}

struct Scene
{
  sc::World world;
  OrcWorld* ow = orc_world_new();
  std::vector<sc::Entity> live;
  std::mt19937 rng{ 12345 };
  float uni(float a, float b) { return std::uniform_real_distribution<float>(a, b)(rng); }

  sc::Entity spawn(sc::Entity parent, bool mesh, bool bounds)
  {
    const sc::Entity e = world.create();
    const uint32_t oe = orc_entity_create(ow);
    CHECK(oe == e.value, "entity handles diverged %u vs %u", oe, e.value);
    sc::Transform& t = world.add<sc::Transform>(e);
    OrcTransform* ot = orc_add_transform(ow, oe);
    const float pos[3] = { uni(-300, 300), uni(0, 20), uni(-300, 300) };
    const float rot[3] = { uni(-3, 3), uni(-3, 3), uni(-3, 3) };
    const float scl[3] = { uni(0.3f, 3), uni(0.3f, 3), uni(0.3f, 3) };
    sc::setLocal(t, pos, rot, scl);
    std::memcpy(ot->localPos, pos, 12); std::memcpy(ot->localRot, rot, 12); std::memcpy(ot->localScale, scl, 12); ot->dirty = 1;
    if (sc::isValidEntity(parent)) { sc::setParent(t, parent); ot->parent = parent.value; }
    if (mesh) {
      sc::RenderMesh& rm = world.add<sc::RenderMesh>(e);
      rm.meshId = rng() % 7; rm.materialId = rng() % 5;
      OrcRenderMesh* om = orc_add_render_mesh(ow, oe); om->meshId = rm.meshId; om->materialId = rm.materialId;
    }
    if (bounds) {
      sc::Bounds& b = world.add<sc::Bounds>(e);
      b.localAabb.min = { -uni(0.2f, 2), -uni(0.2f, 2), -uni(0.2f, 2) };
      b.localAabb.max = { uni(0.2f, 2), uni(0.2f, 2), uni(0.2f, 2) };
      OrcBounds* ob = orc_add_bounds(ow, oe);
      ob->min[0] = b.localAabb.min.x; ob->min[1] = b.localAabb.min.y; ob->min[2] = b.localAabb.min.z;
      ob->max[0] = b.localAabb.max.x; ob->max[1] = b.localAabb.max.y; ob->max[2] = b.localAabb.max.z;
    }
    live.push_back(e);
    return e;
  }
  void kill(size_t k)
  {
    const sc::Entity e = live[k];
    CHECK(world.destroy(e), "destroy failed");
    CHECK(orc_entity_destroy(ow, e.value), "oracle destroy failed");
    live[k] = live.back(); live.pop_back();
  }
};

int main(int argc, char** argv)
{
  const uint32_t N = argc > 1 ? (uint32_t)std::atoi(argv[1]) : 20000u;
  Scene sc_;
  // camera entity first, as the sandbox's spawner does (sc_ecs.cpp:320-331)
  const sc::Entity cam = sc_.world.create();
  const uint32_t ocam = orc_entity_create(sc_.ow);
  {
    sc::Transform& t = sc_.world.add<sc::Transform>(cam);
    const float p[3] = { 0, 40, 0 }, r[3] = { -0.5f, 0.4f, 0.1f }, s[3] = { 1, 1, 1 };
    sc::setLocal(t, p, r, s);
    sc::Camera& c = sc_.world.add<sc::Camera>(cam); c.active = true;
    OrcTransform* ot = orc_add_transform(sc_.ow, ocam);
    std::memcpy(ot->localPos, p, 12); std::memcpy(ot->localRot, r, 12); ot->dirty = 1;
    OrcCamera* oc = orc_add_camera(sc_.ow, ocam); oc->active = 1;
  }
  for (uint32_t i = 0; i < N; ++i) {
    sc::Entity parent = sc::kInvalidEntity;
    if (!sc_.live.empty() && (sc_.rng() % 100) < 45) parent = sc_.live[sc_.rng() % sc_.live.size()];
    sc_.spawn(parent, (sc_.rng() % 100) < 85, (sc_.rng() % 100) < 85);
  }

  sc::CullingState culling; culling.frame = &sc_.world.renderFrame();
  sc::WorldStreamingState streaming; streaming.budgets.maxDrawsBudget = 700;
  sc::RenderPrepStreamingState renderPrep; renderPrep.frame = &sc_.world.renderFrame(); renderPrep.culling = &culling; renderPrep.streaming = &streaming;
  sc::CameraSystemState cameraState; cameraState.frame = &sc_.world.renderFrame();

  sc_amd::TickAdapter* adapter = sc_amd::CreateTickAdapter(0, N * 2 + 16);
  if (!adapter) { std::printf("CreateTickAdapter failed: %s\n", scTickGetLastError(nullptr)); return 2; }
  adapter->culling = &culling;
  adapter->renderPrep = &renderPrep;
  uint32_t drawsSeen = 0;
  adapter->onDraw = [](void* u, const sc::DrawItem&) { ++*static_cast<uint32_t*>(u); };
  adapter->onDrawUser = &drawsSeen;

  OrcCameraState ocs{}; ocs.aspect = cameraState.aspect;
  OrcCullingState* ocull = orc_culling_state_new();

  for (int frame = 0; frame < 6; ++frame) {
    // ---- gameplay between frames: moves, silent edits, reparenting, zero scales, churn ----
    if (frame > 0) {
      for (int k = 0; k < 400; ++k) {
        const sc::Entity e = sc_.live[sc_.rng() % sc_.live.size()];
        const float x = sc_.uni(-300, 300), y = sc_.uni(0, 20), z = sc_.uni(-300, 300);
        sc::setLocalPosition(*sc_.world.get<sc::Transform>(e), x, y, z);
        OrcTransform* ot = orc_get_transform(sc_.ow, e.value); ot->localPos[0] = x; ot->localPos[1] = y; ot->localPos[2] = z; ot->dirty = 1;
      }
      for (int k = 0; k < 50; ++k) {          // silent edits: locals change, dirty stays false -> must NOT move
        const sc::Entity e = sc_.live[sc_.rng() % sc_.live.size()];
        const float x = sc_.uni(-300, 300);
        sc_.world.get<sc::Transform>(e)->localPos[0] = x;
        orc_get_transform(sc_.ow, e.value)->localPos[0] = x;
      }
      for (int k = 0; k < 30; ++k) {          // reparent (may create cycles: those freeze, as in the engine)
        const sc::Entity e = sc_.live[sc_.rng() % sc_.live.size()], p = sc_.live[sc_.rng() % sc_.live.size()];
        sc::setParent(*sc_.world.get<sc::Transform>(e), p);
        OrcTransform* ot = orc_get_transform(sc_.ow, e.value); ot->parent = p.value; ot->dirty = 1;
      }
      for (int k = 0; k < 5; ++k) {
        const sc::Entity e = sc_.live[sc_.rng() % sc_.live.size()];
        sc::Transform& t = *sc_.world.get<sc::Transform>(e); t.localScale[0] = t.localScale[1] = t.localScale[2] = 0.0f;
        OrcTransform* ot = orc_get_transform(sc_.ow, e.value); ot->localScale[0] = ot->localScale[1] = ot->localScale[2] = 0.0f;
      }
      if (frame == 2 || frame == 4) {         // despawn / spawn: the pool's dense order changes (swap-remove)
        for (int k = 0; k < 200; ++k) sc_.kill(sc_.rng() % sc_.live.size());
        for (int k = 0; k < 150; ++k) sc_.spawn(sc_.live[sc_.rng() % sc_.live.size()], true, (k % 3) != 0);
      }
      if (frame == 3) { culling.freezeCulling = true; ocull->freezeCulling = 1; }
      if (frame == 4) { culling.freezeCulling = false; ocull->freezeCulling = 0; streaming.budgets.maxDrawsBudget = 0; }
      if (frame == 5) adapter->transformsOnDevice = false;       // culling-only deployment: CPU transforms, GPU cull
    }

    // ---- the frame: adapter systems in the sandbox's order (main.cpp:256-259) ----
    if (adapter->transformsOnDevice) sc_amd::TransformSystem(sc_.world, 0.016f, adapter);
    orc_transform_system(sc_.ow);
    if (!adapter->transformsOnDevice) {       // stand-in for the engine's CPU TransformSystem: take the oracle's result
      const uint32_t n = orc_transform_count(sc_.ow);
      const uint32_t* de = orc_transform_dense_entities(sc_.ow);
      OrcTransform* dd = orc_transform_dense_data(sc_.ow);
      for (uint32_t k = 0; k < n; ++k) {
        sc::Transform* t = sc_.world.get<sc::Transform>(sc::Entity{ de[k] });
        std::memcpy(t->worldMatrix.m, dd[k].worldMatrix, 64); t->dirty = dd[k].dirty != 0; t->parent.value = dd[k].parent;
        std::memcpy(t->localScale, dd[k].localScale, 12);
      }
    }
    sc::CameraSystem(sc_.world, 0.016f, &cameraState);
    orc_camera_system(sc_.ow, &ocs);
    sc_amd::CullingSystem(sc_.world, 0.016f, adapter);
    orc_culling_system(sc_.ow, ocull, ocs.viewProj);
    drawsSeen = 0;
    sc_amd::RenderPrepStreamingSystem(sc_.world, 0.016f, adapter);
    CHECK(adapter->lastError[0] == 0, "adapter error: %s", adapter->lastError);

    // ---- compare everything ----
    const uint32_t n = orc_transform_count(sc_.ow);
    CHECK(n == sc_.world.componentCount<sc::Transform>(), "transform count %u vs %u", n, sc_.world.componentCount<sc::Transform>());
    const uint32_t* de = orc_transform_dense_entities(sc_.ow);
    const OrcTransform* dd = orc_transform_dense_data(sc_.ow);
    uint32_t k = 0;
    sc_.world.ForEach<sc::Transform>([&](sc::Entity e, sc::Transform& t) {
      CHECK(e.value == de[k], "dense order differs at %u", k);
      CHECK(sameValues(t.worldMatrix.m, dd[k].worldMatrix, 16), "frame %d worldMatrix of dense %u differs", frame, k);
      CHECK((t.dirty ? 1 : 0) == dd[k].dirty, "frame %d dirty of dense %u: %d vs %d", frame, k, (int)t.dirty, (int)dd[k].dirty);
      CHECK(t.parent.value == dd[k].parent, "frame %d parent of dense %u", frame, k);
      CHECK(sameValues(t.localScale, dd[k].localScale, 3), "frame %d localScale of dense %u", frame, k);
      ++k;
    });
    CHECK(std::memcmp(sc_.world.renderFrame().viewProj.m, ocs.viewProj, 64) == 0, "frame %d viewProj differs", frame);
    CHECK(culling.candidates.size() == ocull->candidatesLen && std::memcmp(culling.candidates.data(), ocull->candidates, 4 * ocull->candidatesLen) == 0, "frame %d candidates differ", frame);
    CHECK(culling.visible.size() == ocull->visibleLen && std::memcmp(culling.visible.data(), ocull->visible, 4 * ocull->visibleLen) == 0,
          "frame %d visible differs (%zu vs %u)", frame, culling.visible.size(), ocull->visibleLen);
    CHECK(culling.culled.size() == ocull->culledLen && std::memcmp(culling.culled.data(), ocull->culled, 4 * ocull->culledLen) == 0, "frame %d culled differs", frame);
    CHECK(culling.stats.renderablesTotal == ocull->renderablesTotal && culling.stats.visible == ocull->visibleCount && culling.stats.culled == ocull->culledCount, "frame %d stats", frame);
    if (!culling.freezeCulling) {
      CHECK(std::memcmp(culling.visibilityMask.data(), ocull->visibilityMask, ocull->candidatesLen) == 0, "frame %d visibilityMask differs", frame);
      for (int p = 0; p < 6; ++p)
        CHECK(std::memcmp(culling.frustum.planes[p].n, ocull->frustum.planes[p].n, 12) == 0 && culling.frustum.planes[p].d == ocull->frustum.planes[p].d, "frame %d plane %d", frame, p);
      CHECK(culling.frustum.valid, "frustum valid");
    }
    std::vector<OrcDrawItem> want(ocull->visibleLen + 1);
    uint32_t dropped = 0;
    const uint32_t emitted = orc_render_prep_streaming(sc_.ow, ocull, streaming.budgets.maxDrawsBudget, want.data(), (uint32_t)want.size(), &dropped);
    const auto& draws = sc_.world.renderFrame().draws;
    CHECK(draws.size() == emitted && renderPrep.stats.drawsEmitted == emitted && renderPrep.stats.drawsDroppedByBudget == dropped && drawsSeen == emitted,
          "frame %d draws %zu vs %u (dropped %u vs %u)", frame, draws.size(), emitted, renderPrep.stats.drawsDroppedByBudget, dropped);
    for (uint32_t q = 0; q < emitted && q < draws.size(); ++q)
      CHECK(draws[q].entity.value == want[q].entity && draws[q].meshId == want[q].meshId && draws[q].materialId == want[q].materialId &&
            sameValues(draws[q].model.m, want[q].model, 16), "frame %d draw %u differs", frame, q);
    std::printf("frame %d: %u transforms, %zu candidates, %zu visible, %zu draws%s\n", frame, n, culling.candidates.size(), culling.visible.size(), draws.size(), gFail ? "  (FAILURES)" : "");
  }

  sc_amd::DestroyTickAdapter(adapter);
  orc_culling_state_free(ocull);
  orc_world_free(sc_.ow);
  std::printf(gFail ? "test_adapter: %d check(s) FAILED\n" : "test_adapter: all checks passed\n", gFail);
  return gFail ? 1 : 0;
}
