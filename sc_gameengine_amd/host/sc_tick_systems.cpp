// sc_tick_systems.cpp -- the adapter systems ("compat" mode): the engine's World stays the source of
// truth, the device holds a mirror in Transform-pool dense order, and after each system the engine's
// structures are in the state the original system would have left them in.
//
// Built against the engine's real headers with -DSC_TICK_USE_REFERENCE_HEADERS (as a maintainer
// would), or against sc_api_mirror.h where those headers do not exist (tests on the GPU box).
#ifdef SC_TICK_USE_REFERENCE_HEADERS
#include "sc_ecs.h"
#include "sc_world_partition.h"
#else
#include "sc_api_mirror.h"
#endif
#include "sc_tick_systems.h"

#include <algorithm>
#include <cstdio>
#include <cstring>

namespace sc_amd
{
  namespace
  {
    bool note(TickAdapter* a, const char* what)
    {
      std::snprintf(a->lastError, sizeof a->lastError, "%s: %s", what, scTickGetLastError(a->ctx));
      return false;
    }
#define SC_TRY(a, call) do { if (!(call)) { note((a), #call); return false; } } while (0)

    struct Gathered
    {
      std::vector<sc::Entity> ents;
      std::vector<sc::Transform*> xf;
    };

    void gather(sc::World& world, Gathered& g)
    {
      g.ents.clear(); g.xf.clear();
      world.ForEach<sc::Transform>([&](sc::Entity e, sc::Transform& t) { g.ents.push_back(e); g.xf.push_back(&t); });
    }

    // (Re)build the device mirror when the pool's membership or order changed; otherwise push only what
    // the engine marked dirty.  Returns false on an ABI failure.
    bool syncTransforms(TickAdapter* a, sc::World& world, Gathered& g)
    {
      const uint32_t n = (uint32_t)g.ents.size();
      if (n > a->capacity) { std::snprintf(a->lastError, sizeof a->lastError, "more Transforms (%u) than adapter capacity (%u)", n, a->capacity); return false; }

      uint32_t maxIndex = 0;
      for (const sc::Entity e : g.ents) maxIndex = std::max(maxIndex, e.index());
      bool relayout = !a->deviceValid || a->dense.size() != n;
      if (!relayout) relayout = std::memcmp(a->dense.data(), g.ents.data(), (size_t)n * sizeof(sc::Entity)) != 0;
      if (relayout) {
        a->dense = g.ents;
        a->indexToDense.assign((size_t)maxIndex + 1u, 0u);
        for (uint32_t k = 0; k < n; ++k) a->indexToDense[g.ents[k].index()] = k + 1u;
      }

      // parent validation + zero-scale repair, exactly TransformSystem's first pass (sc_ecs.cpp:139-165)
      std::vector<int32_t> parents(n);
      for (uint32_t k = 0; k < n; ++k) {
        sc::Transform& t = *g.xf[k];
        if (t.localScale[0] == 0.0f && t.localScale[1] == 0.0f && t.localScale[2] == 0.0f) {
          t.localScale[0] = t.localScale[1] = t.localScale[2] = 1.0f;
          t.dirty = true;
        }
        const sc::Entity p = t.parent;
        const bool valid = sc::isValidEntity(p) && p != g.ents[k] && world.isAlive(p) && world.has<sc::Transform>(p);
        if (!valid) {
          if (sc::isValidEntity(p)) t.dirty = true;
          t.parent = sc::kInvalidEntity;
          parents[k] = SC_TICK_NO_PARENT;
        } else {
          parents[k] = (int32_t)(a->indexToDense[p.index()] - 1u);
        }
      }

      if (a->localsCache.size() < 9 * (size_t)n) a->localsCache.resize(9 * (size_t)n);
      std::vector<float> pos, rot, scl;
      auto pushRange = [&](uint32_t first, uint32_t count) -> bool {
        pos.resize((size_t)count * 3); rot.resize((size_t)count * 3); scl.resize((size_t)count * 3);
        for (uint32_t i = 0; i < count; ++i) {
          const sc::Transform& t = *g.xf[first + i];
          std::memcpy(&pos[3 * (size_t)i], t.localPos, 12); std::memcpy(&rot[3 * (size_t)i], t.localRot, 12); std::memcpy(&scl[3 * (size_t)i], t.localScale, 12);
          float* c = &a->localsCache[9 * (size_t)(first + i)];
          std::memcpy(c, t.localPos, 12); std::memcpy(c + 3, t.localRot, 12); std::memcpy(c + 6, t.localScale, 12);
        }
        SC_TRY(a, scTickUploadLocals(a->ctx, first, count, pos.data(), rot.data(), scl.data(), nullptr));
        return true;
      };

      if (relayout) {
        // full re-sync: locals, stored matrices and the exact dirty flags of every entity
        SC_TRY(a, scTickSetEntityCount(a->ctx, n));
        if (n) {
          if (!pushRange(0, n)) return false;
          std::vector<float> mats((size_t)n * 16);
          std::vector<uint8_t> flags(n);
          for (uint32_t k = 0; k < n; ++k) { std::memcpy(&mats[16 * (size_t)k], g.xf[k]->worldMatrix.m, 64); flags[k] = g.xf[k]->dirty ? 1 : 0; }
          SC_TRY(a, scTickUploadWorldMatrices(a->ctx, 0, n, mats.data()));
          SC_TRY(a, scTickSetDirtyFlags(a->ctx, 0, n, flags.data()));
        }
        a->parentIndex = parents;
        SC_TRY(a, scTickSetTopology(a->ctx, parents.data(), n));
        a->deviceValid = true;
        return true;
      }

      // steady state: the device must hold the CURRENT locals of every entity the tick may rebuild.
      // That is more than the dirty ones: a clean entity below a dirty ancestor is rebuilt too, from
      // whatever its locals are now (sc_ecs.cpp:184-188).  So every entity whose locals differ from the
      // mirrored copy is pushed, and the range's dirty flags are then set back to exactly the engine's.
      std::vector<uint8_t> flags;
      for (uint32_t k = 0; k < n;) {
        auto changed = [&](uint32_t q) {
          const sc::Transform& t = *g.xf[q];
          const float* c = &a->localsCache[9 * (size_t)q];
          return t.dirty || std::memcmp(c, t.localPos, 12) != 0 || std::memcmp(c + 3, t.localRot, 12) != 0 || std::memcmp(c + 6, t.localScale, 12) != 0;
        };
        if (!changed(k)) { ++k; continue; }
        uint32_t e = k;
        while (e < n && changed(e)) ++e;
        if (!pushRange(k, e - k)) return false;
        flags.resize(e - k);
        for (uint32_t q = k; q < e; ++q) flags[q - k] = g.xf[q]->dirty ? 1 : 0;
        SC_TRY(a, scTickSetDirtyFlags(a->ctx, k, e - k, flags.data()));
        k = e;
      }
      if (parents != a->parentIndex) {
        a->parentIndex = parents;
        SC_TRY(a, scTickSetTopology(a->ctx, parents.data(), n));
      }
      return true;
    }

    bool syncRenderComponents(TickAdapter* a, sc::World& world, const Gathered& g, bool force)
    {
      const uint32_t n = (uint32_t)g.ents.size();
      std::vector<uint8_t> flags(n);
      for (uint32_t k = 0; k < n; ++k)
        flags[k] = (uint8_t)((world.has<sc::RenderMesh>(g.ents[k]) ? 1u : 0u) | (world.has<sc::Bounds>(g.ents[k]) ? 2u : 0u));
      if (!force && flags == a->compFlags) return true;     // component membership unchanged since the last frame
      a->compFlags = flags;
      if (!n) return true;
      std::vector<uint8_t> hasMesh(n), hasBounds(n);
      std::vector<uint32_t> mesh(n, 0u), material(n, 0u);
      std::vector<float> bmin((size_t)n * 3, 0.0f), bmax((size_t)n * 3, 0.0f);
      for (uint32_t k = 0; k < n; ++k) {
        hasMesh[k] = flags[k] & 1u; hasBounds[k] = (flags[k] >> 1) & 1u;
        if (hasMesh[k]) { const sc::RenderMesh* rm = world.get<sc::RenderMesh>(g.ents[k]); mesh[k] = rm->meshId; material[k] = rm->materialId; }
        if (hasBounds[k]) {
          const sc::Bounds* b = world.get<sc::Bounds>(g.ents[k]);
          bmin[3 * (size_t)k] = b->localAabb.min.x; bmin[3 * (size_t)k + 1] = b->localAabb.min.y; bmin[3 * (size_t)k + 2] = b->localAabb.min.z;
          bmax[3 * (size_t)k] = b->localAabb.max.x; bmax[3 * (size_t)k + 1] = b->localAabb.max.y; bmax[3 * (size_t)k + 2] = b->localAabb.max.z;
        }
      }
      SC_TRY(a, scTickUploadRenderMeshes(a->ctx, 0, n, hasMesh.data(), mesh.data(), material.data()));
      SC_TRY(a, scTickUploadBounds(a->ctx, 0, n, bmin.data(), bmax.data(), hasBounds.data()));
      return true;
    }
  } // namespace

  TickAdapter* CreateTickAdapter(int device, uint32_t capacity)
  {
    ScTickContextDesc d{};
    d.device_ordinal = device;
    d.capacity = capacity;
    ScTickContext* ctx = scTickCreateContext(&d);
    if (!ctx) return nullptr;
    TickAdapter* a = new TickAdapter();
    a->ctx = ctx;
    a->capacity = capacity;
    return a;
  }

  void DestroyTickAdapter(TickAdapter* a)
  {
    if (!a) return;
    scTickDestroyContext(a->ctx);
    delete a;
  }

  // ---- TransformSystem ----------------------------------------------------------------------------
  void TransformSystem(sc::World& world, float, void* user)
  {
    TickAdapter* a = static_cast<TickAdapter*>(user);
    if (!a || !a->ctx) return;                                   // null user: silent no-op, as the engine's systems
    Gathered g;
    gather(world, g);
    if (g.ents.empty()) return;
    if (!syncTransforms(a, world, g)) return;
    if (!scTickRun(a->ctx, SC_TICK_XFORM)) { note(a, "scTickRun"); return; }

    // write back: worldMatrix of everything, dirty as the device left it (false, except cycle members)
    const uint32_t n = (uint32_t)g.ents.size();
    std::vector<float> mats((size_t)n * 16);
    std::vector<uint8_t> dirty(n);
    if (!scTickReadWorldMatrices(a->ctx, 0, n, mats.data()) || !scTickReadDirty(a->ctx, 0, n, dirty.data())) { note(a, "read back"); return; }
    for (uint32_t k = 0; k < n; ++k) {
      sc::Transform& t = *g.xf[k];
      // an entity nobody recomputed keeps its matrix bit for bit (the device copy came from it)
      if (t.dirty || std::memcmp(t.worldMatrix.m, &mats[16 * (size_t)k], 64) != 0) std::memcpy(t.worldMatrix.m, &mats[16 * (size_t)k], 64);
      t.dirty = dirty[k] != 0;
    }
    a->frames++;
  }

  // ---- CullingSystem ------------------------------------------------------------------------------
  void CullingSystem(sc::World& world, float, void* user)
  {
    TickAdapter* a = static_cast<TickAdapter*>(user);
    if (!a || !a->ctx || !a->culling || !a->culling->frame) return;
    sc::CullingState& st = *a->culling;

    Gathered g;
    gather(world, g);
    bool fresh = false;
    const bool poolMoved = a->dense.size() != g.ents.size() ||
                           std::memcmp(a->dense.data(), g.ents.data(), g.ents.size() * sizeof(sc::Entity)) != 0;
    if (!a->transformsOnDevice || !a->deviceValid || poolMoved) {
      // culling only (or the pool changed since TransformSystem ran): mirror the matrices as they stand;
      // the next sc_amd::TransformSystem then re-syncs everything
      const uint32_t n = (uint32_t)g.ents.size();
      if (n > a->capacity) return;
      a->dense = g.ents;
      a->deviceValid = false;
      if (!scTickSetEntityCount(a->ctx, n)) return;
      if (n) {
        std::vector<float> mats((size_t)n * 16);
        for (uint32_t k = 0; k < n; ++k) std::memcpy(&mats[16 * (size_t)k], g.xf[k]->worldMatrix.m, 64);
        std::vector<int32_t> roots(n, SC_TICK_NO_PARENT);
        if (!scTickSetTopology(a->ctx, roots.data(), n) || !scTickUploadWorldMatrices(a->ctx, 0, n, mats.data())) { note(a, "matrix upload"); return; }
      }
      fresh = true;
    }
    if (!syncRenderComponents(a, world, g, fresh)) return;

    st.candidates.clear();
    for (uint32_t k = 0; k < g.ents.size(); ++k) if (a->compFlags[k] & 1u) st.candidates.push_back(g.ents[k]);
    const uint32_t total = (uint32_t)st.candidates.size();
    st.stats.renderablesTotal = total;
    st.visible.clear(); st.culled.clear();
    st.visible.reserve(total); st.culled.reserve(total);
    if (total == 0) { st.stats.visible = 0; st.stats.culled = 0; return; }

    if (st.freezeCulling) {                                      // sc_world_partition.cpp:1227-1233: frustum and mask untouched
      st.visible.insert(st.visible.end(), st.candidates.begin(), st.candidates.end());
      st.stats.visible = total; st.stats.culled = 0;
      scTickSetFreezeCulling(a->ctx, 1);
      scTickSetViewProj(a->ctx, st.frame->viewProj.m);
      scTickRun(a->ctx, SC_TICK_CULL);                           // keeps the device's visible list in step for RenderPrep
      return;
    }
    scTickSetFreezeCulling(a->ctx, 0);
    if (!scTickSetViewProj(a->ctx, st.frame->viewProj.m) || !scTickRun(a->ctx, SC_TICK_CULL | SC_TICK_CULLED_LIST)) { note(a, "cull"); return; }

    float planes[24]; int valid = 0;
    scTickGetFrustumPlanes(a->ctx, planes, &valid);
    for (int p = 0; p < 6; ++p) { std::memcpy(st.frustum.planes[p].n, planes + 4 * p, 12); st.frustum.planes[p].d = planes[4 * p + 3]; }
    st.frustum.valid = valid != 0;

    const uint32_t n = (uint32_t)g.ents.size();
    std::vector<uint32_t> vis(n), cul(n);
    uint32_t nv = 0, nc = 0;
    std::vector<uint64_t> bits((n + 63u) / 64u + 1u);
    if (!scTickReadVisible(a->ctx, vis.data(), n, &nv) || !scTickReadCulled(a->ctx, cul.data(), n, &nc) ||
        !scTickReadVisibilityBits(a->ctx, bits.data(), (uint32_t)bits.size())) { note(a, "cull read back"); return; }
    for (uint32_t k = 0; k < nv; ++k) st.visible.push_back(g.ents[vis[k]]);
    for (uint32_t k = 0; k < nc; ++k) st.culled.push_back(g.ents[cul[k]]);
    if (st.visibilityMask.size() < total) st.visibilityMask.resize(total);      // only ever grows (:1236-1237)
    uint32_t ordinal = 0;
    for (uint32_t k = 0; k < n; ++k) if (a->compFlags[k] & 1u) st.visibilityMask[ordinal++] = (uint8_t)((bits[k >> 6] >> (k & 63u)) & 1ull);
    st.stats.visible = (uint32_t)st.visible.size();
    st.stats.culled = (uint32_t)st.culled.size();
  }

  // ---- RenderPrepStreamingSystem (draw emission) -----------------------------------------------------
  void RenderPrepStreamingSystem(sc::World& world, float, void* user)
  {
    (void)world;
    TickAdapter* a = static_cast<TickAdapter*>(user);
    if (!a || !a->ctx || !a->renderPrep || !a->renderPrep->frame) return;
    sc::RenderPrepStreamingState& st = *a->renderPrep;
    sc::RenderFrameData& frame = *st.frame;
    frame.clear();
    const uint32_t maxDraws = st.streaming ? st.streaming->budgets.maxDrawsBudget : 0u;
    scTickSetDrawBudget(a->ctx, maxDraws);
    // the visible list of this frame's CullingSystem is still on the device: emit from it
    if (!scTickRun(a->ctx, SC_TICK_DRAWS)) { note(a, "draws"); return; }
    uint32_t count = 0;
    scTickReadDraws(a->ctx, nullptr, 0, &count);
    std::vector<ScTickDrawItem> items(count ? count : 1u);
    if (count && !scTickReadDraws(a->ctx, items.data(), count, &count)) { note(a, "draw read back"); return; }
    frame.draws.reserve(count);
    for (uint32_t k = 0; k < count; ++k) {
      sc::DrawItem d{};
      d.entity = a->dense[items[k].dense_index];
      d.meshId = items[k].mesh_id;
      d.materialId = items[k].material_id;
      std::memcpy(d.model.m, items[k].model, 64);
      frame.draws.push_back(d);
      if (a->onDraw) a->onDraw(a->onDrawUser, frame.draws.back());
    }
    ScTickCounts c{};
    scTickGetCounts(a->ctx, &c);
    st.stats.drawsEmitted = c.draws_emitted;
    st.stats.drawsDroppedByBudget = c.draws_dropped;
  }
}
