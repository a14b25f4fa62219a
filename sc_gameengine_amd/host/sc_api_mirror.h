// sc_api_mirror.h -- stand-alone mirror of the slice of the engine's public API that the tick
// adapter touches, for building and testing the adapter where the engine's headers are absent
// (the GPU box).  Same names, fields and observable behaviour as the engine's own headers --
//   src/core/include/sc_math.h, src/core/include/sc_ecs.h (Entity, Transform + mutators, Camera,
//   RenderMesh, DrawItem, RenderFrameData, World, CameraSystemState, CameraSystem) and
//   src/engine/world/sc_world_partition.h (Vec3, AABB, Plane, Frustum, Bounds, CullingStats,
//   CullingState, RenderPrepStats, WorldStreamingBudgets/State, RenderPrepStreamingState)
// -- but an independent implementation (type-erased paged sparse sets, math through the C ABI's
// host helpers).  In the engine itself this file is not used: sc_tick_systems.cpp is compiled with
// -DSC_TICK_USE_REFERENCE_HEADERS against the real headers (oracle/Makefile: adapter-check).
#pragma once
#include <cstdint>
#include <cstring>
#include <memory>
#include <typeindex>
#include <unordered_map>
#include <utility>
#include <vector>

#include "sc_tick.h"

namespace sc
{
  // ---------------- math (sc_math.h) ----------------
  struct alignas(16) Mat4
  {
    float m[16]{};
    static Mat4 identity() noexcept { Mat4 r; r.m[0] = r.m[5] = r.m[10] = r.m[15] = 1.0f; return r; }
  };
  inline Mat4 mat4_identity() noexcept { return Mat4::identity(); }
  inline Mat4 mat4_mul(const Mat4& a, const Mat4& b) noexcept { Mat4 r; scTickHostMat4Mul(a.m, b.m, r.m); return r; }
  inline Mat4 mat4_trs(const float pos[3], const float rot[3], const float scale[3]) noexcept { Mat4 r; scTickHostMat4Trs(pos, rot, scale, r.m); return r; }
  inline Mat4 mat4_inverse(const Mat4& a) noexcept { Mat4 r; scTickHostMat4Inverse(a.m, r.m); return r; }
  inline Mat4 mat4_perspective_rh_zo(float fovY, float aspect, float zn, float zf, bool flipY) noexcept
  { Mat4 r; scTickHostMat4PerspectiveRhZo(fovY, aspect, zn, zf, flipY ? 1 : 0, r.m); return r; }

  // ---------------- entity handle (sc_ecs.h:14-37) ----------------
  struct Entity
  {
    uint32_t value = 0;
    static constexpr uint32_t INDEX_BITS = 24, GENERATION_BITS = 8, INDEX_MASK = (1u << INDEX_BITS) - 1u;
    static Entity fromParts(uint32_t index, uint32_t generation) { Entity e; e.value = (generation << INDEX_BITS) | (index & INDEX_MASK); return e; }
    uint32_t index() const { return value & INDEX_MASK; }
    uint32_t generation() const { return value >> INDEX_BITS; }
    bool operator==(const Entity& o) const { return value == o.value; }
    bool operator!=(const Entity& o) const { return value != o.value; }
  };
  static constexpr Entity kInvalidEntity{ 0xFFFFFFFFu };
  inline bool isValidEntity(Entity e) { return e.value != kInvalidEntity.value; }

  // ---------------- components (sc_ecs.h:63-111, 159-173) ----------------
  struct Transform
  {
    Entity parent = kInvalidEntity;
    float localPos[3] = { 0, 0, 0 };
    float localRot[3] = { 0, 0, 0 };
    float localScale[3] = { 1, 1, 1 };
    Mat4 worldMatrix = Mat4::identity();
    bool dirty = true;
  };
  inline void markDirty(Transform& t) { t.dirty = true; }
  inline void setLocal(Transform& t, const float pos[3], const float rot[3], const float scale[3])
  {
    std::memcpy(t.localPos, pos, 12); std::memcpy(t.localRot, rot, 12); std::memcpy(t.localScale, scale, 12);
    t.dirty = true;
  }
  inline void setParent(Transform& t, Entity parent) { t.parent = parent; t.dirty = true; }
  inline void setLocalPosition(Transform& t, float x, float y, float z) { t.localPos[0] = x; t.localPos[1] = y; t.localPos[2] = z; t.dirty = true; }

  struct Camera { float fovY = 60.0f, nearZ = 0.1f, farZ = 1000.0f, aspect = 16.0f / 9.0f; bool active = false; };
  struct RenderMesh { uint32_t meshId = 0, materialId = 0; };
  struct DrawItem { Entity entity{}; uint32_t meshId = 0, materialId = 0; Mat4 model = Mat4::identity(); };
  struct RenderFrameData
  {
    Mat4 viewProj = Mat4::identity();
    std::vector<DrawItem> draws;
    void clear() { draws.clear(); }
    void reserve(uint32_t n) { draws.reserve(n); }
  };

  // ---------------- world (sc_ecs.h:282-418 surface) ----------------
  // Storage: one sparse set per component type; removal moves the last dense element into the hole,
  // which is the engine's observable dense-order rule (sc_ecs.h:240-262).
  class World
  {
    struct SetBase
    {
      std::vector<Entity> owners;          // dense
      std::vector<uint32_t> where;         // entity index -> dense slot + 1
      virtual ~SetBase() = default;
      virtual void dropSlot(uint32_t slot) = 0;
      uint32_t slotOf(Entity e) const { const uint32_t i = e.index(); return i < where.size() ? where[i] : 0u; }
      void erase(Entity e)
      {
        const uint32_t s = slotOf(e);
        if (!s) return;
        const uint32_t hole = s - 1u, last = (uint32_t)owners.size() - 1u;
        if (hole != last) { owners[hole] = owners[last]; where[owners[hole].index()] = hole + 1u; }
        dropSlot(hole);
        owners.pop_back();
        where[e.index()] = 0u;
      }
    };
    template <typename T> struct Set final : SetBase
    {
      std::vector<T> items;
      void dropSlot(uint32_t slot) override { if (slot + 1u != items.size()) items[slot] = items.back(); items.pop_back(); }
      T& place(Entity e)
      {
        const uint32_t i = e.index();
        if (i >= where.size()) where.resize(i + 1u, 0u);
        if (where[i]) return items[where[i] - 1u];
        owners.push_back(e); items.emplace_back(); where[i] = (uint32_t)owners.size();
        return items.back();
      }
    };

  public:
    Entity create()
    {
      if (!m_recycled.empty()) { const uint32_t i = m_recycled.back(); m_recycled.pop_back(); ++m_alive; return Entity::fromParts(i, m_gen[i]); }
      m_gen.push_back(0u); ++m_alive;
      return Entity::fromParts((uint32_t)m_gen.size() - 1u, 0u);
    }
    bool destroy(Entity e)
    {
      const uint32_t i = e.index();
      if (i >= m_gen.size() || m_gen[i] != e.generation()) return false;
      ++m_gen[i]; m_recycled.push_back(i); if (m_alive) --m_alive;
      for (auto& kv : m_sets) kv.second->erase(e);
      return true;
    }
    bool isAlive(Entity e) const { const uint32_t i = e.index(); return i < m_gen.size() && m_gen[i] == e.generation(); }
    void reserveEntities(uint32_t n) { m_gen.reserve(n); }

    template <typename T, typename... A> T& add(Entity e, A&&... a) { T& c = set<T>().place(e); c = T{ std::forward<A>(a)... }; return c; }
    template <typename T> bool has(Entity e) const { const Set<T>* s = find<T>(); return s && s->slotOf(e) != 0u; }
    template <typename T> T* get(Entity e) { Set<T>* s = findMut<T>(); const uint32_t k = s ? s->slotOf(e) : 0u; return k ? &s->items[k - 1u] : nullptr; }
    template <typename T> void remove(Entity e) { if (Set<T>* s = findMut<T>()) s->erase(e); }
    template <typename T> uint32_t componentCount() const { const Set<T>* s = find<T>(); return s ? (uint32_t)s->owners.size() : 0u; }
    uint32_t entityAliveCount() const { return m_alive; }
    uint32_t entityCapacity() const { return (uint32_t)m_gen.size(); }
    RenderFrameData& renderFrame() { return m_frame; }
    const RenderFrameData& renderFrame() const { return m_frame; }

    // iteration in the dense order of the FIRST type's pool (sc_ecs.h:393-408)
    template <typename T0, typename... Ts, typename F> void ForEach(F&& f)
    {
      Set<T0>* lead = findMut<T0>();
      if (!lead) return;
      for (uint32_t k = 0; k < lead->owners.size(); ++k) {
        const Entity e = lead->owners[k];
        if ((has<Ts>(e) && ... && true)) f(e, lead->items[k], *get<Ts>(e)...);
      }
    }

  private:
    template <typename T> Set<T>& set()
    {
      auto& p = m_sets[std::type_index(typeid(T))];
      if (!p) p = std::make_unique<Set<T>>();
      return static_cast<Set<T>&>(*p);
    }
    template <typename T> const Set<T>* find() const { auto it = m_sets.find(std::type_index(typeid(T))); return it == m_sets.end() ? nullptr : static_cast<const Set<T>*>(it->second.get()); }
    template <typename T> Set<T>* findMut() { auto it = m_sets.find(std::type_index(typeid(T))); return it == m_sets.end() ? nullptr : static_cast<Set<T>*>(it->second.get()); }

    std::vector<uint32_t> m_gen, m_recycled;
    uint32_t m_alive = 0;
    std::unordered_map<std::type_index, std::unique_ptr<SetBase>> m_sets;
    RenderFrameData m_frame;
  };

  // ---------------- camera (sc_ecs.h:445-450, sc_ecs.cpp:213-272) ----------------
  struct CameraSystemState { RenderFrameData* frame = nullptr; Entity activeCamera = kInvalidEntity; float aspect = 16.0f / 9.0f; };
  inline void CameraSystem(World& world, float, void* user)
  {
    auto* st = static_cast<CameraSystemState*>(user);
    if (!st || !st->frame) return;
    Camera* pick = nullptr; Transform* pickT = nullptr; Entity pickE = kInvalidEntity;
    bool haveActive = false;
    world.ForEach<Camera, Transform>([&](Entity e, Camera& c, Transform& t) {
      if (!pick) { pick = &c; pickT = &t; pickE = e; }                    // fallback: first camera
      if (!haveActive && c.active) { pick = &c; pickT = &t; pickE = e; haveActive = true; }
    });
    if (!pick) { st->frame->viewProj = Mat4::identity(); st->activeCamera = kInvalidEntity; return; }
    pick->aspect = (st->aspect > 0.0f) ? st->aspect : pick->aspect;
    scTickHostCameraViewProj(pickT->worldMatrix.m, pick->fovY, pick->aspect, pick->nearZ, pick->farZ, st->frame->viewProj.m);
    st->activeCamera = pickE;
  }

  // ---------------- culling / render-prep state (sc_world_partition.h) ----------------
  class AssetManager;                          // renderer-side, never dereferenced here
  struct Vec3 { float x = 0, y = 0, z = 0; };
  struct AABB { Vec3 min{}; Vec3 max{}; };
  struct Plane { float n[3] = { 0, 0, 0 }; float d = 0; };
  struct Frustum { Plane planes[6]{}; bool valid = false; };
  struct Bounds { AABB localAabb{}; };
  struct CullingStats { uint32_t renderablesTotal = 0, visible = 0, culled = 0; };
  struct CullingState
  {
    RenderFrameData* frame = nullptr;
    bool freezeCulling = false;
    Frustum frustum{};
    CullingStats stats{};
    std::vector<Entity> candidates, visible, culled;
    std::vector<uint8_t> visibilityMask;
  };
  struct RenderPrepStats { uint32_t drawsEmitted = 0, drawsDroppedByBudget = 0; };
  struct WorldStreamingBudgets { uint32_t maxDrawsBudget = 4096u; };
  struct WorldStreamingState { WorldStreamingBudgets budgets{}; uint64_t frameIndex = 0; bool freezeEviction = false; };
  struct RenderPrepStreamingState
  {
    RenderFrameData* frame = nullptr;
    CullingState* culling = nullptr;
    WorldStreamingState* streaming = nullptr;
    AssetManager* assets = nullptr;
    RenderPrepStats stats{};
  };
}
