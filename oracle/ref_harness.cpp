// ref_harness.cpp -- C entry points over the REAL reference code, for pinning the oracle.
// TEST INFRASTRUCTURE ONLY.  Built by oracle/Makefile into oracle/_ref/libsc_ref.so, and only
// in a container where /root/reference exists.  It is linked with the reference's own
// src/core/src/sc_math.cpp, compiled unmodified from where it lies, and uses the header-only
// parts of src/core/include/sc_ecs.h (Entity, Transform, ComponentPool<T>).  Nothing from the
// reference is copied into this repository; no stand-in headers or libraries are involved.
// (sc_ecs.cpp and sc_world_partition.cpp do NOT build here without stand-ins -- strncpy_s,
// <windows.h>, <vulkan/vulkan.h> -- so TransformSystem/CullingSystem themselves are not
// available as compiled reference: see DESIGN.md "Oracle".)
#include "sc_ecs.h"
#include "sc_math.h"
#include "sc_world_partition.h"

#include <cstddef>
#include <cstdint>
#include <cstring>

extern "C" {

void ref_mat4_mul(const float* a, const float* b, float* out)
{
  sc::Mat4 A, B;
  std::memcpy(A.m, a, 64); std::memcpy(B.m, b, 64);
  const sc::Mat4 r = sc::mat4_mul(A, B);
  std::memcpy(out, r.m, 64);
}
void ref_mat4_rotation_xyz(float rx, float ry, float rz, float* out)
{
  const sc::Mat4 r = sc::mat4_rotation_xyz(rx, ry, rz);
  std::memcpy(out, r.m, 64);
}
void ref_mat4_trs(const float* pos, const float* rot, const float* scale, float* out)
{
  const sc::Mat4 r = sc::mat4_trs(pos, rot, scale);
  std::memcpy(out, r.m, 64);
}
void ref_mat4_inverse(const float* a, float* out)
{
  sc::Mat4 A; std::memcpy(A.m, a, 64);
  const sc::Mat4 r = sc::mat4_inverse(A);
  std::memcpy(out, r.m, 64);
}
void ref_mat4_perspective_rh_zo(float fov, float aspect, float zn, float zf, int flipY, float* out)
{
  const sc::Mat4 r = sc::mat4_perspective_rh_zo(fov, aspect, zn, zf, flipY != 0);
  std::memcpy(out, r.m, 64);
}

// Drives the reference's sparse-set pool with a script of operations and returns the dense order.
// ops[i] = +(index+1) -> add Entity(index, gen 0); -(index+1) -> remove it.
uint32_t ref_pool_script(const int32_t* ops, uint32_t nOps, uint32_t* denseOut, uint32_t cap)
{
  sc::ComponentPool<sc::Transform> pool;
  for (uint32_t i = 0; i < nOps; ++i) {
    const int32_t op = ops[i];
    if (op > 0) pool.add(sc::Entity::fromParts((uint32_t)(op - 1), 0));
    else if (op < 0) pool.remove(sc::Entity::fromParts((uint32_t)(-op - 1), 0));
  }
  const auto& d = pool.denseEntities();
  for (uint32_t i = 0; i < d.size() && i < cap; ++i) denseOut[i] = d[i].value;
  return (uint32_t)d.size();
}

uint32_t ref_entity_pack(uint32_t index, uint32_t generation) { return sc::Entity::fromParts(index, generation).value; }

// Layout facts of the types on the path (SURVEY.md section 8 sizes), as the reference's headers define them.
// out: sizeof(Transform), off parent, localPos, localRot, localScale, worldMatrix, dirty,
//      sizeof(Mat4), sizeof(DrawItem), off DrawItem.model, sizeof(Bounds), sizeof(Plane),
//      sizeof(Frustum), sizeof(RenderMesh), sizeof(Entity), sizeof(Camera)
void ref_layout(uint32_t* out)
{
  uint32_t k = 0;
  out[k++] = (uint32_t)sizeof(sc::Transform);
  out[k++] = (uint32_t)offsetof(sc::Transform, parent);
  out[k++] = (uint32_t)offsetof(sc::Transform, localPos);
  out[k++] = (uint32_t)offsetof(sc::Transform, localRot);
  out[k++] = (uint32_t)offsetof(sc::Transform, localScale);
  out[k++] = (uint32_t)offsetof(sc::Transform, worldMatrix);
  out[k++] = (uint32_t)offsetof(sc::Transform, dirty);
  out[k++] = (uint32_t)sizeof(sc::Mat4);
  out[k++] = (uint32_t)sizeof(sc::DrawItem);
  out[k++] = (uint32_t)offsetof(sc::DrawItem, model);
  out[k++] = (uint32_t)sizeof(sc::Bounds);
  out[k++] = (uint32_t)sizeof(sc::Plane);
  out[k++] = (uint32_t)sizeof(sc::Frustum);
  out[k++] = (uint32_t)sizeof(sc::RenderMesh);
  out[k++] = (uint32_t)sizeof(sc::Entity);
  out[k++] = (uint32_t)sizeof(sc::Camera);
}

// Default-constructed Transform as the reference's header initialises it (sc_ecs.h:63-71).
void ref_default_transform(uint32_t* parent, float* pos, float* rot, float* scale, float* world, uint8_t* dirty)
{
  const sc::Transform t{};
  *parent = t.parent.value;
  std::memcpy(pos, t.localPos, 12); std::memcpy(rot, t.localRot, 12); std::memcpy(scale, t.localScale, 12);
  std::memcpy(world, t.worldMatrix.m, 64);
  *dirty = t.dirty ? 1 : 0;
}

} // extern "C"
