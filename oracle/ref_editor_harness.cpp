// ref_editor_harness.cpp -- C entry point over the REAL sc::editor::intersectRayAABB, for pinning the oracle's
// ray-box slab test (and through it the product's ray queries and traffic front rays, which restate the same arithmetic).
// TEST INFRASTRUCTURE ONLY.  Built by oracle/Makefile (target `ref`) into oracle/_ref/libsc_ref_editor.so together with the
// reference's own tools/world_editor/editor_core/editor_core.cpp and src/core/src/sc_math.cpp, both compiled UNMODIFIED from
// where they lie, against the reference's own headers only -- no stand-in header, library or generated file is involved.
// editor_core.cpp also refers to the renderer's C ABI (scRender*), which is not built here: the shared object carries those
// names as undefined symbols that nothing on this path calls, so it must be opened with lazy binding (oracle/ref_ray_driver.cpp
// does; Python's ctypes forces RTLD_NOW and cannot).
#include "editor_core.h"

extern "C" int ref_intersect_ray_aabb(const float* origin3, const float* dir3, const float* bmin3, const float* bmax3, float* tOut)
{
  sc::editor::Ray ray;
  for (int k = 0; k < 3; ++k) { ray.origin[k] = origin3[k]; ray.dir[k] = dir3[k]; }
  return sc::editor::intersectRayAABB(ray, bmin3, bmax3, tOut) ? 1 : 0;      // tools/world_editor/editor_core/editor_core.cpp:438-469
}
