// ref_sector_harness.cpp -- C entry points over the REAL reference sector-format code
// (tools/shared/world_format.cpp, compiled unmodified from where it lies and linked into
// oracle/_ref/libsc_ref.so).  TEST INFRASTRUCTURE ONLY: it writes the golden .scsector fixtures
// (oracle/make_golden.py) with the reference's own WriteSectorFile and reads files back with its
// own ReadSectorFile, so the product's reader (sc_tick_sector.cpp) is pinned against both.
#include "world_format.h"

#include <cstdint>
#include <cstring>
#include <string>

extern "C" {

// Instances from flat arrays; lanes / spawners / colliders are filled with a fixed arithmetic pattern
// (they only have to be there for the reader to walk past).
int ref_sector_write(const char* path, uint32_t version, int32_t sx, int32_t sz, uint32_t n,
                     const uint64_t* id, const uint64_t* model, const uint64_t* mesh, const uint64_t* material,
                     const uint64_t* albedo, const uint32_t* matFlags, const uint32_t* tags,
                     const float* trs9, const char* name64,
                     uint32_t nLanes, uint32_t pointsPerLane, uint32_t nSpawners, uint32_t nColliders)
{
  sc_world::SectorFile f{};
  f.version = version;
  f.sector = { sx, sz };
  f.instances.resize(n);
  for (uint32_t i = 0; i < n; ++i) {
    sc_world::Instance& in = f.instances[i];
    in.id = id[i]; in.model_id = model[i]; in.mesh_id = mesh[i]; in.material_id = material[i];
    in.albedo_texture_id = albedo[i]; in.material_flags = matFlags[i]; in.tags = tags[i];
    std::memcpy(in.transform.position, trs9 + 9 * i, 12);
    std::memcpy(in.transform.rotation, trs9 + 9 * i + 3, 12);
    std::memcpy(in.transform.scale, trs9 + 9 * i + 6, 12);
    std::memcpy(in.name, name64 + 64 * i, 64);
  }
  f.lanes.resize(nLanes);
  for (uint32_t i = 0; i < nLanes; ++i) {
    f.lanes[i].id = 1000u + i; f.lanes[i].flags = i & 3u;
    f.lanes[i].points.resize(pointsPerLane + i);
    for (uint32_t k = 0; k < f.lanes[i].points.size(); ++k) f.lanes[i].points[k] = { (float)i, 0.25f * (float)k, (float)(i + k) };
  }
  f.spawners.resize(nSpawners);
  for (uint32_t i = 0; i < nSpawners; ++i) { f.spawners[i].id = 2000u + i; f.spawners[i].type = i; f.spawners[i].rate = 0.5f + (float)i; }
  f.colliders.resize(nColliders);
  for (uint32_t i = 0; i < nColliders; ++i) { f.colliders[i].id = 3000u + i; f.colliders[i].shape = i % 3u; f.colliders[i].size[1] = 2.0f + (float)i; }
  return sc_world::WriteSectorFile(path, f) ? 1 : 0;
}

// counts4: instances, lanes, spawners, colliders; lanePoints: total points over all lanes
int ref_sector_read(const char* path, uint32_t* version, int32_t* sxz, uint32_t* counts4, uint32_t* lanePoints, uint32_t cap,
                    uint64_t* id, uint64_t* model, uint64_t* mesh, uint64_t* material,
                    uint64_t* albedo, uint32_t* matFlags, uint32_t* tags, float* trs9, char* name64)
{
  sc_world::SectorFile f{};
  if (!sc_world::ReadSectorFile(path, &f)) return 0;
  *version = f.version; sxz[0] = f.sector.x; sxz[1] = f.sector.z;
  counts4[0] = (uint32_t)f.instances.size(); counts4[1] = (uint32_t)f.lanes.size();
  counts4[2] = (uint32_t)f.spawners.size(); counts4[3] = (uint32_t)f.colliders.size();
  uint32_t pts = 0;
  for (const auto& l : f.lanes) pts += (uint32_t)l.points.size();
  *lanePoints = pts;
  for (uint32_t i = 0; i < f.instances.size() && i < cap; ++i) {
    const sc_world::Instance& in = f.instances[i];
    id[i] = in.id; model[i] = in.model_id; mesh[i] = in.mesh_id; material[i] = in.material_id;
    albedo[i] = in.albedo_texture_id; matFlags[i] = in.material_flags; tags[i] = in.tags;
    std::memcpy(trs9 + 9 * i, in.transform.position, 12);
    std::memcpy(trs9 + 9 * i + 3, in.transform.rotation, 12);
    std::memcpy(trs9 + 9 * i + 6, in.transform.scale, 12);
    std::memcpy(name64 + 64 * i, in.name, 64);
  }
  return 1;
}

uint64_t ref_hash_asset_path(const char* path) { return sc_world::HashAssetPath(path); }

uint32_t ref_sector_path(const char* root, int32_t x, int32_t z, char* out, uint32_t cap)
{
  const std::string s = sc_world::BuildSectorPath(root, { x, z });
  if (out && cap) { const size_t k = s.size() < cap - 1 ? s.size() : cap - 1; std::memcpy(out, s.data(), k); out[k] = 0; }
  return (uint32_t)s.size();
}

} // extern "C"
