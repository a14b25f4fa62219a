// sc_tick_sector.cpp -- host-side reader of the .scsector format into SoA arrays (no GPU work).
//
// Behaviour follows sc_world::ReadSectorFile (tools/shared/world_format.cpp:185-338) over a byte
// buffer instead of an ifstream.  The one subtle part is what an ifstream does when the data runs
// out: the short read copies the bytes that are there, sets failbit, and every later read or seek
// is a no-op -- so the rest of the records keep their default-constructed values (world_format.h:
// 31-47: id 0, scale 1, empty name).  `Cursor` models exactly that.
#include "../../include/sc_tick.h"

#include <cctype>
#include <cstdio>
#include <cstring>
#include <filesystem>
#include <string>
#include <vector>

namespace {

constexpr uint32_t fourcc(char a, char b, char c, char d)
{
  return (uint32_t)(uint8_t)a | ((uint32_t)(uint8_t)b << 8) | ((uint32_t)(uint8_t)c << 16) | ((uint32_t)(uint8_t)d << 24);
}
constexpr uint32_t kMagicSect = fourcc('S', 'E', 'C', 'T');      // kSectorMagic 0x54434553 (world_format.h:11)
constexpr uint32_t kInst = fourcc('I', 'N', 'S', 'T'), kLane = fourcc('L', 'A', 'N', 'E');
constexpr uint32_t kSpwn = fourcc('S', 'P', 'W', 'N'), kColl = fourcc('C', 'O', 'L', 'L');
constexpr uint32_t kNameBytes = 64;                               // kInstanceNameMax
constexpr uint32_t kTrsBytes = 36;                                // sc_world::Transform: 9 floats
constexpr uint32_t kRecV3 = 8 + 8 + 8 + kTrsBytes + 4;            // id, mesh, material, transform, tags
constexpr uint32_t kOverrideBytes = 8 + 4;                        // albedo_texture_id, material_flags

struct Cursor
{
  const uint8_t* p; uint64_t size, at = 0;
  bool failed = false;      // failbit: sticky
  bool shortRead = false;   // a read other than a chunk header at the exact end came up short
  uint64_t left() const { return at < size ? size - at : 0; }
  // istream::read into a value the caller has already set to its default
  bool take(void* dst, uint64_t n)
  {
    if (failed) return false;
    const uint64_t have = left();
    if (have < n) { if (have) std::memcpy(dst, p + at, have); at = size; failed = true; shortRead = true; return false; }
    std::memcpy(dst, p + at, n); at += n;
    return true;
  }
  template <typename T> bool val(T& v) { return take(&v, sizeof(T)); }
  void skip(uint64_t n) { if (!failed) at += n; }      // seekg(cur): may land past the end; the next read then fails
};

struct Record
{
  uint64_t id = 0, model = 0, mesh = 0, material = 0, albedo = 0;
  uint32_t matFlags = 0, tags = 0;
  float trs[9] = { 0, 0, 0, 0, 0, 0, 1, 1, 1 };
  char name[kNameBytes] = {};
};

void store(const ScTickSectorInstances* o, uint32_t i, const Record& r)
{
  if (!o || i >= o->capacity) return;
  if (o->id) o->id[i] = r.id;
  if (o->model_id) o->model_id[i] = r.model;
  if (o->mesh_id) o->mesh_id[i] = r.mesh;
  if (o->material_id) o->material_id[i] = r.material;
  if (o->albedo_texture_id) o->albedo_texture_id[i] = r.albedo;
  if (o->material_flags) o->material_flags[i] = r.matFlags;
  if (o->tags) o->tags[i] = r.tags;
  if (o->pos3) std::memcpy(o->pos3 + 3 * (size_t)i, r.trs, 12);
  if (o->rot3) std::memcpy(o->rot3 + 3 * (size_t)i, r.trs + 3, 12);
  if (o->scale3) std::memcpy(o->scale3 + 3 * (size_t)i, r.trs + 6, 12);
  if (o->name64) std::memcpy(o->name64 + (size_t)kNameBytes * i, r.name, kNameBytes);
}

void parseInstances(Cursor& c, uint32_t version, uint32_t chunkSize, ScTickSectorInfo& info, const ScTickSectorInstances* out)
{
  uint32_t count = 0;
  c.val(count);
  // record length is what the chunk says it is (:214-216); fields present are decided from it (:218-224)
  uint32_t rec = kRecV3;
  if (count > 0 && chunkSize >= 4) rec = (chunkSize - 4) / count;
  const bool hasModel = version >= 4;
  const uint32_t fixed = kRecV3 + (hasModel ? 8u : 0u);
  const bool hasName = rec >= fixed + kNameBytes;
  const uint32_t withName = fixed + (hasName ? kNameBytes : 0u);
  const bool hasOverrides = rec >= withName + kOverrideBytes;
  const uint32_t known = withName + (hasOverrides ? kOverrideBytes : 0u);
  const uint32_t pad = rec > known ? rec - known : 0u;

  info.instances = count;                         // a later INST chunk replaces an earlier one (resize + overwrite)
  const uint32_t keep = out ? (count < out->capacity ? count : out->capacity) : 0u;
  for (uint32_t i = 0; i < count; ++i) {
    if (c.failed && i >= keep) break;             // nothing left to read and nothing left to store
    Record r;
    c.val(r.id);
    if (hasModel) c.val(r.model);
    c.val(r.mesh);
    c.val(r.material);
    c.take(r.trs, kTrsBytes);
    if (hasName) { c.take(r.name, kNameBytes); r.name[kNameBytes - 1] = '\0'; }
    c.val(r.tags);
    if (hasOverrides) { c.val(r.albedo); c.val(r.matFlags); }
    c.skip(pad);
    store(out, i, r);
  }
}

void parseLanes(Cursor& c, ScTickSectorInfo& info)
{
  uint32_t count = 0;
  c.val(count);
  info.lanes = count; info.lane_points = 0;
  for (uint32_t i = 0; i < count && !c.failed; ++i) {
    uint64_t id = 0; uint32_t flags = 0, points = 0;
    c.val(id); c.val(flags); c.val(points);
    info.lane_points += points;
    for (uint32_t k = 0; k < points && !c.failed; ++k) { float xyz[3]; c.take(xyz, 12); }
  }
}

void parseFixed(Cursor& c, uint32_t recordBytes, uint32_t& countOut)
{
  uint32_t count = 0;
  c.val(count);
  countOut = count;
  uint8_t scratch[64];
  for (uint32_t i = 0; i < count && !c.failed; ++i) c.take(scratch, recordBytes);
}

int parse(const uint8_t* data, uint64_t size, ScTickSectorInfo* info, const ScTickSectorInstances* out)
{
  if (!data || !info) return 0;
  std::memset(info, 0, sizeof *info);
  Cursor c{ data, size };
  uint32_t magic = 0;
  c.val(magic);
  if (magic != kMagicSect) return 0;
  c.val(info->version);
  c.val(info->sector_x);
  c.val(info->sector_z);

  while (!c.failed) {
    if (c.left() == 0) break;                      // clean end of file at a chunk boundary
    struct { uint32_t id = 0, size = 0; } ch;
    if (!c.take(&ch, 8)) break;
    if (ch.size == 0) continue;
    if (ch.id == kInst) parseInstances(c, info->version, ch.size, *info, out);
    else if (ch.id == kLane) parseLanes(c, *info);
    else if (ch.id == kSpwn) parseFixed(c, 8 + kTrsBytes + 4 + 4, info->spawners);     // id, transform, type, rate
    else if (ch.id == kColl) parseFixed(c, 8 + 4 + kTrsBytes + 12, info->colliders);   // id, shape, transform, size[3]
    else c.skip(ch.size);
  }
  info->truncated = c.shortRead ? 1u : 0u;
  return 1;
}

} // namespace

extern "C" {

int scTickSectorParse(const void* data, uint64_t size, ScTickSectorInfo* info, const ScTickSectorInstances* out)
{
  return parse(static_cast<const uint8_t*>(data), size, info, out);
}

int scTickSectorReadFile(const char* path, ScTickSectorInfo* info, const ScTickSectorInstances* out)
{
  if (!path || !info) return 0;
  std::FILE* f = std::fopen(path, "rb");
  if (!f) return 0;
  std::vector<uint8_t> bytes;
  uint8_t chunk[1 << 16];
  size_t got;
  while ((got = std::fread(chunk, 1, sizeof chunk, f)) > 0) bytes.insert(bytes.end(), chunk, chunk + got);
  std::fclose(f);
  static const uint8_t none = 0;
  return parse(bytes.empty() ? &none : bytes.data(), bytes.size(), info, out);
}

uint64_t scTickHashAssetPath(const char* path)
{
  if (!path) return 0;
  const std::string text = std::filesystem::path(path).lexically_normal().generic_string();
  // FNV-1a 64 with the reference's own starting value (world_format.cpp:66): it is NOT the standard offset
  // basis 14695981039346656037 -- the last digit is missing there, and asset ids on disk depend on it
  uint64_t h = 1469598103934665603ull;
  for (unsigned char ch : text) { h ^= (uint64_t)(unsigned char)std::tolower(ch); h *= 0x100000001b3ull; }
  return h;
}

uint32_t scTickSectorPath(const char* world_root, int32_t x, int32_t z, char* out, uint32_t capacity)
{
  std::filesystem::path p(world_root ? world_root : ".");
  p /= "sectors";
  p /= "sector_" + std::to_string(x) + "_" + std::to_string(z) + ".scsector";
  const std::string s = p.string();
  if (out && capacity) { const size_t k = s.size() < capacity - 1 ? s.size() : capacity - 1; std::memcpy(out, s.data(), k); out[k] = '\0'; }
  return (uint32_t)s.size();
}

} // extern "C"
