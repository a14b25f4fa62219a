// test_tile_host.cpp -- a NATIVE host for the tiled tick: no Python, no torch.  One process per GPU (RANK / WORLD_SIZE /
// LOCAL_RANK from the environment, as any launcher sets them); the only thing the ranks hand each other outside the
// library is rank 0's 128-byte communicator id, here through a file (SC_TICK_ID_FILE).  Everything else -- communicator,
// border buffers, the RCCL group, the pipelined pair half -- is the library's (include/sc_tick.h, "the exchange itself").
// With WORLD_SIZE = 1 the tile is the centre of a 3x3 world whose eight neighbours are the rank itself (loop-back), which
// is what a one-GPU box can run.  After the steps the tile's visible list and every world matrix are compared with the
// oracle's tick on the same tile.  TEST INFRASTRUCTURE (links oracle/liboracle.so as the checker).  Exit code 0 = pass.
#include "sc_tick.h"
#include "sc_oracle.h"

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <string>
#include <thread>
#include <vector>

static uint32_t gState = 2463534242u;
static float rnd(float a, float b) { gState ^= gState << 13; gState ^= gState >> 17; gState ^= gState << 5; return a + (b - a) * (float)(gState & 0xFFFFFF) / 16777215.0f; }
static int envInt(const char* k, int d) { const char* v = std::getenv(k); return v ? std::atoi(v) : d; }
#define REQUIRE(cond, ...) do { if (!(cond)) { std::printf("FAIL %s:%d: ", __FILE__, __LINE__); std::printf(__VA_ARGS__); std::printf("\n"); return 1; } } while (0)

int main(int argc, char** argv)
{
  const int S = argc > 1 ? std::atoi(argv[1]) : 32;                  // tile side in sectors
  const int steps = argc > 2 ? std::atoi(argv[2]) : 40;
  const int world = envInt("WORLD_SIZE", 1), rank = envInt("RANK", 0), local = envInt("LOCAL_RANK", 0);
  const uint32_t tilesX = world == 1 ? 3u : (world == 2 ? 2u : (world == 4 ? 2u : 4u)), tilesZ = world == 1 ? 3u : (world <= 2 ? 1u : 2u);
  REQUIRE(world == 1 || tilesX * tilesZ == (uint32_t)world, "world size must be 1, 2, 4 or 8");
  const uint32_t tx = world == 1 ? 1u : (uint32_t)rank % tilesX, tz = world == 1 ? 1u : (uint32_t)rank / tilesX;
  gState += 977u * (uint32_t)rank;

  // ---- the tile's entities: per sector a ground slab and 15 props, one of them a dynamic body; every fourth prop a child
  const uint32_t per = 16, n = (uint32_t)(S * S) * per;
  std::vector<float> pos(3 * n), rot(3 * n, 0.0f), scl(3 * n);
  std::vector<int32_t> parent(n, SC_TICK_NO_PARENT);
  std::vector<uint32_t> group(n, 2u), mask(n, 1u);
  const float ox = (float)(tx * S) * 64.0f, oz = (float)(tz * S) * 64.0f;
  for (int sz = 0; sz < S; ++sz) for (int sx = 0; sx < S; ++sx) for (uint32_t k = 0; k < per; ++k) {
    const uint32_t i = ((uint32_t)(sz * S + sx)) * per + k;
    const float mx = ox + 64.0f * (float)sx, mz = oz + 64.0f * (float)sz;
    if (k == 0) { pos[3 * i] = mx + 32.0f; pos[3 * i + 1] = -0.55f; pos[3 * i + 2] = mz + 32.0f; scl[3 * i] = 64.0f; scl[3 * i + 1] = 0.1f; scl[3 * i + 2] = 64.0f; continue; }
    scl[3 * i] = rnd(0.4f, 1.9f); scl[3 * i + 1] = rnd(0.5f, 3.2f); scl[3 * i + 2] = rnd(0.4f, 1.9f);
    pos[3 * i] = rnd(mx - 0.5f, mx + 64.5f); pos[3 * i + 1] = scl[3 * i + 1] * 0.5f; pos[3 * i + 2] = rnd(mz - 0.5f, mz + 64.5f);   // some straddle the sector's (and the tile's) edges
    rot[3 * i + 1] = rnd(0.0f, 6.28f);
    if (k % 4 == 2) { parent[i] = (int32_t)i - 1; pos[3 * i] = rnd(-1, 1); pos[3 * i + 1] = rnd(0, 1); pos[3 * i + 2] = rnd(-1, 1); }
    if (k == 5) { group[i] = 1u; mask[i] = 0xFFFFFFFFu; }
  }

  // ---- context + communicator
  ScTickContextDesc desc{};
  desc.device_ordinal = local; desc.capacity = n; desc.tile_origin_x = (int32_t)(tx * S); desc.tile_origin_z = (int32_t)(tz * S);
  desc.tile_sectors_x = desc.tile_sectors_z = (uint32_t)S; desc.sector_size = 64.0f;
  ScTickContext* ctx = scTickCreateContext(&desc);
  REQUIRE(ctx, "scTickCreateContext: %s", scTickGetLastError(nullptr));
  auto ok = [&](int r, const char* what) { if (!r) std::printf("FAIL %s: %s\n", what, scTickGetLastError(ctx)); return r != 0; };
  if (!ok(scTickSetEntityCount(ctx, n), "count") || !ok(scTickUploadLocals(ctx, 0, n, pos.data(), rot.data(), scl.data(), nullptr), "locals")) return 1;
  std::vector<float> bmin(3 * n, -0.5f), bmax(3 * n, 0.5f);
  if (!ok(scTickUploadBounds(ctx, 0, n, bmin.data(), bmax.data(), nullptr), "bounds") || !ok(scTickUploadRenderMeshes(ctx, 0, n, nullptr, nullptr, nullptr), "meshes") ||
      !ok(scTickUploadLayers(ctx, 0, n, group.data(), mask.data()), "layers") || !ok(scTickSetTopology(ctx, parent.data(), n), "topology")) return 1;
  if (!ok(scTickSetTile(ctx, world == 1 ? 4u : (uint32_t)rank, 0), "tile") || !ok(scTickSetTileGrid(ctx, tx, tz, tilesX, tilesZ), "grid")) return 1;

  uint8_t id[SC_TICK_COMM_ID_BYTES];
  const char* idFile = std::getenv("SC_TICK_ID_FILE");
  if (rank == 0) {
    if (!ok(scTickCommGetUniqueId(id), "unique id")) { std::printf("%s\n", scTickGetLastError(nullptr)); return 1; }
    if (world > 1) {
      REQUIRE(idFile, "SC_TICK_ID_FILE must name a file every rank can reach");
      const std::string tmp = std::string(idFile) + ".tmp";
      { std::ofstream f(tmp, std::ios::binary); f.write(reinterpret_cast<const char*>(id), sizeof id); }
      std::rename(tmp.c_str(), idFile);
    }
  } else {
    REQUIRE(idFile, "SC_TICK_ID_FILE must name a file every rank can reach");
    bool got = false;
    for (int tries = 0; tries < 3000 && !got; ++tries) {
      std::ifstream f(idFile, std::ios::binary);
      if (f && f.read(reinterpret_cast<char*>(id), sizeof id)) got = true; else std::this_thread::sleep_for(std::chrono::milliseconds(10));
    }
    REQUIRE(got, "rank %d: no communicator id in %s", rank, idFile);
  }
  if (world == 1) { const int32_t self[8] = { 0, 0, 0, 0, 0, 0, 0, 0 }; if (!ok(scTickCommSetPeers(ctx, self), "peers")) return 1; }
  if (!ok(scTickCommInit(ctx, id, (uint32_t)world, (uint32_t)rank), "scTickCommInit") || !ok(scTickSetPipelined(ctx, 1), "pipelined")) return 1;

  // ---- camera over the middle of the world, frame producer, steps
  const float W = 64.0f * (float)(tilesX * S), H = 64.0f * (float)(tilesZ * S);
  const float camPos[3] = { W * 0.5f, 30.0f, H * 0.5f }, camRot[3] = { -0.3f, 0.7f, 0.0f }, one[3] = { 1, 1, 1 };
  float camWorld[16], vp[16];
  scTickHostMat4Trs(camPos, camRot, one, camWorld);
  scTickHostCameraViewProj(camWorld, 60.0f, 16.0f / 9.0f, 0.1f, 1000.0f, vp);
  if (!ok(scTickSetViewProj(ctx, vp), "viewProj") || !ok(scTickSetFrameProducer(ctx, 1, 0.01f), "producer") || !ok(scTickNudgeRootsX(ctx, 0.01f), "first nudge")) return 1;
  const uint32_t flags = SC_TICK_FULL | SC_TICK_PRODUCE_NEXT;
  for (int k = 0; k < 10; ++k) if (!ok(scTickTileStep(ctx, flags), "scTickTileStep")) return 1;
  scTickSynchronize(ctx);
  const auto t0 = std::chrono::steady_clock::now();
  for (int k = 0; k < steps; ++k) if (!ok(scTickTileStep(ctx, flags), "scTickTileStep")) return 1;
  const double issued = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
  scTickSynchronize(ctx);
  const double total = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();

  // ---- the checker: the oracle ticks the same tile (10 + steps producer applications, one after the other)
  OrcWorld* ow = orc_world_new();
  std::vector<uint8_t> all(n, 1);
  orc_world_build(ow, n, pos.data(), rot.data(), scl.data(), parent.data(), all.data(), nullptr, nullptr, all.data(), bmin.data(), bmax.data());
  for (int k = 0; k < 10 + steps; ++k) orc_nudge_roots_x(ow, 0.01f);
  orc_transform_system(ow);
  OrcCullingState* cs = orc_culling_state_new();
  orc_culling_system(ow, cs, vp);
  std::vector<uint32_t> vis(n); uint32_t nv = 0;
  if (!ok(scTickReadVisible(ctx, vis.data(), n, &nv), "read visible")) return 1;
  REQUIRE(nv == cs->visibleLen, "visible count %u vs oracle %u", nv, cs->visibleLen);
  REQUIRE(std::memcmp(vis.data(), cs->visible, (size_t)nv * 4u) == 0, "visible list differs from the oracle's");
  // result assembly: every rank's visible count over the library's communicator -> this rank's offset in the global
  // list (tiles in rank order, entities tile-major: the concatenation is the reference's compaction order) and the list's length
  std::vector<uint32_t> counts((size_t)world, 0u);
  uint64_t offset = 0, totalVisible = 0;
  if (!ok(scTickGatherVisibleCounts(ctx, counts.data(), (uint32_t)world, &offset, &totalVisible), "scTickGatherVisibleCounts")) return 1;
  REQUIRE(counts[(size_t)rank] == nv, "the gathered count of this rank is %u, its list holds %u", counts[(size_t)rank], nv);
  uint64_t before = 0, sum = 0;
  for (int r = 0; r < world; ++r) { if (r < rank) before += counts[(size_t)r]; sum += counts[(size_t)r]; }
  REQUIRE(offset == before && totalVisible == sum, "offset %llu / total %llu do not follow from the counts", (unsigned long long)offset, (unsigned long long)totalVisible);
  std::vector<float> got(16 * (size_t)n), want(16 * (size_t)n);
  if (!ok(scTickReadWorldMatrices(ctx, 0, n, got.data()), "read matrices")) return 1;
  orc_read_world_matrices(ow, want.data());
  size_t bad = 0;
  for (size_t i = 0; i < got.size(); ++i) if (!(got[i] == want[i])) ++bad;
  REQUIRE(bad == 0, "%zu matrix elements differ from the oracle's", bad);
  ScTickCounts c{};
  scTickGetCounts(ctx, &c);
  REQUIRE(c.border_lost == 0, "border_lost %u", c.border_lost);
  std::printf("{\"rank\": %d, \"world_size\": %d, \"entities\": %u, \"steps\": %d, \"us_per_step\": %.2f, \"host_issue_us\": %.2f, \"visible\": %u, \"visible_offset\": %llu, \"visible_total\": %llu, \"pairs\": %u}\n",
              rank, world, n, steps, total / steps, issued / steps, nv, (unsigned long long)offset, (unsigned long long)totalVisible, c.pairs);
  std::printf("all checks passed\n");
  orc_culling_state_free(cs); orc_world_free(ow);
  scTickCommDestroy(ctx);
  scTickDestroyContext(ctx);
  return 0;
}
