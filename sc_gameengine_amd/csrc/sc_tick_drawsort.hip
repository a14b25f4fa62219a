// sc_tick_drawsort.hip -- the renderer's draw order on the device (second half).
//
// VkRenderer::recordCommandBuffer drops draws whose mesh or material handle does not exist and sorts the
// rest by (material's pipeline, material handle, mesh handle) before binding (src/engine/src/sc_vk.cpp:
// 1842-1864).  Here the budgeted visible list is keyed, sorted and only then expanded into 80-byte
// DrawItems, so the items are written once, already in bind order.  std::sort leaves the order of equal
// keys unspecified; this sort is stable (equal keys keep visible-list order), which is one of the orders
// the reference may produce and makes the result reproducible.
//
// Sort: least-significant-digit radix, 8-bit digits, only over the key bytes that can differ (the host knows
// the handle ranges).  A workgroup of 16 waves owns 8192 consecutive keys, each wave 512 of them in index
// order; per-wave digit counts live in LDS as packed 16-bit pairs, so ranks need no barrier inside the
// scatter.  Up to 8192 draws (budgets are 4096-6000 in the reference, sc_world_partition.h:309,
// sandbox main.cpp:96) one workgroup does a whole pass in one launch.
#include "sc_tick_internal.h"

namespace sctick {

namespace {

constexpr uint32_t kWavesPerGroup = kSortThreads / 64;           // 16
constexpr uint32_t kKeysPerWave = kSortGroup / kWavesPerGroup;   // 512
constexpr uint32_t kRounds = kKeysPerWave / 64;                  // 8
constexpr uint32_t kDigits = 256;

__device__ inline uint32_t budgeted(const DeviceState& d, uint32_t budget)
{
  const uint32_t visible = d.counters[0];
  return (budget > 0 && visible > budget) ? budget : visible;   // sc_world_partition.cpp:1306-1312
}

// key = pipeline << 48 | material << 24 | mesh; a draw the renderer would skip gets kDrawInvalid (sorts last)
__global__ __launch_bounds__(kTile) void k_draw_keys(const DeviceState d, const DrawSortState st, uint32_t budget)
{
  const uint32_t visible = d.counters[0];
  const uint32_t emitted = budgeted(d, budget);
  if (blockIdx.x == 0 && threadIdx.x == 0) { d.counters[4] = emitted; d.counters[5] = visible - emitted; }
  for (uint32_t base = blockIdx.x * kTile; base < emitted; base += gridDim.x * kTile) {
    const uint32_t t = base + threadIdx.x;
    bool ok = false;
    if (t < emitted) {
      const uint32_t j = d.visibleIdx[t];
      const uint32_t mesh = d.meshId[j], mat = d.materialId[j];
      uint64_t key = kDrawInvalid;
      if (mesh < st.meshCount && mat < st.materialCount) {
        const uint32_t pipe = st.pipeline[mat];
        if (pipe != kNoMaterial) { ok = true; key = (uint64_t)pipe << 48 | (uint64_t)mat << 24 | mesh; }
      }
      st.key[0][t] = key;
      st.idx[0][t] = t;
    }
    const uint64_t okLanes = ballot64(ok);
    if ((threadIdx.x & 63u) == 0 && okLanes) atomicAdd(&d.counters[kCtrDrawsSorted], (uint32_t)__popcll(okLanes));
  }
}

struct PassArgs {
  const uint64_t* keyIn; const uint32_t* idxIn; uint64_t* keyOut; uint32_t* idxOut;
  const uint32_t* counters; uint32_t shift; uint32_t* hist; uint32_t groups;
};

// digit totals of every workgroup's slice, laid out [digit][group] for the scan
__global__ __launch_bounds__(kSortThreads) void k_radix_hist(const PassArgs a)
{
  __shared__ uint32_t tot[kDigits];
  const uint32_t n = a.counters[4];
  const uint32_t base = blockIdx.x * kSortGroup;
  if (base >= n) return;                         // the scan only visits groups that hold keys
  if (threadIdx.x < kDigits) tot[threadIdx.x] = 0u;
  __syncthreads();
  for (uint32_t r = 0; r < kSortGroup / kSortThreads; ++r) {
    const uint32_t i = base + r * kSortThreads + threadIdx.x;
    if (i < n) atomicAdd(&tot[(uint32_t)(a.keyIn[i] >> a.shift) & 255u], 1u);
  }
  __syncthreads();
  if (threadIdx.x < kDigits) a.hist[threadIdx.x * a.groups + blockIdx.x] = tot[threadIdx.x];
}

// exclusive scan, in place and in [digit][group] order, over the groups that hold keys this tick (one workgroup)
__global__ __launch_bounds__(kSortThreads) void k_radix_scan(uint32_t* hist, const uint32_t* counters, uint32_t groups)
{
  __shared__ uint32_t part[kSortThreads];
  const uint32_t n = counters[4];
  const uint32_t active = min((n + kSortGroup - 1) / kSortGroup, groups);
  const uint32_t entries = kDigits * active;
  const uint32_t per = (entries + kSortThreads - 1) / kSortThreads;
  const uint32_t b = min(threadIdx.x * per, entries), e = min(b + per, entries);
  auto at = [&](uint32_t i) -> uint32_t& { return hist[(i / active) * groups + (i % active)]; };
  uint32_t sum = 0;
  for (uint32_t i = b; i < e; ++i) sum += at(i);
  part[threadIdx.x] = sum;
  __syncthreads();
  for (uint32_t step = 1; step < kSortThreads; step <<= 1) {        // Hillis-Steele over the 1024 partial sums
    const uint32_t v = threadIdx.x >= step ? part[threadIdx.x - step] : 0u;
    __syncthreads();
    part[threadIdx.x] += v;
    __syncthreads();
  }
  uint32_t run = part[threadIdx.x] - sum;
  for (uint32_t i = b; i < e; ++i) { const uint32_t v = at(i); at(i) = run; run += v; }
}

// one stable pass over this workgroup's 8192 keys
template <bool kSingle>
__global__ __launch_bounds__(kSortThreads) void k_radix_pass(const PassArgs a)
{
  __shared__ uint32_t waveCnt[kWavesPerGroup][kDigits / 2];   // two 16-bit counts per word (a wave holds 512 keys)
  __shared__ uint32_t digitBase[kDigits];
  const uint32_t n = a.counters[4];
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  const uint32_t first = blockIdx.x * kSortGroup + wave * kKeysPerWave + lane;
  if (blockIdx.x * kSortGroup >= n) return;

  for (uint32_t i = threadIdx.x; i < kWavesPerGroup * (kDigits / 2); i += kSortThreads) (&waveCnt[0][0])[i] = 0u;
  __syncthreads();

  uint64_t key[kRounds];
  uint32_t digit[kRounds];
#pragma unroll
  for (uint32_t r = 0; r < kRounds; ++r) {
    const uint32_t i = first + r * 64u;
    key[r] = i < n ? a.keyIn[i] : 0ull;
    digit[r] = (uint32_t)(key[r] >> a.shift) & 255u;
    if (i < n) atomicAdd(&waveCnt[wave][digit[r] >> 1], 1u << ((digit[r] & 1u) * 16u));
  }
  __syncthreads();

  // counts -> exclusive prefix over the waves, both halves of a word at once (no field can overflow: <= 8192)
  uint32_t total = 0;
  if (threadIdx.x < kDigits / 2) {
    for (uint32_t w = 0; w < kWavesPerGroup; ++w) { const uint32_t c = waveCnt[w][threadIdx.x]; waveCnt[w][threadIdx.x] = total; total += c; }
    if (kSingle) { digitBase[2u * threadIdx.x] = total & 0xFFFFu; digitBase[2u * threadIdx.x + 1u] = total >> 16; }
  }
  __syncthreads();
  if (kSingle) {
    // exclusive scan of the 256 digit totals by the first wave, four digits per lane
    if (wave == 0) {
      uint32_t v[4], s = 0;
      for (uint32_t q = 0; q < 4; ++q) { v[q] = digitBase[4u * lane + q]; s += v[q]; }
      uint32_t incl = s;
      for (uint32_t off = 1; off < 64; off <<= 1) { const uint32_t up = __shfl_up(incl, off); if (lane >= off) incl += up; }
      uint32_t run = incl - s;
      for (uint32_t q = 0; q < 4; ++q) { digitBase[4u * lane + q] = run; run += v[q]; }
    }
  } else {
    if (threadIdx.x < kDigits) digitBase[threadIdx.x] = a.hist[threadIdx.x * a.groups + blockIdx.x];
  }
  __syncthreads();

  // scatter, round by round inside each wave: rank = digit base + keys of this digit in earlier waves and
  // rounds + peers in lower lanes.  Only this wave touches waveCnt[wave], in program order.
#pragma unroll
  for (uint32_t r = 0; r < kRounds; ++r) {
    const uint32_t i = first + r * 64u;
    const bool live = i < n;
    uint64_t peers = ballot64(live);
#pragma unroll
    for (uint32_t b = 0; b < 8; ++b) {
      const bool bit = (digit[r] >> b) & 1u;
      const uint64_t bal = ballot64(live && bit);
      peers &= bit ? bal : ~bal;
    }
    const uint32_t below = (uint32_t)__popcll(peers & ((1ull << lane) - 1ull));
    const uint32_t before = (waveCnt[wave][digit[r] >> 1] >> ((digit[r] & 1u) * 16u)) & 0xFFFFu;
    __builtin_amdgcn_wave_barrier();
    if (live && below == 0u) atomicAdd(&waveCnt[wave][digit[r] >> 1], (uint32_t)__popcll(peers) << ((digit[r] & 1u) * 16u));
    __builtin_amdgcn_wave_barrier();
    if (live) {
      const uint32_t pos = digitBase[digit[r]] + before + below;
      a.keyOut[pos] = key[r];
      a.idxOut[pos] = a.idxIn[i];
    }
  }
}

struct DrawItem80s { uint32_t dense, mesh, material, pad; float model[16]; };

// DrawItem{entity, mesh, material, worldMatrix} (sc_world_partition.cpp:1314-1321) in sorted order
__global__ __launch_bounds__(kTile) void k_emit_draws_sorted(const DeviceState d, const uint32_t* __restrict__ order, DrawItem80s* __restrict__ items)
{
  const uint32_t count = d.counters[kCtrDrawsSorted];
  for (uint32_t t = blockIdx.x * kTile + threadIdx.x; t < count; t += gridDim.x * kTile) {
    const uint32_t j = d.visibleIdx[order[t]];
    const float4 r0 = d.w0[j], r1 = d.w1[j], r2 = d.w2[j];
    float4* o = reinterpret_cast<float4*>(&items[t]);
    o[0] = make_float4(__uint_as_float(j), __uint_as_float(d.meshId[j]), __uint_as_float(d.materialId[j]), 0.0f);
    o[1] = make_float4(r0.x, r1.x, r2.x, 0.0f);
    o[2] = make_float4(r0.y, r1.y, r2.y, 0.0f);
    o[3] = make_float4(r0.z, r1.z, r2.z, 0.0f);
    o[4] = make_float4(r0.w, r1.w, r2.w, 1.0f);
  }
}

} // namespace

// `bound`: the most draws this tick can emit (budget, else the entity count); it sizes the launches
void launchSortedDraws(const DeviceState& d, const DrawSortState& st, uint32_t budget, uint32_t bound, void* items, hipStream_t s)
{
  hipMemsetAsync(d.counters + kCtrDrawsSorted, 0, 4, s);
  const uint32_t keyBlocks = std::min<uint32_t>((bound + kTile - 1) / kTile, 1024u);
  hipLaunchKernelGGL(k_draw_keys, dim3(std::max(keyBlocks, 1u)), dim3(kTile), 0, s, d, st, budget);
  const uint32_t groups = std::max((bound + kSortGroup - 1) / kSortGroup, 1u);
  uint32_t from = 0;
  for (uint32_t k = 0; k < st.passes; ++k) {
    const PassArgs a = { st.key[from], st.idx[from], st.key[from ^ 1u], st.idx[from ^ 1u], d.counters, st.shift[k], st.hist, groups };
    if (groups == 1) hipLaunchKernelGGL((k_radix_pass<true>), dim3(1), dim3(kSortThreads), 0, s, a);
    else {
      hipLaunchKernelGGL(k_radix_hist, dim3(groups), dim3(kSortThreads), 0, s, a);
      hipLaunchKernelGGL(k_radix_scan, dim3(1), dim3(kSortThreads), 0, s, st.hist, d.counters, groups);
      hipLaunchKernelGGL((k_radix_pass<false>), dim3(groups), dim3(kSortThreads), 0, s, a);
    }
    from ^= 1u;
  }
  hipLaunchKernelGGL(k_emit_draws_sorted, dim3(std::max(keyBlocks, 1u)), dim3(kTile), 0, s, d, st.idx[from], (DrawItem80s*)items);
}

} // namespace sctick
