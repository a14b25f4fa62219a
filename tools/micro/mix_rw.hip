// The fused kernel's REAL read : write mix as plain streams (PMC, profiles/r02: 70 MB in, 98 MB out per launch at 1M entities):
// per entity 16 dwords (or 4 float4) in, three float4 rows + a 32-byte bin record + 8 bytes out.  What does the memory
// system give this mix (a) with one entity per thread and one tile per workgroup, (b) with the kernel's span loop (1536
// workgroups walking their tiles one after the other), (c) with a returning atomic in front of the record store?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t err_ = (x); if (err_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(err_)); return 1; } } while (0)
template <bool kF4, bool kAtomic>
__device__ __forceinline__ void one(const float* __restrict__ in, float4* __restrict__ rows, float4* __restrict__ bins, uint32_t* __restrict__ ctr, float2* __restrict__ tail, uint32_t i, uint32_t stride)
{
  float v[16];
  if (kF4) {
    const float4* in4 = reinterpret_cast<const float4*>(in);
#pragma unroll
    for (int k = 0; k < 4; ++k) { const float4 q = in4[(size_t)k * stride + i]; v[4 * k] = q.x; v[4 * k + 1] = q.y; v[4 * k + 2] = q.z; v[4 * k + 3] = q.w; }
  } else {
#pragma unroll
    for (int k = 0; k < 16; ++k) v[k] = in[(size_t)k * stride + i];
  }
  float s = 0.0f;
#pragma unroll
  for (int k = 12; k < 16; ++k) s += v[k];
  rows[i] = make_float4(v[0], v[1], v[2], v[3] + s);
  rows[(size_t)stride + i] = make_float4(v[4], v[5], v[6], v[7]);
  rows[2 * (size_t)stride + i] = make_float4(v[8], v[9], v[10], v[11]);
  uint32_t slot = i;
  if (kAtomic) {                               // one reservation per 16 entities, as the sector runs of the fused kernel
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t base = 0;
    if ((lane & 15u) == 0u) base = atomicAdd(&ctr[i >> 4], 16u);
    base = __shfl(base, lane & ~15u, 64);
    slot = (i & ~15u) + ((base + lane) & 15u);
  }
  bins[2 * (size_t)slot] = make_float4(v[0] + s, v[1], v[2], v[3]);
  bins[2 * (size_t)slot + 1] = make_float4(v[4] + s, v[5], v[6], v[7]);
  tail[i] = make_float2(s, v[0] + s);
}
template <bool kF4, bool kAtomic>
__global__ __launch_bounds__(256) void k_oneshot(const float* in, float4* rows, float4* bins, uint32_t* ctr, float2* tail, uint32_t n, uint32_t stride)
{
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i < n) one<kF4, kAtomic>(in, rows, bins, ctr, tail, i, stride);
}
template <bool kF4, bool kAtomic>
__global__ __launch_bounds__(256) void k_span(const float* in, float4* rows, float4* bins, uint32_t* ctr, float2* tail, uint32_t n, uint32_t stride, uint32_t span)
{
  const uint32_t begin = blockIdx.x * span, end = begin + span < n ? begin + span : n;
  for (uint32_t base = begin; base < end; base += 256u) { const uint32_t i = base + threadIdx.x; if (i < n) one<kF4, kAtomic>(in, rows, bins, ctr, tail, i, stride); }
}
int main()
{
  const uint32_t n = 1u << 20, stride = n + 4096u;
  float* in; CK(hipMalloc(&in, (size_t)16 * stride * 4)); CK(hipMemset(in, 0, (size_t)16 * stride * 4));
  float4 *rows, *bins; float2* tail; uint32_t* ctr;
  CK(hipMalloc(&rows, (size_t)3 * stride * 16)); CK(hipMalloc(&bins, (size_t)2 * stride * 16)); CK(hipMalloc(&tail, (size_t)stride * 8)); CK(hipMalloc(&ctr, (size_t)stride / 4)); CK(hipMemset(ctr, 0, (size_t)stride / 4));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const double moved = (double)n * (64.0 + 48.0 + 32.0 + 8.0);
  auto run = [&](const char* name, auto launch) {
    std::vector<float> ms;
    for (int r = 0; r < 60; ++r) { hipEventRecord(e0, 0); launch(); hipEventRecord(e1, 0); hipEventSynchronize(e1); float t; hipEventElapsedTime(&t, e0, e1); if (r >= 10) ms.push_back(t); }
    std::sort(ms.begin(), ms.end());
    std::printf("%-58s %7.1f GB/s (%.1f us median, %.1f min)\n", name, moved / (ms[ms.size() / 2] * 1e-3) / 1e9, ms[ms.size() / 2] * 1e3, ms[0] * 1e3);
  };
  const uint32_t span = 768, g = (n + span - 1) / span;
  run("64 B in (16 dwords), 88 B out: one tile per workgroup", [&] { hipLaunchKernelGGL((k_oneshot<false, false>), dim3(n / 256), dim3(256), 0, 0, in, rows, bins, ctr, tail, n, stride); });
  run("64 B in (4 float4), 88 B out: one tile per workgroup", [&] { hipLaunchKernelGGL((k_oneshot<true, false>), dim3(n / 256), dim3(256), 0, 0, in, rows, bins, ctr, tail, n, stride); });
  run("16 dwords in: spans of 3 tiles, 1366 workgroups", [&] { hipLaunchKernelGGL((k_span<false, false>), dim3(g), dim3(256), 0, 0, in, rows, bins, ctr, tail, n, stride, span); });
  run("4 float4 in: spans of 3 tiles, 1366 workgroups", [&] { hipLaunchKernelGGL((k_span<true, false>), dim3(g), dim3(256), 0, 0, in, rows, bins, ctr, tail, n, stride, span); });
  run("16 dwords in + returning atomic per 16: one tile / wg", [&] { hipLaunchKernelGGL((k_oneshot<false, true>), dim3(n / 256), dim3(256), 0, 0, in, rows, bins, ctr, tail, n, stride); });
  run("16 dwords in + returning atomic per 16: spans", [&] { hipLaunchKernelGGL((k_span<false, true>), dim3(g), dim3(256), 0, 0, in, rows, bins, ctr, tail, n, stride, span); });
  run("4 float4 in + returning atomic per 16: one tile / wg", [&] { hipLaunchKernelGGL((k_oneshot<true, true>), dim3(n / 256), dim3(256), 0, 0, in, rows, bins, ctr, tail, n, stride); });
  run("16 dwords in, one tile per workgroup (again)", [&] { hipLaunchKernelGGL((k_oneshot<false, false>), dim3(n / 256), dim3(256), 0, 0, in, rows, bins, ctr, tail, n, stride); });
  return 0;
}
