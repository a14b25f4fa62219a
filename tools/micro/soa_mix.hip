// The fused kernel's traffic (1M entities: 112 B read, 56 B written each) as plain streams, two layouts:
//   dword SoA  -- 28 arrays of one float per entity in, 3 float4 rows + 2 dwords out (what the kernel does)
//   float4 AoSoA -- the same bytes as 7 float4 arrays in, 3 float4 + 1 float2 out
// Repeated launches over the same 176 MB, like ticks.  What rate does the memory system give each?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t err_ = (x); if (err_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(err_)); return 1; } } while (0)
template <int NIN>
__global__ __launch_bounds__(256) void k_dword(const float* __restrict__ in, float4* __restrict__ o0, float4* __restrict__ o1, float4* __restrict__ o2, float* __restrict__ o3, float* __restrict__ o4, uint32_t n, uint32_t stride)
{
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i >= n) return;
  float v[NIN];
#pragma unroll
  for (int k = 0; k < NIN; ++k) v[k] = in[(size_t)k * stride + i];
  float s = 0.0f;
#pragma unroll
  for (int k = 12; k < NIN; ++k) s += v[k];
  o0[i] = make_float4(v[0], v[1], v[2], v[3]);
  o1[i] = make_float4(v[4], v[5], v[6], v[7]);
  o2[i] = make_float4(v[8] + s, v[9], v[10], v[11]);
  o3[i] = s; o4[i] = v[0] + s;
}
__global__ __launch_bounds__(256) void k_float4(const float4* __restrict__ in, float4* __restrict__ o0, float4* __restrict__ o1, float4* __restrict__ o2, float2* __restrict__ o3, uint32_t n, uint32_t stride)
{
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i >= n) return;
  float4 v[7];
#pragma unroll
  for (int k = 0; k < 7; ++k) v[k] = in[(size_t)k * stride + i];
  const float s = v[3].x + v[4].y + v[5].z + v[6].w;
  o0[i] = v[0]; o1[i] = v[1]; o2[i] = make_float4(v[2].x + s, v[2].y, v[2].z, v[2].w);
  o3[i] = make_float2(s, v[0].x + s);
}
// dword SoA layout, but a lane takes FOUR consecutive entities: every request is 16 B per lane, 28 streams
__global__ __launch_bounds__(256) void k_dword_x4(const float4* __restrict__ in, float4* __restrict__ o0, float4* __restrict__ o1, float4* __restrict__ o2, float4* __restrict__ o3, float4* __restrict__ o4, uint32_t n4, uint32_t stride4)
{
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i >= n4) return;
  float4 v[28];
#pragma unroll
  for (int k = 0; k < 28; ++k) v[k] = in[(size_t)k * stride4 + i];
  float4 s = make_float4(0, 0, 0, 0);
#pragma unroll
  for (int k = 12; k < 28; ++k) { s.x += v[k].x; s.y += v[k].y; s.z += v[k].z; s.w += v[k].w; }
  // rows of the four entities (transposed in registers)
  o0[4u * i + 0u] = make_float4(v[0].x, v[1].x, v[2].x, v[3].x); o0[4u * i + 1u] = make_float4(v[0].y, v[1].y, v[2].y, v[3].y);
  o0[4u * i + 2u] = make_float4(v[0].z, v[1].z, v[2].z, v[3].z); o0[4u * i + 3u] = make_float4(v[0].w, v[1].w, v[2].w, v[3].w);
  o1[4u * i + 0u] = make_float4(v[4].x, v[5].x, v[6].x, v[7].x); o1[4u * i + 1u] = make_float4(v[4].y, v[5].y, v[6].y, v[7].y);
  o1[4u * i + 2u] = make_float4(v[4].z, v[5].z, v[6].z, v[7].z); o1[4u * i + 3u] = make_float4(v[4].w, v[5].w, v[6].w, v[7].w);
  o2[4u * i + 0u] = make_float4(v[8].x + s.x, v[9].x, v[10].x, v[11].x); o2[4u * i + 1u] = make_float4(v[8].y + s.y, v[9].y, v[10].y, v[11].y);
  o2[4u * i + 2u] = make_float4(v[8].z + s.z, v[9].z, v[10].z, v[11].z); o2[4u * i + 3u] = make_float4(v[8].w + s.w, v[9].w, v[10].w, v[11].w);
  o3[i] = s; o4[i] = make_float4(v[0].x + s.x, v[0].y + s.y, v[0].z + s.z, v[0].w + s.w);
}
// the layouts under discussion for the fused kernel, same bytes: NDW dword streams + NF4 float4 streams in (16 + 3: today --
// 16 dword fields and the parent's rows; 4 + 6: position and link as dwords, the per-entity constants in three float4 groups)
template <int NDW, int NF4>
__global__ __launch_bounds__(256) void k_mixed(const float* __restrict__ in, const float4* __restrict__ in4, float4* __restrict__ o0, float4* __restrict__ o1, float4* __restrict__ o2, float* __restrict__ o3, float* __restrict__ o4, uint32_t n, uint32_t stride)
{
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i >= n) return;
  float v[NDW]; float4 q[NF4];
#pragma unroll
  for (int k = 0; k < NDW; ++k) v[k] = in[(size_t)k * stride + i];
#pragma unroll
  for (int k = 0; k < NF4; ++k) q[k] = in4[(size_t)k * stride + i];
  float s = 0.0f;
#pragma unroll
  for (int k = 0; k < NDW; ++k) s += v[k];
  float4 t = make_float4(s, s, s, s);
#pragma unroll
  for (int k = 0; k < NF4; ++k) { t.x += q[k].x; t.y += q[k].y; t.z += q[k].z; t.w += q[k].w; }
  o0[i] = t; o1[i] = make_float4(t.y, t.x, t.w, t.z); o2[i] = make_float4(t.z, t.w, t.x, t.y);
  o3[i] = s; o4[i] = t.x + s;
}
int main()
{
  const uint32_t n = 1u << 20, stride = n + 4096u;
  float* in; CK(hipMalloc(&in, (size_t)28 * stride * 4)); CK(hipMemset(in, 0, (size_t)28 * stride * 4));
  float4 *o0, *o1, *o2; float *o3, *o4;
  CK(hipMalloc(&o0, (size_t)n * 16)); CK(hipMalloc(&o1, (size_t)n * 16)); CK(hipMalloc(&o2, (size_t)n * 16)); CK(hipMalloc(&o3, (size_t)n * 8)); CK(hipMalloc(&o4, (size_t)n * 4));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const double moved = (double)n * (112.0 + 56.0);
  auto run = [&](const char* name, auto launch) {
    std::vector<float> ms;
    for (int r = 0; r < 60; ++r) { hipEventRecord(e0, 0); launch(); hipEventRecord(e1, 0); hipEventSynchronize(e1); float t; hipEventElapsedTime(&t, e0, e1); if (r >= 10) ms.push_back(t); }
    std::sort(ms.begin(), ms.end());
    std::printf("%-40s %7.1f GB/s (%.1f us median, %.1f min)\n", name, moved / (ms[ms.size() / 2] * 1e-3) / 1e9, ms[ms.size() / 2] * 1e3, ms[0] * 1e3);
  };
  run("dword SoA, 28 streams in", [&] { hipLaunchKernelGGL(k_dword<28>, dim3(n / 256), dim3(256), 0, 0, in, o0, o1, o2, o3, o4, n, stride); });
  run("float4 streams, 7 in", [&] { hipLaunchKernelGGL(k_float4, dim3(n / 256), dim3(256), 0, 0, (const float4*)in, o0, o1, o2, (float2*)o3, n, stride); });
  run("dword SoA, lane = 4 entities (16 B/lane)", [&] { hipLaunchKernelGGL(k_dword_x4, dim3(n / 1024), dim3(256), 0, 0, (const float4*)in, o0, o1, o2, (float4*)o3, (float4*)o4, n / 4, stride / 4); });
  {
    float4* in4; if (hipMalloc(&in4, (size_t)8 * stride * 16) != hipSuccess) return 1;
    hipMemset(in4, 0, (size_t)8 * stride * 16);
    run("16 dword + 3 float4 in (today)", [&] { hipLaunchKernelGGL((k_mixed<16, 3>), dim3(n / 256), dim3(256), 0, 0, in, in4, o0, o1, o2, o3, o4, n, stride); });
    run("4 dword + 6 float4 in (constants packed)", [&] { hipLaunchKernelGGL((k_mixed<4, 6>), dim3(n / 256), dim3(256), 0, 0, in, in4, o0, o1, o2, o3, o4, n, stride); });
    run("16 dword + 3 float4 in (today) again", [&] { hipLaunchKernelGGL((k_mixed<16, 3>), dim3(n / 256), dim3(256), 0, 0, in, in4, o0, o1, o2, o3, o4, n, stride); });
    run("1 dword + 7 float4 in (all packed)", [&] { hipLaunchKernelGGL((k_mixed<1, 7>), dim3(n / 256), dim3(256), 0, 0, in, in4, o0, o1, o2, o3, o4, n, stride); });
  }
  run("dword SoA again", [&] { hipLaunchKernelGGL(k_dword<28>, dim3(n / 256), dim3(256), 0, 0, in, o0, o1, o2, o3, o4, n, stride); });
  return 0;
}
