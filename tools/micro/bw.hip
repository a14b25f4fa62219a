// HBM ceilings of this box by access mix: read-only, write-only, copy (1:1), two reads per write (the fused kernel's mix).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t err_ = (x); if (err_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(err_)); return 1; } } while (0)
__global__ __launch_bounds__(256) void k_read(const float4* __restrict__ a, float4* __restrict__ out, size_t n)
{
  float4 s = make_float4(0, 0, 0, 0);
  for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += (size_t)gridDim.x * 256ull) { const float4 v = a[i]; s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w; }
  if (s.x == 12345.678f) out[0] = s;
}
__global__ __launch_bounds__(256) void k_write(float4* __restrict__ a, size_t n)
{
  for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += (size_t)gridDim.x * 256ull) a[i] = make_float4(1, 2, 3, 4);
}
__global__ __launch_bounds__(256) void k_copy(const float4* __restrict__ a, float4* __restrict__ b, size_t n)
{
  for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += (size_t)gridDim.x * 256ull) b[i] = a[i];
}
__global__ __launch_bounds__(256) void k_two_reads_one_write(const float4* __restrict__ a, const float4* __restrict__ c, float4* __restrict__ b, size_t n)
{
  for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += (size_t)gridDim.x * 256ull) { const float4 x = a[i], y = c[i]; b[i] = make_float4(x.x + y.x, x.y + y.y, x.z + y.z, x.w + y.w); }
}
int main()
{
  const size_t bytes = 1ull << 30, n = bytes / 16;
  float4 *a, *b, *c; CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes)); CK(hipMalloc(&c, bytes));
  CK(hipMemset(a, 1, bytes)); CK(hipMemset(b, 0, bytes)); CK(hipMemset(c, 2, bytes));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (unsigned grid : { 2048u, 8192u, 32768u, 0u }) {
    const unsigned g = grid ? grid : (unsigned)(n / 256);
    auto run = [&](const char* name, double moved, auto launch) {
      std::vector<float> ms;
      for (int r = 0; r < 12; ++r) { hipEventRecord(e0, 0); launch(); hipEventRecord(e1, 0); hipEventSynchronize(e1); float t; hipEventElapsedTime(&t, e0, e1); if (r >= 2) ms.push_back(t); }
      std::sort(ms.begin(), ms.end());
      std::printf("grid %7u  %-24s %7.1f GB/s (median of 10, %.1f us)\n", g, name, moved / (ms[ms.size() / 2] * 1e-3) / 1e9, ms[ms.size() / 2] * 1e3);
    };
    run("read", (double)bytes, [&] { hipLaunchKernelGGL(k_read, dim3(g), dim3(256), 0, 0, a, b, n); });
    run("write", (double)bytes, [&] { hipLaunchKernelGGL(k_write, dim3(g), dim3(256), 0, 0, b, n); });
    run("copy (1 read : 1 write)", 2.0 * bytes, [&] { hipLaunchKernelGGL(k_copy, dim3(g), dim3(256), 0, 0, a, b, n); });
    run("2 reads : 1 write", 3.0 * bytes, [&] { hipLaunchKernelGGL(k_two_reads_one_write, dim3(g), dim3(256), 0, 0, a, c, b, n); });
  }
  // the fused kernel's size class: 176 MB per launch (does the ramp-up / tail of a 35 us kernel cost bandwidth?)
  const size_t small = 58ull << 20, ns = small / 16;
  std::vector<float> ms;
  for (int r = 0; r < 40; ++r) { hipEventRecord(e0, 0); hipLaunchKernelGGL(k_two_reads_one_write, dim3(4096), dim3(256), 0, 0, a, c, b, ns); hipEventRecord(e1, 0); hipEventSynchronize(e1); float t; hipEventElapsedTime(&t, e0, e1); if (r >= 5) ms.push_back(t); }
  std::sort(ms.begin(), ms.end());
  std::printf("2 reads : 1 write, 174 MB per launch: %7.1f GB/s (%.1f us)\n", 3.0 * small / (ms[ms.size() / 2] * 1e-3) / 1e9, ms[ms.size() / 2] * 1e3);
  return 0;
}
