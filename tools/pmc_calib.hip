// pmc_calib.hip -- known-byte-count streaming kernels to calibrate rocprofv3's FETCH_SIZE / WRITE_SIZE
// on gfx950 for the access widths the tick kernels use (MI355X_MICROARCH.md, HBM section: FETCH_SIZE
// reads 1/2 for 16 B/lane streams; other widths must be calibrated on a known byte count).
// Each kernel moves exactly BYTES bytes in and BYTES bytes out of buffers larger than the 256 MiB
// Infinity Cache.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

__global__ void calib_copy_dword(const float* __restrict__ a, float* __restrict__ b, size_t n)
{
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) b[i] = a[i];
}
__global__ void calib_copy_float4(const float4* __restrict__ a, float4* __restrict__ b, size_t n)
{
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) b[i] = a[i];
}
// the fused kernel's shape: 13 dword streams in, 3 float4 streams out (per "entity")
__global__ void calib_soa13_to_rows(const float* __restrict__ in, float4* __restrict__ o0, float4* __restrict__ o1, float4* __restrict__ o2, size_t n)
{
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    float v[13];
#pragma unroll
    for (int k = 0; k < 13; ++k) v[k] = in[(size_t)k * n + i];
    o0[i] = make_float4(v[0], v[1], v[2], v[3]);
    o1[i] = make_float4(v[4], v[5], v[6], v[7]);
    o2[i] = make_float4(v[8] + v[12], v[9], v[10], v[11]);
  }
}

int main()
{
  const size_t bytes = 1024ull << 20;                       // 1 GiB each way
  float *a, *b;
  if (hipMalloc(&a, bytes) != hipSuccess || hipMalloc(&b, bytes) != hipSuccess) { std::printf("alloc failed\n"); return 1; }
  hipMemset(a, 1, bytes); hipMemset(b, 0, bytes);
  for (int rep = 0; rep < 3; ++rep) {
    hipLaunchKernelGGL(calib_copy_dword, dim3(2048), dim3(256), 0, 0, a, b, bytes / 4);
    hipLaunchKernelGGL(calib_copy_float4, dim3(2048), dim3(256), 0, 0, (const float4*)a, (float4*)b, bytes / 16);
    // 16M "entities": 13*4 = 52 B in, 48 B out each -> 832 MiB in, 768 MiB out
    const size_t n = 16u << 20;
    hipLaunchKernelGGL(calib_soa13_to_rows, dim3(2048), dim3(256), 0, 0, a, (float4*)b, (float4*)b + n, (float4*)b + 2 * n, n);
  }
  hipDeviceSynchronize();
  std::printf("calib: copy kernels moved %zu bytes each way; soa13 kernel %zu in / %zu out\n", bytes, (size_t)(16u << 20) * 52, (size_t)(16u << 20) * 48);
  hipFree(a); hipFree(b);
  return 0;
}
