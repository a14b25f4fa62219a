// Host cost of the runtime calls a pipelined tile step is made of, on two streams with real kernels in flight.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <chrono>
#include <cstdio>
#include <vector>
__global__ void k_spin(float* p, int n) { float v = p[threadIdx.x]; for (int i = 0; i < n; ++i) v = v * 1.0001f + 0.5f; p[threadIdx.x] = v; }
__global__ void k_write_after_spin(uint32_t* flag, uint32_t v, float* p, int n) { float x = p[threadIdx.x]; for (int i = 0; i < n; ++i) x = x * 1.0001f + 0.5f; p[threadIdx.x] = x; __syncthreads(); if (threadIdx.x == 0) *flag = v; }
__global__ void k_copy_flag(const uint32_t* flag, uint32_t* out) { *out = *flag; }
static double now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
int main()
{
  hipStream_t a, b; CK(hipStreamCreateWithFlags(&a, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&b, hipStreamNonBlocking));
  float* d; CK(hipMalloc(&d, 4096)); CK(hipMemset(d, 0, 4096));
  uint32_t* flag; CK(hipMalloc(&flag, 64)); CK(hipMemset(flag, 0, 64));
  hipEvent_t ev[4];
  for (auto& evt : ev) CK(hipEventCreateWithFlags(&evt, hipEventDisableTiming | hipEventReleaseToDevice));
  const int N = 2000;
  auto bench = [&](const char* name, auto fn) {
    for (int i = 0; i < 50; ++i) fn(i);
    hipDeviceSynchronize();
    const double t0 = now();
    for (int i = 0; i < N; ++i) { fn(i); if ((i & 63) == 63) hipDeviceSynchronize(); }
    const double t1 = now();
    hipDeviceSynchronize();
    std::printf("%-44s %.2f us per call group\n", name, (t1 - t0) / N);
    return 0;
  };
  bench("kernel launch (stream a)", [&](int) { hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, a, d, 10); });
  bench("launch a + launch b", [&](int) { hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, a, d, 10); hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, b, d + 64, 10); });
  bench("launch a + eventRecord(a)", [&](int i) { hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, a, d, 10); hipEventRecord(ev[i & 3], a); });
  bench("launch a + record(a) + waitEvent(b) + launch b", [&](int i) { hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, a, d, 10); hipEventRecord(ev[i & 3], a); hipStreamWaitEvent(b, ev[i & 3], 0); hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, b, d + 64, 10); });
  bench("launch a + eventQuery", [&](int i) { hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, a, d, 10); (void)hipEventQuery(ev[i & 3]); (void)hipGetLastError(); });
  uint32_t seq = 0;
  bench("launch a + writeValue32(a) + waitValue32(b) + launch b", [&](int) { ++seq; hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, a, d, 10); hipStreamWriteValue32(a, flag, seq, 0); hipStreamWaitValue32(b, flag, seq, hipStreamWaitValueGte, 0xFFFFFFFFu); hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, b, d + 64, 10); });
  // the event attached to the dispatch itself (hipExtLaunchKernelGGL's stop event) instead of a separate record
  bench("extLaunch a (stop event) + waitEvent(b) + launch b", [&](int i) { hipExtLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, a, nullptr, ev[i & 3], 0, d, 10); hipStreamWaitEvent(b, ev[i & 3], 0); hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, b, d + 64, 10); });
  // does the wait really order?  a: long spin, then writes i; b: copies what it sees
  {
    uint32_t* out; CK(hipMalloc(&out, 4096 * 4)); CK(hipMemset(out, 0xFF, 4096 * 4));
    hipEvent_t e2[8]; for (auto& evt : e2) CK(hipEventCreateWithFlags(&evt, hipEventDisableTiming | hipEventReleaseToDevice));
    int bad = 0;
    for (int i = 0; i < 1000; ++i) {
      hipExtLaunchKernelGGL(k_write_after_spin, dim3(1), dim3(64), 0, a, nullptr, e2[i & 7], 0, flag + 8, (uint32_t)i + 1u, d, 20000);
      hipStreamWaitEvent(b, e2[i & 7], 0);
      hipLaunchKernelGGL(k_copy_flag, dim3(1), dim3(1), 0, b, flag + 8, out + i);
      if ((i & 7) == 7) hipDeviceSynchronize();
    }
    hipDeviceSynchronize();
    std::vector<uint32_t> h(1000); CK(hipMemcpy(h.data(), out, 4000, hipMemcpyDeviceToHost));
    for (int i = 0; i < 1000; ++i) if (h[i] != (uint32_t)i + 1u) ++bad;
    std::printf("ordering through a stop event: %d of 1000 out of order\n", bad);
  }
  std::printf("done\n");
  return 0;
}
