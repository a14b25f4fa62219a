// Two graphs on two streams ordered by EXTERNAL event nodes (record at the end of graph A, wait at the start of graph B,
// record at the end of B queried from the host): do the calls succeed inside a capture, and does the ordering hold?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k_write_after_spin(uint32_t* flag, uint32_t v, float* p, int n) { float x = p[threadIdx.x]; for (int i = 0; i < n; ++i) x = x * 1.0001f + 0.5f; p[threadIdx.x] = x; __syncthreads(); if (threadIdx.x == 0) atomicAdd(flag, v); }
__global__ void k_copy_flag(const uint32_t* flag, uint32_t* out, uint32_t* cursor) { out[atomicAdd(cursor, 1u)] = *flag; }
#define CK(x) do { hipError_t err_ = (x); std::printf("%-70s %s\n", #x, hipGetErrorString(err_)); if (err_ != hipSuccess) failed = 1; } while (0)
int main()
{
  int failed = 0;
  hipStream_t a, b; CK(hipStreamCreateWithFlags(&a, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&b, hipStreamNonBlocking));
  float* d; CK(hipMalloc(&d, 4096)); CK(hipMemset(d, 0, 4096));
  uint32_t* flag; CK(hipMalloc(&flag, 64)); CK(hipMemset(flag, 0, 64));
  uint32_t* out; CK(hipMalloc(&out, 4096 * 4)); CK(hipMemset(out, 0, 4096 * 4));
  for (int variant = 0; variant < 2; ++variant) {
    const unsigned evFlags = variant == 0 ? (hipEventDisableTiming | hipEventReleaseToDevice) : hipEventDisableTiming;
    std::printf("---- events created with flags 0x%x\n", evFlags);
    hipEvent_t packed, done; CK(hipEventCreateWithFlags(&packed, evFlags)); CK(hipEventCreateWithFlags(&done, evFlags));
    hipGraph_t ga, gb; hipGraphExec_t ea, eb;
    CK(hipStreamBeginCapture(a, hipStreamCaptureModeThreadLocal));
    hipLaunchKernelGGL(k_write_after_spin, dim3(1), dim3(64), 0, a, flag, 1u, d, 20000);
    CK(hipEventRecordWithFlags(packed, a, hipEventRecordExternal));
    CK(hipStreamEndCapture(a, &ga));
    CK(hipGraphInstantiate(&ea, ga, nullptr, nullptr, 0));
    CK(hipStreamBeginCapture(b, hipStreamCaptureModeThreadLocal));
    CK(hipStreamWaitEvent(b, packed, hipEventWaitExternal));
    hipLaunchKernelGGL(k_copy_flag, dim3(1), dim3(1), 0, b, flag, out, flag + 4);
    CK(hipEventRecordWithFlags(done, b, hipEventRecordExternal));
    CK(hipStreamEndCapture(b, &gb));
    CK(hipGraphInstantiate(&eb, gb, nullptr, nullptr, 0));
    CK(hipMemset(flag, 0, 64));
    hipError_t e1 = hipSuccess, e2 = hipSuccess;
    for (int i = 0; i < 200 && e1 == hipSuccess && e2 == hipSuccess; ++i) { e1 = hipGraphLaunch(ea, a); e2 = hipGraphLaunch(eb, b); if ((i & 7) == 7) { hipStreamSynchronize(a); hipStreamSynchronize(b); } }
    std::printf("launches: %s / %s; query(done) = %s\n", hipGetErrorString(e1), hipGetErrorString(e2), hipGetErrorString(hipEventQuery(done)));
    CK(hipDeviceSynchronize());
    std::printf("query(done) after sync = %s\n", hipGetErrorString(hipEventQuery(done)));
    std::vector<uint32_t> h(200); CK(hipMemcpy(h.data(), out, 800, hipMemcpyDeviceToHost));
    int bad = 0; for (int i = 0; i < 200; ++i) if (h[i] < (uint32_t)i + 1u) ++bad;
    std::printf("copies that ran before their writer: %d of 200\n", bad);
    CK(hipMemset(out, 0, 4096 * 4));
    hipGraphExecDestroy(ea); hipGraphExecDestroy(eb); hipGraphDestroy(ga); hipGraphDestroy(gb);
    // the same events in a second capture (a stale graph re-captured), and used eagerly in between
    CK(hipEventRecord(packed, a)); CK(hipStreamWaitEvent(b, packed, 0));
    CK(hipStreamBeginCapture(a, hipStreamCaptureModeThreadLocal));
    hipLaunchKernelGGL(k_write_after_spin, dim3(1), dim3(64), 0, a, flag, 1u, d, 200);
    CK(hipEventRecordWithFlags(packed, a, hipEventRecordExternal));
    CK(hipStreamEndCapture(a, &ga));
    hipGraphDestroy(ga);
    // capture on b in RELAXED mode while stream a is not capturing
    CK(hipStreamBeginCapture(b, hipStreamCaptureModeRelaxed));
    CK(hipStreamWaitEvent(b, packed, hipEventWaitExternal));
    hipLaunchKernelGGL(k_copy_flag, dim3(1), dim3(1), 0, b, flag, out, flag + 4);
    CK(hipEventRecordWithFlags(done, b, hipEventRecordExternal));
    CK(hipStreamEndCapture(b, &gb));
    hipGraphDestroy(gb);
    CK(hipDeviceSynchronize());
  }
  std::printf(failed ? "FAILED\n" : "ok\n");
  return failed;
}
