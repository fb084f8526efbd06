// Host latency of one "launch a tiny kernel, copy 8 bytes to pinned memory, wait" round trip on a stream, three ways of waiting:
// hipStreamSynchronize, hipEventSynchronize, and spinning on hipEventQuery.  Build: hipcc --offload-arch=gfx950 -O2 -o sync_latency sync_latency.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void k_tiny(unsigned long long* out, unsigned long long v) { if (threadIdx.x == 0 && blockIdx.x == 0) *out = v; }
__global__ void k_busy(float* p, int iters) { float v = p[threadIdx.x]; for (int i = 0; i < iters; ++i) v = v * 1.0001f + 0.5f; p[threadIdx.x] = v; }
int main() {
  hipStream_t s; CK(hipStreamCreate(&s));
  unsigned long long* d; CK(hipMalloc(&d, 8));
  unsigned long long* h; CK(hipHostMalloc(&h, 8));
  float* busy; CK(hipMalloc(&busy, 4096));
  hipEvent_t ev; CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
  const int N = 2000;
  for (int kernel_us = 0; kernel_us <= 200; kernel_us += 100) {
    const int iters = kernel_us * 250;     // ~ that many microseconds of a one-block kernel in front of the round trip
    for (int mode = 0; mode < 3; ++mode) {
      std::vector<double> t;
      for (int i = 0; i < N; ++i) {
        auto t0 = std::chrono::steady_clock::now();
        if (iters) hipLaunchKernelGGL(k_busy, dim3(1), dim3(64), 0, s, busy, iters);
        hipLaunchKernelGGL(k_tiny, dim3(1), dim3(64), 0, s, d, (unsigned long long)i);
        CK(hipMemcpyAsync(h, d, 8, hipMemcpyDeviceToHost, s));
        if (mode == 0) CK(hipStreamSynchronize(s));
        else { CK(hipEventRecord(ev, s)); if (mode == 1) CK(hipEventSynchronize(ev)); else while (hipEventQuery(ev) == hipErrorNotReady) __builtin_ia32_pause(); }
        if (*h != (unsigned long long)i) { fprintf(stderr, "stale value\n"); return 1; }
        t.push_back(std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count());
      }
      std::sort(t.begin(), t.end());
      printf("kernel ~%3d us, %-22s median %.1f us, p10 %.1f, p90 %.1f\n", kernel_us, mode == 0 ? "hipStreamSynchronize" : mode == 1 ? "hipEventSynchronize" : "spin on hipEventQuery", t[N / 2], t[N / 10], t[9 * N / 10]);
    }
  }
  return 0;
}
