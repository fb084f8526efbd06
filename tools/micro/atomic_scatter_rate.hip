// Cost of scattering per-edge sums with device-scope atomics on the device at hand: n "observations", each adds one 64-bit value
// to the slots of its 5 grid neighbours in two arrays (the in-edge weight sums a regrouped pass 2 of path (B) would need).
//   hipcc --offload-arch=gfx950 -O3 -o atomic_scatter_rate tools/micro/atomic_scatter_rate.hip && ./atomic_scatter_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

__device__ inline size_t nb(size_t i, int k, size_t n, int W) {
  const long long off[5] = {1, -1, W, -W, W + 1};
  long long j = (long long)i + off[k];
  if (j < 0) j += n;
  if (j >= (long long)n) j -= n;
  return (size_t)j;
}

template <int MODE>
__global__ __launch_bounds__(256) void k_scatter(unsigned long long* a, unsigned long long* b, size_t n, int W) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const unsigned long long w = (i * 2654435761ull) >> 40, r = (i * 40503ull) >> 38;
#pragma unroll
    for (int k = 0; k < 5; ++k) {
      const size_t j = nb(i, k, n, W);
      if (MODE == 0) {
        __hip_atomic_fetch_add(a + j, w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_add(b + j, r + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      } else if (MODE == 1) {
        unsafeAtomicAdd(reinterpret_cast<double*>(a) + j, (double)w);
        unsafeAtomicAdd(reinterpret_cast<double*>(b) + j, (double)(r + k));
      } else if (MODE == 2) {
        __hip_atomic_fetch_add(reinterpret_cast<unsigned*>(a) + j, (unsigned)w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_add(reinterpret_cast<unsigned*>(b) + j, (unsigned)(r + k), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      } else if (MODE == 3) {
        __hip_atomic_fetch_add(a + j, w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __hip_atomic_fetch_add(b + j, r + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      } else {
        a[j] = w; b[j] = r + k;
      }
    }
  }
}

// the gather formulation of the same sums: every slot reads the values of its 5 in-neighbours (known here because the graph is a grid)
__global__ __launch_bounds__(256) void k_gather(const double* w, const double* r, double* a, double* b, size_t n, int W) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    double sa = 0, sb = 0;
#pragma unroll
    for (int k = 0; k < 5; ++k) { const size_t j = nb(i, k, n, W); sa += w[j]; sb += r[5 * j + k]; }
    a[i] = sa; b[i] = sb;
  }
}

int main() {
  const size_t n = 10000000; const int W = 3651;
  unsigned long long *a, *b; double* r;
  hipMalloc(&a, 8 * n); hipMalloc(&b, 8 * n); hipMalloc(&r, 8 * 5 * n);
  hipMemset(a, 0, 8 * n); hipMemset(b, 0, 8 * n); hipMemset(r, 0, 40 * n);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto time = [&](auto launch) {
    launch(); hipDeviceSynchronize();
    hipEventRecord(e0); for (int q = 0; q < 5; ++q) launch(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms / 5; };
  const int blocks = 256 * 8;
  printf("u64 atomics, agent scope:     %.3f ms\n", time([&] { hipLaunchKernelGGL(k_scatter<0>, dim3(blocks), dim3(256), 0, 0, a, b, n, W); }));
  printf("f64 atomics (unsafeAtomicAdd): %.3f ms\n", time([&] { hipLaunchKernelGGL(k_scatter<1>, dim3(blocks), dim3(256), 0, 0, a, b, n, W); }));
  printf("u32 atomics, agent scope:     %.3f ms\n", time([&] { hipLaunchKernelGGL(k_scatter<2>, dim3(blocks), dim3(256), 0, 0, a, b, n, W); }));
  printf("u64 atomics, workgroup scope: %.3f ms\n", time([&] { hipLaunchKernelGGL(k_scatter<3>, dim3(blocks), dim3(256), 0, 0, a, b, n, W); }));
  printf("plain stores:                 %.3f ms\n", time([&] { hipLaunchKernelGGL(k_scatter<4>, dim3(blocks), dim3(256), 0, 0, a, b, n, W); }));
  printf("gather of 5 + 5 f64:          %.3f ms\n", time([&] { hipLaunchKernelGGL(k_gather, dim3(blocks), dim3(256), 0, 0, (double*)a, r, (double*)b, r + n, n, W); }));
  return 0;
}
