// Issue rate of v_mfma_f64_16x16x4_f64 and of v_fma_f64 on the device at hand (DESIGN.md section 10.1 quotes the result).
//   hipcc --offload-arch=gfx950 -O3 -o mfma_f64_rate tools/micro/mfma_f64_rate.hip && ./mfma_f64_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));

template <int CHAINS>
__global__ __launch_bounds__(256) void k_mfma(double* out, int iters, double a0, double b0) {
  d4 acc[CHAINS];
  for (int c = 0; c < CHAINS; ++c) acc[c] = d4{0, 0, 0, 0};
  double a = a0 + threadIdx.x, b = b0;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) acc[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[c], 0, 0, 0);
  }
  double s = 0;
  for (int c = 0; c < CHAINS; ++c) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
  out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int CHAINS>
__global__ __launch_bounds__(256) void k_fma(double* out, int iters, double a0, double b0) {
  double acc[CHAINS];
  for (int c = 0; c < CHAINS; ++c) acc[c] = c;
  double a = a0 + threadIdx.x, b = b0;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) acc[c] = __builtin_fma(a, acc[c], b);
  }
  double s = 0;
  for (int c = 0; c < CHAINS; ++c) s += acc[c];
  out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main() {
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
  const int cus = p.multiProcessorCount, blocks = cus * 4, iters = 20000;
  double* out; hipMalloc(&out, sizeof(double) * blocks * 256);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto time = [&](auto launch) { launch(); hipDeviceSynchronize(); hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); return (double)ms * 1e-3; };
  const double clk = p.clockRate * 1e3;
  {
    const double t = time([&] { hipLaunchKernelGGL(k_mfma<4>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0, 1e-9); });
    const double n = (double)blocks * 4 * iters * 4;      // wave-level MFMA instructions
    printf("mfma_f64_16x16x4: %.1f TFLOP/s, %.1f cycles per instruction and SIMD (clock %.2f GHz, %d CUs)\n", n * 2048 / t * 1e-12,
           t * clk / (n / (cus * 4.0)), clk * 1e-9, cus);
  }
  {
    const double t = time([&] { hipLaunchKernelGGL(k_fma<8>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0000001, 1e-9); });
    const double n = (double)blocks * 4 * iters * 8;
    printf("v_fma_f64:        %.1f TFLOP/s, %.2f cycles per wave instruction and SIMD\n", n * 128 / t * 1e-12, t * clk / (n / (cus * 4.0)));
  }
  return 0;
}
