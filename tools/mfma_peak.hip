// Sustained f32-MFMA ceiling of the box (GPU box only): every wave issues independent v_mfma_f32_32x32x2_f32 back to back
// (4 accumulators, no memory traffic); prints TFLOP/s and the shader clock seen by clock64() against the 100 MHz wall clock.
// usage: mfma_peak [waves_per_simd=2] [iters=4096]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(256, 2) void peak_kernel(float* out, int iters, unsigned long long* clk) {
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  float a = threadIdx.x * 1e-3f, b = blockIdx.x * 1e-3f;
  const unsigned long long c0 = clock64(), w0 = wall_clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
  }
  const unsigned long long c1 = clock64(), w1 = wall_clock64();
  float s = 0.f;
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x < 8) { clk[2 * blockIdx.x] = c1 - c0; clk[2 * blockIdx.x + 1] = w1 - w0; }
}
int main(int argc, char** argv) {
  const int wps = argc > 1 ? atoi(argv[1]) : 2, iters = argc > 2 ? atoi(argv[2]) : 4096;
  const int grid = 256 * wps;
  float* out; unsigned long long* clk;
  hipMalloc(&out, (size_t)grid * 256 * 4); hipMalloc(&clk, 16 * 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 4; ++rep) {
    hipEventRecord(e0);
    for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(peak_kernel, dim3(grid), dim3(256), 0, 0, out, iters, clk);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[16]; hipMemcpy(h, clk, sizeof(h), hipMemcpyDeviceToHost);
    const double flop = (double)grid * 4 * iters * 16.0 * 4096.0 * 10;
    printf("waves/SIMD %d iters %d: %.1f us/launch  %.1f TFLOP/s   block0: %llu shader cycles in %llu wall ticks -> %.0f MHz; cycles per MFMA per SIMD %.1f\n",
           wps, iters, ms * 1e3 / 10, flop / (ms * 1e-3) / 1e12, h[0], h[1], h[0] / (h[1] / 100.0), (double)h[0] / (iters * 16.0 * wps));
  }
  return 0;
}
