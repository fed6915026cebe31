// Probe (GPU box only): where does global_load_lds_dwordx4 put each lane's 16 bytes?  One wave loads lane l's 16 bytes
// from g + 4*perm(l) floats into an LDS region given by a wave-uniform base; the LDS image is dumped.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
__global__ void probe(const float* __restrict__ g, float* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 2048; i += blockDim.x) sm[i] = -1.f;
  __syncthreads();
  const int src_chunk = (lane * 7 + 3) & 63;            // a permutation of 0..63
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g + wave * 256 + src_chunk * 4),
                                   (__attribute__((address_space(3))) void*)(sm + wave * 256), 16, 0, 0);
  __builtin_amdgcn_s_waitcnt(0);
  __syncthreads();
  for (int i = threadIdx.x; i < 2048; i += blockDim.x) out[i] = sm[i];
}
int main() {
  std::vector<float> h(4096);
  for (int i = 0; i < 4096; ++i) h[i] = (float)i;
  float *g, *o;
  hipMalloc(&g, 4096 * 4); hipMalloc(&o, 2048 * 4);
  hipMemcpy(g, h.data(), 4096 * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(probe, dim3(1), dim3(256), 8192, 0, g, o);
  std::vector<float> r(2048);
  hipMemcpy(r.data(), o, 2048 * 4, hipMemcpyDeviceToHost);
  int ok = 1;
  for (int w = 0; w < 4; ++w)
    for (int l = 0; l < 64; ++l) {
      const int src = (l * 7 + 3) & 63;
      for (int c = 0; c < 4; ++c) if (r[w * 256 + l * 4 + c] != (float)(w * 256 + src * 4 + c)) ok = 0;
    }
  printf("lane l's 16 bytes land at base + 16*l: %s\n", ok ? "YES" : "NO");
  if (!ok) { for (int i = 0; i < 32; ++i) printf("%g ", r[i]); printf("\n"); }
  for (int i = 1024; i < 1032; ++i) printf("%g ", r[i]); printf(" (untouched region should be -1)\n");
  return 0;
}
