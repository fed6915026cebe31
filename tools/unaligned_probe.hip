// Probe (GPU box): do 16-byte global loads / stores work -- and at what rate -- when the address is only 4-byte aligned?
// (k-contiguous GEMM operands with row pitches 425 / 483 / 187 floats.)   hipcc --offload-arch=gfx950 -O3 tools/unaligned_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
typedef float f4 __attribute__((ext_vector_type(4)));
// rows x pitch floats; every lane loads 16 bytes at row*pitch + 4*j (j < cols/4), sums, writes one float per row block
template <typename V>
__global__ void rd(const float* __restrict__ a, int rows, int pitch, int cols4, float* __restrict__ out) {
  float acc = 0.f;
  for (long r = blockIdx.x; r < rows; r += gridDim.x)
    for (int j = threadIdx.x; j < cols4; j += blockDim.x) {
      const V v = *reinterpret_cast<const V*>(a + r * pitch + 4 * j);
      acc += v[0] + v[1] + v[2] + v[3];
    }
  atomicAdd(out, acc);
}
template <typename V>
__global__ void wr(float* __restrict__ a, int rows, int pitch, int cols4) {
  for (long r = blockIdx.x; r < rows; r += gridDim.x)
    for (int j = threadIdx.x; j < cols4; j += blockDim.x) {
      V v; v[0] = (float)r; v[1] = (float)j; v[2] = 1.f; v[3] = 2.f;
      *reinterpret_cast<V*>(a + r * pitch + 4 * j) = v;
    }
}
int main() {
  const int rows = 32768, cols = 424;
  float *a, *out;
  hipMalloc(&a, (size_t)rows * 488 * 4 + 64); hipMalloc(&out, 4);
  std::vector<float> h((size_t)rows * 488 + 16);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u) % 1000) * 1e-3f;
  hipMemcpy(a, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int pitch : {428, 425, 427, 483}) {
    double ref = 0; for (int r = 0; r < rows; ++r) for (int c = 0; c < cols; ++c) ref += h[(size_t)r * pitch + c];
    for (int rep = 0; rep < 2; ++rep) {
      hipMemset(out, 0, 4);
      hipEventRecord(e0);
      if (pitch % 4 == 0 && rep == 0) rd<f4><<<2048, 128>>>(a, rows, pitch, cols / 4, out); else rd<f4u><<<2048, 128>>>(a, rows, pitch, cols / 4, out);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms, got; hipEventElapsedTime(&ms, e0, e1); hipMemcpy(&got, out, 4, hipMemcpyDeviceToHost);
      hipError_t err = hipGetLastError();
      printf("read pitch %d %s: %.1f us  %.0f GB/s  sum %.1f (ref %.1f) %s\n", pitch, (pitch % 4 == 0 && rep == 0) ? "f4 " : "f4u", ms * 1e3,
             (double)rows * cols * 4 / ms / 1e6, got, ref, hipGetErrorString(err));
    }
  }
  // unaligned 16-byte stores: pitch 187
  {
    const int p = 187, c4 = 46;
    hipMemset(a, 0, (size_t)rows * 488 * 4);
    hipEventRecord(e0); wr<f4u><<<2048, 64>>>(a, rows, p, c4); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(h.data(), a, (size_t)rows * p * 4, hipMemcpyDeviceToHost);
    long bad = 0;
    for (int r = 0; r < rows; ++r) for (int j = 0; j < c4; ++j) { const float* q = &h[(size_t)r * p + 4 * j]; if (q[0] != (float)r || q[1] != (float)j || q[2] != 1.f || q[3] != 2.f) ++bad; }
    printf("write pitch 187 f4u: %.1f us, %ld bad of %ld  %s\n", ms * 1e3, bad, (long)rows * c4, hipGetErrorString(hipGetLastError()));
  }
  return 0;
}
