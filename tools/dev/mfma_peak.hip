// Achievable fp32 MFMA rate on this box: N independent accumulator chains per wave, W waves per SIMD, no memory traffic.
// hipcc --offload-arch=gfx950 -O3 tools/dev/mfma_peak.hip -o /tmp/mfma_peak && /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int CH>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
  f32x16 acc[CH];
  for (int c = 0; c < CH; ++c) for (int q = 0; q < 16; ++q) acc[c][q] = 0.f;
  float a = threadIdx.x * 1e-3f, b = blockIdx.x * 1e-3f;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int c = 0; c < CH; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[c], 0, 0, 0);
  }
  float s = 0;
  for (int c = 0; c < CH; ++c) for (int q = 0; q < 16; ++q) s += acc[c][q];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int CH> void run(int blocks_per_cu, int iters) {
  float* out; hipMalloc(&out, 256 * 8 * 256 * 4);
  int nb = 256 * blocks_per_cu;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<CH>, dim3(nb), dim3(256), 0, 0, out, iters);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<CH>, dim3(nb), dim3(256), 0, 0, out, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double fl = (double)nb * 4 * iters * CH * 4096.0;
  printf("chains %d  blocks/CU %d  iters %d: %.3f ms  %.1f TF\n", CH, blocks_per_cu, iters, ms, fl / ms / 1e9);
  hipFree(out);
}
int main() {
  run<1>(1, 20000); run<2>(1, 10000); run<1>(2, 20000); run<2>(3, 10000); run<2>(1, 200); run<2>(3, 100); run<2>(3, 400);
  return 0;
}
