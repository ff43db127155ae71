// micro-benchmark: sustained v_mfma_f32_32x32x2_f32 rate vs accumulators per wave and waves per SIMD
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NACC>
__global__ void __launch_bounds__(256) k(float* out, int iters, float a, float b) {
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  float x = a + threadIdx.x, y = b;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 16; ++u)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, acc[i], 0, 0, 0);
  }
  float s = 0;
  for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NACC> void run(float* d, int blocks) {
  const int iters = 2000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<NACC>, dim3(blocks), dim3(256), 0, 0, d, 10, 1.f, 2.f);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<NACC>, dim3(blocks), dim3(256), 0, 0, d, iters, 1.f, 2.f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double fl = (double)blocks * 4 * iters * 16 * NACC * 4096.0;
  printf("acc/wave %d blocks %5d (%.1f waves/SIMD): %7.3f ms  %6.1f TF/s\n", NACC, blocks, blocks / 256.0, ms, fl / ms / 1e9);
}
int main() {
  float* d; hipMalloc(&d, 4096 * 256 * 4);
  for (int blocks : {256, 512, 768, 1024}) { run<1>(d, blocks); run<2>(d, blocks); run<4>(d, blocks); }
  return 0;
}
