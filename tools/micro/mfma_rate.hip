// Issue rate of the two float32-input MFMAs on gfx950: cycles per instruction with N independent accumulators, one wave
// per SIMD (hipcc --offload-arch=gfx950 -O3 mfma_rate.hip -o mfma_rate && ./mfma_rate)
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ void k16(float* out, unsigned long long* cyc, int iters) {
  f32x4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  float a = threadIdx.x * 0.001f, b = 1.f + threadIdx.x * 0.002f;
  unsigned long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
  }
  unsigned long long t1 = clock64();
  float s = 0.f;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int NACC>
__global__ void k32(float* out, unsigned long long* cyc, int iters) {
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; ++i)
    for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
  float a = threadIdx.x * 0.001f, b = 1.f + threadIdx.x * 0.002f;
  unsigned long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
  }
  unsigned long long t1 = clock64();
  float s = 0.f;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][15];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
int main() {
  float* out; unsigned long long* cyc;
  hipMalloc(&out, 1024 * 256 * 4); hipMalloc(&cyc, 1024 * 8);
  const int iters = 2000;
  unsigned long long h[4];
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
#define RUN(kern, nacc, threads, label)                                                        \
  hipLaunchKernelGGL(kern, dim3(256), dim3(threads), 0, 0, out, cyc, iters);                   \
  hipDeviceSynchronize();                                                                       \
  hipEventRecord(e0, 0);                                                                        \
  hipLaunchKernelGGL(kern, dim3(256), dim3(threads), 0, 0, out, cyc, iters);                   \
  hipEventRecord(e1, 0);                                                                        \
  hipDeviceSynchronize();                                                                       \
  { float ms = 0.f; hipEventElapsedTime(&ms, e0, e1);                                          \
    (void)hipMemcpy(h, cyc, 8, hipMemcpyDeviceToHost);                                         \
    printf("%-44s %6.1f ticks / MFMA   kernel %.3f ms = %.0f ns / MFMA / wave  (tick = %.3f ns)\n", label, \
           (double)h[0] / ((double)iters * nacc), ms, ms * 1e6 / ((double)iters * nacc), ms * 1e6 / (double)h[0]); }
  RUN(k16<8>, 8, 256, "16x16x4 f32, 8 acc, 1 wave/SIMD");
  RUN(k16<8>, 8, 256, "16x16x4 f32, 8 acc, 1 wave/SIMD (again)");
  RUN(k16<1>, 1, 256, "16x16x4 f32, 1 acc (dependent chain)");
  RUN(k16<2>, 2, 256, "16x16x4 f32, 2 acc");
  RUN(k16<8>, 8, 512, "16x16x4 f32, 8 acc, 2 waves/SIMD (per wave)");
  RUN(k32<4>, 4, 256, "32x32x2 f32, 4 acc, 1 wave/SIMD");
  RUN(k32<1>, 1, 256, "32x32x2 f32, 1 acc (dependent chain)");
  RUN(k32<4>, 4, 512, "32x32x2 f32, 4 acc, 2 waves/SIMD (per wave)");
  return 0;
}
