// micro-benchmark: LDS atomic throughput by type on gfx950 (conflict-free consecutive addresses)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
template <int MODE>
__global__ void __launch_bounds__(256) k(float* out, int iters) {
  __shared__ unsigned long long buf64[4096];
  float* bf = reinterpret_cast<float*>(buf64);
  unsigned* bu = reinterpret_cast<unsigned*>(buf64);
  for (int i = threadIdx.x; i < 4096; i += 256) buf64[i] = 0;
  __syncthreads();
  const int t = threadIdx.x;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int idx = (t + 37 * u + it * 5) & 4095;
      if (MODE == 0) atomicAdd(&bf[idx], 1.0f + u);
      else if (MODE == 1) atomicAdd(&bu[idx], 1u + u);
      else if (MODE == 2) atomicAdd(&buf64[idx], 1ull + u);
      else if (MODE == 3) bf[idx] += 1.0f + u;
      else if (MODE == 4) atomicMax(&bu[idx], (unsigned)(it + u));
    }
  }
  __syncthreads();
  if (t == 0) out[blockIdx.x] = bf[5] + (float)bu[7];
}
template <int MODE> void run(const char* name, float* d) {
  const int iters = 2000, blocks = 1024;
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, 10);
  hipEventRecord(a);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, iters);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  double ops = (double)blocks * 256 * iters * 8;
  printf("%-12s %8.3f ms  %8.1f G lane-ops/s  (%.2f lanes/clk/CU @2.4GHz,256CU)\n", name, ms, ops / ms / 1e6,
         ops / (ms * 1e-3) / 256 / 2.4e9);
}
int main() {
  float* d; hipMalloc(&d, 4096 * 4);
  run<0>("ds_add_f32", d); run<1>("ds_add_u32", d); run<2>("ds_add_u64", d); run<3>("plain_rmw", d); run<4>("ds_max_u32", d);
  return 0;
}
