// LDS atomic throughput on gfx950: ns per wave instruction per CU for returnless ds_add_u32 / ds_add_u64 / ds_add_f32,
// conflict-free (lane i -> cell i) and with the rotate adjoint's pattern (8 rows of 8 consecutive cells, row stride 42).
// hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics lds_atomic_rate.hip -o lds_atomic_rate && ./lds_atomic_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int KIND, int PATTERN>
__global__ void __launch_bounds__(1024) k(float* out, int iters) {
  __shared__ unsigned long long cells[8192];
  for (int i = threadIdx.x; i < 8192; i += 1024) cells[i] = 0;
  __syncthreads();
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  int idx = PATTERN == 0 ? lane : (lane >> 3) * 42 + (lane & 7);
  idx += w * 400;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int j = idx + ((u & 1) + (u >> 1 & 1) * 42 + (u >> 2) * 588) % 1200;
      if (KIND == 0) atomicAdd(reinterpret_cast<unsigned*>(cells) + 2 * j, 1u);
      if (KIND == 1) atomicAdd(cells + j, 1ull);
      if (KIND == 2) atomicAdd(reinterpret_cast<float*>(cells) + 2 * j, 1.f);
      if (KIND == 3) { atomicAdd(reinterpret_cast<unsigned*>(cells) + 2 * j, 1u); atomicAdd(reinterpret_cast<unsigned*>(cells) + 2 * j + 1, 1u); }
    }
  }
  __syncthreads();
  out[blockIdx.x * 1024 + threadIdx.x] = (float)cells[threadIdx.x];
}
int main() {
  float* out;
  hipMalloc(&out, 512 * 1024 * 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 2000;
#define RUN(KIND, PAT, label)                                                                    \
  for (int rep = 0; rep < 2; ++rep) {                                                            \
    hipEventRecord(e0, 0);                                                                       \
    hipLaunchKernelGGL((k<KIND, PAT>), dim3(512), dim3(1024), 0, 0, out, iters);                 \
    hipEventRecord(e1, 0);                                                                       \
    hipDeviceSynchronize();                                                                      \
    float ms = 0.f; hipEventElapsedTime(&ms, e0, e1);                                            \
    if (rep) printf("%-40s %.3f ms: %.2f ns per wave atomic per CU\n", label, ms,                \
                    ms * 1e6 / ((double)iters * 8 * 32));                                        \
  }
  RUN(0, 0, "ds_add_u32 lane->cell");
  RUN(1, 0, "ds_add_u64 lane->cell");
  RUN(2, 0, "ds_add_f32 lane->cell");
  RUN(3, 0, "2 x ds_add_u32 lane->cell");
  RUN(0, 1, "ds_add_u32 8 rows x 8");
  RUN(1, 1, "ds_add_u64 8 rows x 8");
  RUN(2, 1, "ds_add_f32 8 rows x 8");
  return 0;
}
