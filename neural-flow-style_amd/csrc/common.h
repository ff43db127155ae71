// Shared helpers for libnfs_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>

#include "../../include/nfs_hip.h"

namespace nfs {

void set_error(const char* fmt, ...);

inline hipStream_t as_stream(nfs_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

inline int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: %s", what, hipGetErrorString(e));
    return NFS_ELAUNCH;
  }
  return NFS_OK;
}

#define NFS_REQUIRE(cond, ...)        \
  do {                                \
    if (!(cond)) {                    \
      nfs::set_error(__VA_ARGS__);    \
      return NFS_EINVAL;              \
    }                                 \
  } while (0)

// Timing-only ablation branches (they skip loads / stores / MFMAs: WRONG RESULTS by construction) exist only in builds
// made with -DNFS_ABLATE (make ABLATE=1); in the product library NFS_DBG() is the constant 0 and the branches, the
// environment reads that set them (NFS_GEMM_DBG, NFS_CONV_DBG) and the `dbg` argument fields compile away.
#ifdef NFS_ABLATE
#define NFS_DBG(args, bit) ((args).dbg & (bit))
#else
#define NFS_DBG(args, bit) 0
#endif

inline unsigned blocks_for(int64_t n, int threads) {
  return (unsigned)((n + threads - 1) / threads);
}

// ---- wave / block reductions (wave = 64 lanes) ------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
// block-wide reductions; `red` is >= 16 floats of LDS; result valid in every thread
__device__ __forceinline__ float block_sum(float v, float* red) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  v = wave_sum(v);
  __syncthreads();
  if (lane == 0) red[wid] = v;
  __syncthreads();
  float r = 0.f;
  for (int i = 0; i < nw; ++i) r += red[i];
  return r;
}
// Zero fill as a KERNEL.  A hipMemsetAsync in front of a kernel that accumulates into the same words (atomicMax / atomicAdd)
// is a memset node in a captured hipGraph, and in replays that node did not reliably complete before a SHORT following
// kernel issued its atomics (round 4: an all-zero gradient in one replay of four; the long kernels that used to follow
// such memsets hid it).  Kernel nodes of one captured stream do run in order.
template <int UNUSED = 0>                             // (a template: one definition across the translation units)
__global__ void __launch_bounds__(256) zero_words_kernel(unsigned* __restrict__ x, long long n) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) x[i] = 0u;
}
static inline void zero_words(void* p, long long n_words, hipStream_t s) {
  hipLaunchKernelGGL(zero_words_kernel<0>, dim3((unsigned)((n_words + 255) / 256)), dim3(256), 0, s,
                     reinterpret_cast<unsigned*>(p), n_words);
}

__device__ __forceinline__ float block_max(float v, float* red) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  v = wave_max(v);
  __syncthreads();
  if (lane == 0) red[wid] = v;
  __syncthreads();
  float r = red[0];
  for (int i = 1; i < nw; ++i) r = fmaxf(r, red[i]);
  return r;
}

// ---- trilinear stencil, exactly as transform.py:343-433 -----------------------------
// normalised coord c in [-1,1] -> index space x=(c+1)(n-1)/2; x0=floor, x1=x0+1, both
// clipped to [0,n-1]; weight dx = x - float(clipped x0)  => border replication.
struct Axis {
  int i0, i1;
  float w1;  // dx ; weight of i0 is (1 - dx)
};
__device__ __forceinline__ Axis axis_setup(float c, int n) {
  const float x = (c + 1.f) * (float)(n - 1) * 0.5f;
  float f = floorf(x);
  f = fminf(fmaxf(f, -1.f), (float)n);  // keep the int conversion defined (also for NaN)
  const int i = (int)f;
  Axis a;
  a.i0 = min(max(i, 0), n - 1);
  a.i1 = min(max(i + 1, 0), n - 1);
  a.w1 = x - (float)a.i0;
  return a;
}

struct Tri {
  int64_t o[8];
  float w[8];
};
// offsets in elements of a [X,Y,Z] volume with C channels (multiply outside by C)
__device__ __forceinline__ void tri_setup(float cx, float cy, float cz, int X, int Y, int Z, Tri& t,
                                          Axis& ax, Axis& ay, Axis& az) {
  ax = axis_setup(cx, X);
  ay = axis_setup(cy, Y);
  az = axis_setup(cz, Z);
  const float wx[2] = {1.f - ax.w1, ax.w1}, wy[2] = {1.f - ay.w1, ay.w1}, wz[2] = {1.f - az.w1, az.w1};
  const int ix[2] = {ax.i0, ax.i1}, iy[2] = {ay.i0, ay.i1}, iz[2] = {az.i0, az.i1};
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const int k = a * 4 + b * 2 + c;
        t.o[k] = ((int64_t)ix[a] * Y + iy[b]) * Z + iz[c];
        t.w[k] = wx[a] * wy[b] * wz[c];
      }
}

// 8-byte load of two W-adjacent cells from a 4-byte-aligned address (gfx950 global loads only need
// dword alignment): the two x-corners of a trilinear stencil come from ONE load instead of two,
// which halves the texture-address work of the gather-bound ray march.
struct __attribute__((packed, aligned(4))) F2u { float x, y; };

// the same stencil as tri_setup/tri_sample1, accumulated in the same order (x0,x1 inner)
__device__ __forceinline__ float tri_sample_pairs(const float* __restrict__ vol, int H, int W, const Axis& az,
                                                  const Axis& ay, const Axis& ax) {
  const int xb = min(ax.i0, W - 2);                 // pair base; clamped stencils select below
  const bool x0_lo = ax.i0 == xb, x1_lo = ax.i1 == xb;
  const float wz[2] = {1.f - az.w1, az.w1}, wy[2] = {1.f - ay.w1, ay.w1}, wx[2] = {1.f - ax.w1, ax.w1};
  const int iz[2] = {az.i0, az.i1}, iy[2] = {ay.i0, ay.i1};
  float s = 0.f;
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const F2u p = *reinterpret_cast<const F2u*>(vol + ((int64_t)iz[a] * H + iy[b]) * W + xb);
      const float v0 = x0_lo ? p.x : p.y, v1 = x1_lo ? p.x : p.y;
      s += wz[a] * wy[b] * wx[0] * v0;
      s += wz[a] * wy[b] * wx[1] * v1;
    }
  return s;
}


// the 8 corner values of the stencil in tri_setup order k = a*4 + b*2 + c (x-pairs from 8-byte loads)
__device__ __forceinline__ void tri_gather_pairs(const float* __restrict__ vol, int H, int W, const Axis& az,
                                                 const Axis& ay, const Axis& ax, float* v) {
  const int xb = min(ax.i0, W - 2);
  const bool x0_lo = ax.i0 == xb, x1_lo = ax.i1 == xb;
  const int iz[2] = {az.i0, az.i1}, iy[2] = {ay.i0, ay.i1};
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const F2u p = *reinterpret_cast<const F2u*>(vol + ((int64_t)iz[a] * H + iy[b]) * W + xb);
      v[a * 4 + b * 2] = x0_lo ? p.x : p.y;
      v[a * 4 + b * 2 + 1] = x1_lo ? p.x : p.y;
    }
}

__device__ __forceinline__ float lin_coord(int i, int n) {
  // tf.linspace(-1, 1, n)[i] = -1 + i*step, step = 2/(n-1)   (transform.py:171-177)
  const float step = n > 1 ? 2.f / (float)(n - 1) : 0.f;
  return -1.f + step * (float)i;
}

}  // namespace nfs
