// A7: Gram matrix G = F^T F, the style loss and its gradient dF = 2 F D
// (styler_base.py:96-102, 152-185) on the f32 MFMA (v_mfma_f32_32x32x2_f32).
//
// gram_fwd: M = N = C channels, K = pixels (up to 40000).  A wave is one independent work
//   unit = (image, 64x64 tile of G, slab of pixels): it streams the two 64-channel column
//   strips of F straight from global memory -- lane (i,h) reads the float2 F[p+h][c0+2i..]
//   (32 lanes = one 256-B row segment, fully coalesced), feeding two MFMA row blocks with
//   MFMA row i <-> channel c0+2i+q -- accumulates 2x2 MFMA tiles and adds its partial tile to
//   G with float atomics.  No LDS: every operand is used by exactly one wave.
// gram_bwd: dF_b = 2 s_b F_b D_b is a plain batched GEMM (M = pixels, N = K = C): it runs on the LDS-staged
//   batched f32-MFMA GEMM of winograd.hip (D is symmetric: its rows are read as columns), ReLU mask fused.
#include "common.h"

namespace nfs {

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct GramArgs {
  const float* F;
  float* G;
  float* ws;    // partial tiles [unit][64*64] (two-pass mode) or null (float atomics into G)
  const float* scale_dev;
  float scale;
  int B, HW, C;
  int slab;     // pixels per wave (even)
  int nslab;    // slabs per image
  int ntile;    // C / 64
};

// One BLOCK per work unit: its 4 waves split the unit's pixel slab four ways (4 waves per SIMD
// resident => the dependent global loads of the k loop overlap across waves; with one wave per
// unit the kernel was HBM-latency-bound at 22 TF/s), then waves 1-3 park their accumulators in LDS
// (lane-contiguous, conflict-free) and wave 0 adds them and emits the tile.
__global__ void __launch_bounds__(256) gram_fwd_kernel(GramArgs a) {
  __shared__ float part[3][64][64];   // [wave-1][acc register][lane]
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int i = lane & 31, h = lane >> 5;
  int64_t unit = blockIdx.x;
  const int64_t per_img = (int64_t)a.nslab * a.ntile * a.ntile;
  const int b = (int)(unit / per_img);
  unit -= (int64_t)b * per_img;
  const int pair = (int)(unit / a.nslab);
  const int sl = (int)(unit - (int64_t)pair * a.nslab);
  const int t1 = pair / a.ntile, t2 = pair - t1 * a.ntile;
  const int s0 = sl * a.slab;
  const int s1 = min(s0 + a.slab, a.HW);
  // this wave's quarter (even number of pixels per quarter so that pairs never straddle waves)
  const int quarter = ((((s1 - s0) + 1) / 2 + 3) / 4) * 2;
  const int p0 = min(s0 + wid * quarter, s1);
  const int p1 = min(p0 + quarter, s1);
  const float* Fb = a.F + (int64_t)b * a.HW * a.C;
  const float* pa = Fb + t1 * 64 + 2 * i;
  const float* pb = Fb + t2 * 64 + 2 * i;

  f32x16 acc[2][2];
#pragma unroll
  for (int x = 0; x < 2; ++x)
#pragma unroll
    for (int y = 0; y < 2; ++y)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[x][y][r] = 0.f;

  // k loop, manually software-pipelined (hipcc refuses to unroll a loop around the MFMA builtin):
  // 8 k-steps (= 16 pixels, 16 float2 loads) are in flight while the previous 8 are multiplied.
  const int nsteps = (p1 - p0) >> 1;
  const float* qa_ = pa + (int64_t)(p0 + h) * a.C;
  const float* qb_ = pb + (int64_t)(p0 + h) * a.C;
  const int64_t step = 2 * (int64_t)a.C;
#define NFS_GRAM_LOAD(A_, B_, s_)                                                      \
  _Pragma("unroll") for (int u = 0; u < 8; ++u) {                                      \
    A_[u] = make_float2(0.f, 0.f); B_[u] = make_float2(0.f, 0.f);                      \
    if ((s_) + u < nsteps) {                                                           \
      A_[u] = *reinterpret_cast<const float2*>(qa_ + ((s_) + u) * step);               \
      B_[u] = *reinterpret_cast<const float2*>(qb_ + ((s_) + u) * step);               \
    }                                                                                  \
  }
#define NFS_GRAM_MMA(A_, B_)                                                                      \
  _Pragma("unroll") for (int u = 0; u < 8; ++u) {                                                 \
    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(A_[u].x, B_[u].x, acc[0][0], 0, 0, 0);      \
    acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(A_[u].x, B_[u].y, acc[0][1], 0, 0, 0);      \
    acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(A_[u].y, B_[u].x, acc[1][0], 0, 0, 0);      \
    acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(A_[u].y, B_[u].y, acc[1][1], 0, 0, 0);      \
  }
  {
    float2 A0[8], B0[8], A1[8], B1[8];
    NFS_GRAM_LOAD(A0, B0, 0)
    for (int s0_ = 0; s0_ < nsteps; s0_ += 16) {
      NFS_GRAM_LOAD(A1, B1, s0_ + 8)
      NFS_GRAM_MMA(A0, B0)
      NFS_GRAM_LOAD(A0, B0, s0_ + 16)
      NFS_GRAM_MMA(A1, B1)
    }
  }
#undef NFS_GRAM_LOAD
#undef NFS_GRAM_MMA
  if ((p1 - p0) & 1) {  // odd tail pixel: the k=1 half contributes zero
    const int p = p1 - 1;
    float2 av = make_float2(0.f, 0.f), bv = make_float2(0.f, 0.f);
    if (h == 0) {
      av = *reinterpret_cast<const float2*>(pa + (int64_t)p * a.C);
      bv = *reinterpret_cast<const float2*>(pb + (int64_t)p * a.C);
    }
    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, bv.x, acc[0][0], 0, 0, 0);
    acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, bv.y, acc[0][1], 0, 0, 0);
    acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, bv.x, acc[1][0], 0, 0, 0);
    acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, bv.y, acc[1][1], 0, 0, 0);
  }
  // block reduction of the 4 partial tiles
  if (wid > 0) {
#pragma unroll
    for (int qa = 0; qa < 2; ++qa)
#pragma unroll
      for (int qb = 0; qb < 2; ++qb)
#pragma unroll
        for (int r = 0; r < 16; ++r) part[wid - 1][(qa * 2 + qb) * 16 + r][lane] = acc[qa][qb][r];
  }
  __syncthreads();
  if (wid > 0) return;
#pragma unroll
  for (int qa = 0; qa < 2; ++qa)
#pragma unroll
    for (int qb = 0; qb < 2; ++qb)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int k = (qa * 2 + qb) * 16 + r;
        acc[qa][qb][r] += part[0][k][lane] + part[1][k][lane] + part[2][k][lane];
      }
  // acc[qa][qb][r]: G row = t1*64 + 2*row + qa, col = t2*64 + 2*(lane&31) + qb
  if (a.ws) {
    // two-pass mode: the raw partial tile goes to the workspace (float2 = 256-B coalesced rows);
    // gram_reduce_kernel sums the slabs in a fixed order (deterministic, no atomics)
    float* wt = a.ws + (int64_t)blockIdx.x * 4096;
#pragma unroll
    for (int qa = 0; qa < 2; ++qa)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = 2 * ((r & 3) + 8 * (r >> 2) + 4 * h) + qa;
        *reinterpret_cast<float2*>(wt + row * 64 + 2 * i) = make_float2(acc[qa][0][r], acc[qa][1][r]);
      }
    return;
  }
  const float sc = a.scale * (a.scale_dev ? a.scale_dev[b] : 1.f);
  float* Gb = a.G + (int64_t)b * a.C * a.C;
#pragma unroll
  for (int qa = 0; qa < 2; ++qa)
#pragma unroll
    for (int qb = 0; qb < 2; ++qb)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
        atomicAdd(Gb + (int64_t)(t1 * 64 + 2 * row + qa) * a.C + t2 * 64 + 2 * i + qb, acc[qa][qb][r] * sc);
      }
}

// G[b][c1][c2] = scale_b * sum_slab ws[((b*npair + pair)*nslab + slab)][64x64 tile]
__global__ void __launch_bounds__(256) gram_reduce_kernel(GramArgs a) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;   // element of G
  const int64_t per_img = (int64_t)a.C * a.C;
  if (e >= per_img * a.B) return;
  const int b = (int)(e / per_img);
  const int rem = (int)(e - (int64_t)b * per_img);
  const int c1 = rem / a.C, c2 = rem - c1 * a.C;
  const int pair = (c1 >> 6) * a.ntile + (c2 >> 6);
  const float* p = a.ws + (((int64_t)b * a.ntile * a.ntile + pair) * a.nslab) * 4096 + (c1 & 63) * 64 + (c2 & 63);
  float s = 0.f;
  for (int k = 0; k < a.nslab; ++k) s += p[(int64_t)k * 4096];
  a.G[e] = s * a.scale * (a.scale_dev ? a.scale_dev[b] : 1.f);
}

// loss += weight * sum (G - Gs)^2 ; Dmat = 2*weight*(G - Gs)
__global__ void __launch_bounds__(256) style_loss_kernel(const float* __restrict__ G, const float* __restrict__ Gs,
                                                         float* __restrict__ loss, float* __restrict__ Dmat, int B,
                                                         int Bs, int CC, float weight) {
  __shared__ float red[16];
  const int b = blockIdx.y;
  float part = 0.f;
  // grid-stride: at most 32 blocks per image => at most 32 same-address atomics per image
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < CC; e += (int64_t)gridDim.x * blockDim.x) {
    const float diff = G[(int64_t)b * CC + e] - Gs[(int64_t)(b % Bs) * CC + e];
    part += weight * diff * diff;
    Dmat[(int64_t)b * CC + e] = 2.f * weight * diff;
  }
  part = block_sum(part, red);
  if (threadIdx.x == 0) atomicAdd(loss + b, part);
}

// winograd.hip: batched f32-MFMA GEMM (LDS-staged, double-buffered) shared with the Winograd convolution
int gram_bwd_gemm(const float* F, const float* Dm, float* dF, int B, int HW, int C, float alpha, const float* alpha_dev,
                  int relu_mask, int cus, hipStream_t s);

}  // namespace nfs

using namespace nfs;

extern "C" {

static void gram_plan(GramArgs& a) {
  a.ntile = a.C / 64;
  // one block (4 waves) per unit, 3 blocks resident per CU (48 KB of LDS each): aim at ~3 units per
  // CU, slabs of >= 128 pixels
  const int64_t pairs = (int64_t)a.B * a.ntile * a.ntile;
  int64_t want = (3 * 256 + pairs - 1) / pairs;
  if (want < 1) want = 1;
  int slab = (int)((a.HW + want - 1) / want);
  if (slab < 128) slab = 128;
  slab = (slab + 1) & ~1;
  a.slab = slab;
  a.nslab = (a.HW + slab - 1) / slab;
}

int64_t nfs_gram_workspace_floats(int B, int HW, int C) {
  if (B <= 0 || HW <= 0 || C <= 0 || C % 64) return 0;
  GramArgs a;
  a.B = B; a.HW = HW; a.C = C;
  gram_plan(a);
  return (int64_t)B * a.ntile * a.ntile * a.nslab * 4096 + 4 * 4096;
}

int nfs_gram_fwd(const float* F, float* G, int B, int HW, int C, const float* scale_dev, float scale,
                 float* workspace, int64_t workspace_floats, nfs_stream_t stream) {
  NFS_REQUIRE(F && G, "nfs_gram_fwd: null pointer");
  NFS_REQUIRE(B > 0 && HW > 0, "nfs_gram_fwd: non-positive dimension");
  NFS_REQUIRE(C > 0 && C % 64 == 0, "nfs_gram_fwd: C must be a multiple of 64");
  GramArgs a;
  a.F = F; a.G = G; a.scale_dev = scale_dev; a.scale = scale; a.B = B; a.HW = HW; a.C = C;
  gram_plan(a);
  const int64_t units = (int64_t)B * a.ntile * a.ntile * a.nslab;
  a.ws = (workspace && workspace_floats >= nfs_gram_workspace_floats(B, HW, C)) ? workspace : nullptr;
  hipLaunchKernelGGL(gram_fwd_kernel, dim3((unsigned)units), dim3(256), 0, as_stream(stream), a);
  if (a.ws)
    hipLaunchKernelGGL(gram_reduce_kernel, dim3(blocks_for((int64_t)B * C * C, 256)), dim3(256), 0, as_stream(stream),
                       a);
  return check_launch("nfs_gram_fwd");
}

int nfs_style_loss_fwd(const float* G, const float* Gs, float* loss_acc, float* Dmat, int B, int Bs, int C,
                       float weight, nfs_stream_t stream) {
  NFS_REQUIRE(G && Gs && loss_acc && Dmat, "nfs_style_loss_fwd: null pointer");
  NFS_REQUIRE(B > 0 && Bs > 0 && C > 0, "nfs_style_loss_fwd: non-positive dimension");
  const int CC = C * C;
  const unsigned nb = blocks_for(CC, 256) < 32u ? blocks_for(CC, 256) : 32u;
  hipLaunchKernelGGL(style_loss_kernel, dim3(nb, B), dim3(256), 0, as_stream(stream), G, Gs, loss_acc, Dmat, B, Bs,
                     CC, weight);
  return check_launch("nfs_style_loss_fwd");
}

int nfs_gram_bwd(const float* F, const float* Dmat, float* dF, int B, int HW, int C, const float* scale_dev,
                 float scale, int relu_mask, nfs_stream_t stream) {
  NFS_REQUIRE(F && Dmat && dF, "nfs_gram_bwd: null pointer");
  NFS_REQUIRE(B > 0 && HW > 0, "nfs_gram_bwd: non-positive dimension");
  NFS_REQUIRE(C > 0 && C % 64 == 0, "nfs_gram_bwd: C must be a multiple of 64");
  // dF[b] = 2 * scale_b * F[b] @ D[b]: M = pixels, N = K = C; D is symmetric, so its rows serve as columns
  static int cus = 0;
  if (cus == 0) { const int c = nfs_device_cus(); cus = c > 0 ? c : 256; }
  return gram_bwd_gemm(F, Dmat, dF, B, HW, C, 2.f * scale, scale_dev, relu_mask, cus, as_stream(stream));
}

}  // extern "C"
