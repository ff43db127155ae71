// A7: Gram matrix G = F^T F, the style loss and its gradient dF = 2 F D
// (styler_base.py:96-102, 152-185) on the f32 MFMA (v_mfma_f32_32x32x2_f32).
//
// gram_fwd: M = N = C channels, K = pixels (up to 40000).  A wave is one independent work
//   unit = (image, 64x64 tile of G, slab of pixels): it streams the two 64-channel column
//   strips of F straight from global memory -- lane (i,h) reads the float2 F[p+h][c0+2i..]
//   (32 lanes = one 256-B row segment, fully coalesced), feeding two MFMA row blocks with
//   MFMA row i <-> channel c0+2i+q -- accumulates 2x2 MFMA tiles and adds its partial tile to
//   G with float atomics.  No LDS: every operand is used by exactly one wave.
// gram_bwd: M = pixels, N = K = C.  A wave = (image, 64 pixels, 64 output channels); A rows are
//   b128 reads of F (4 consecutive k per lane feed 4 MFMA steps, as in the conv kernel).
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

__global__ void __launch_bounds__(256) gram_fwd_kernel(GramArgs a) {
  const int lane = threadIdx.x & 63;
  const int i = lane & 31, h = lane >> 5;
  int64_t unit = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int64_t per_img = (int64_t)a.nslab * a.ntile * a.ntile;
  if (unit >= per_img * a.B) return;
  const int b = (int)(unit / per_img);
  unit -= (int64_t)b * per_img;
  const int pair = (int)(unit / a.nslab);
  const int sl = (int)(unit - (int64_t)pair * a.nslab);
  const int t1 = pair / a.ntile, t2 = pair - t1 * a.ntile;
  const int p0 = sl * a.slab;
  const int p1 = min(p0 + a.slab, a.HW);
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

  int p = p0;
#pragma unroll 4
  for (; p + 1 < p1; p += 2) {
    const float2 av = *reinterpret_cast<const float2*>(pa + (int64_t)(p + h) * a.C);
    const float2 bv = *reinterpret_cast<const float2*>(pb + (int64_t)(p + h) * a.C);
    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, bv.x, acc[0][0], 0, 0, 0);
    acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, bv.y, acc[0][1], 0, 0, 0);
    acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, bv.x, acc[1][0], 0, 0, 0);
    acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, bv.y, acc[1][1], 0, 0, 0);
  }
  if (p < p1) {  // odd tail: k=1 half contributes zero
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
  // acc[qa][qb][r]: G row = t1*64 + 2*row + qa, col = t2*64 + 2*(lane&31) + qb
  if (a.ws) {
    // two-pass mode: the raw partial tile goes to the workspace (float2 = 256-B coalesced rows);
    // gram_reduce_kernel sums the slabs in a fixed order (deterministic, no atomics)
    const int64_t full_unit = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    float* wt = a.ws + full_unit * 4096;
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

struct GramBwdArgs {
  const float* F;
  const float* Dm;
  float* dF;
  const float* scale_dev;
  float scale;
  int B, HW, C;
  int npb;   // pixel blocks (64) per image
  int ntile; // C / 64
  int relu_mask;
};

__global__ void __launch_bounds__(256) gram_bwd_kernel(GramBwdArgs a) {
  const int lane = threadIdx.x & 63;
  const int i = lane & 31, h = lane >> 5;
  int64_t unit = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int64_t per_img = (int64_t)a.npb * a.ntile;
  if (unit >= per_img * a.B) return;
  const int b = (int)(unit / per_img);
  unit -= (int64_t)b * per_img;
  const int pb = (int)(unit / a.ntile);
  const int nt0 = (int)(unit - (int64_t)pb * a.ntile);
  const float* Fb = a.F + (int64_t)b * a.HW * a.C;
  const float* Db = a.Dm + (int64_t)b * a.C * a.C;
  const int pbase = pb * 64;
  // A rows: pixels pbase + mt*32 + i (clamped; out-of-range rows are never stored)
  const float* arow[2];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) arow[mt] = Fb + (int64_t)min(pbase + mt * 32 + i, a.HW - 1) * a.C + 4 * h;
  const int ncol = nt0 * 64 + i;

  f32x16 acc[2][2];
#pragma unroll
  for (int x = 0; x < 2; ++x)
#pragma unroll
    for (int y = 0; y < 2; ++y)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[x][y][r] = 0.f;

#pragma unroll 2
  for (int k8 = 0; k8 < a.C; k8 += 8) {
    const float4 a0 = *reinterpret_cast<const float4*>(arow[0] + k8);
    const float4 a1 = *reinterpret_cast<const float4*>(arow[1] + k8);
    float b0[4], b1[4];
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
      const float* dr = Db + (int64_t)(k8 + 4 * h + jj) * a.C + ncol;
      b0[jj] = dr[0];
      b1[jj] = dr[32];
    }
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
      const float x0 = (&a0.x)[jj], x1 = (&a1.x)[jj];
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(x0, b0[jj], acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(x0, b1[jj], acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(x1, b0[jj], acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(x1, b1[jj], acc[1][1], 0, 0, 0);
    }
  }
  const float sc = 2.f * a.scale * (a.scale_dev ? a.scale_dev[b] : 1.f);
  float* dFb = a.dF + (int64_t)b * a.HW * a.C;
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int px = pbase + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
      if (px >= a.HW) continue;
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        const int64_t idx = (int64_t)px * a.C + nt0 * 64 + nt * 32 + i;
        float v = acc[mt][nt][r] * sc;
        if (a.relu_mask) v = Fb[idx] > 0.f ? v : 0.f;
        dFb[idx] = v;
      }
    }
}

}  // namespace nfs

using namespace nfs;

extern "C" {

static void gram_plan(GramArgs& a) {
  a.ntile = a.C / 64;
  // enough waves to fill the chip (>= ~4 per CU), slabs of >= 128 pixels
  const int64_t pairs = (int64_t)a.B * a.ntile * a.ntile;
  int64_t want = (4 * 256 + pairs - 1) / pairs;
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
  hipLaunchKernelGGL(gram_fwd_kernel, dim3(blocks_for(units, 4)), dim3(256), 0, as_stream(stream), a);
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
  GramBwdArgs a;
  a.F = F; a.Dm = Dmat; a.dF = dF; a.scale_dev = scale_dev; a.scale = scale;
  a.B = B; a.HW = HW; a.C = C; a.npb = (HW + 63) / 64; a.ntile = C / 64; a.relu_mask = relu_mask;
  const int64_t units = (int64_t)B * a.npb * a.ntile;
  hipLaunchKernelGGL(gram_bwd_kernel, dim3(blocks_for(units, 4)), dim3(256), 0, as_stream(stream), a);
  return check_launch("nfs_gram_bwd");
}

}  // extern "C"
