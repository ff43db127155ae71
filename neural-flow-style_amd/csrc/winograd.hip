// A6, wide layers: 3x3 convolution by Winograd F(m x m, 3x3) in float32, m = 4 (default) or 2.
//
//   Y = A^T [ (G g G^T) (.) (B^T d B) ] A      per m x m output tile, summed over input channels,
//
// i.e. (m+2)^2 independent GEMMs  M_k[tile, co] = sum_ci V_k[tile, ci] * U_k[ci, co]  instead of 9 taps:
// 36 multiplies per 16 outputs (m = 4: 4x fewer MFMA flops than the direct implicit GEMM of vgg.hip) or 16 per
// 4 outputs (m = 2: 2.25x fewer).  All arithmetic stays float32 (products on the exact-f32 MFMA); the result
// differs from the direct form only by f32 rounding: ~4e-7 relative L2 for m = 2 (same as direct), ~3e-6 for
// m = 4 with the standard interpolation points 0, +-1, +-2, inf (what cuDNN's fp32 Winograd, which the
// reference runs on, uses as well).  NFS_WINOGRAD_TILE=2 selects m = 2.
// Used for layers with >= 64 input and output channels (conv1_2 ... conv5_1 and their data gradients).
//
//   winograd_input*_kernel   x [B,H,W,K]            -> V [(m+2)^2][T][K]    T = B * ceil(H/m) * ceil(W/m) tiles
//   winograd_gemm_kernel     V, U [(m+2)^2][K/32][N][32] -> M [(m+2)^2][T][N]  f32 MFMA, fragment scheme of vgg.hip
//   winograd_output*_kernel  M                      -> y [B,H,W,N]    + bias/ReLU (fwd) or ReLU mask/addend (dgrad)
#include "common.h"
#include "winograd_math.h"
#include "winograd_gemm.h"

#include <algorithm>
#include <atomic>
#include <map>
#include <mutex>
#include <tuple>
#include <vector>

namespace nfs {

// winograd_fused.hip
bool winograd_fusable(int K, int N);
bool winograd_fused_takes(int B, int H, int W, int K, int N);
int64_t winograd_fused_packed_floats(int K, int N);
int winograd_pack_fused(const float* up, float* uf, int K, int N, hipStream_t s);
int winograd_fused_conv(const float* x, const float* Uf, const float* aux0, const float* aux1, float* y, int B, int H,
                        int W, int K, int N, int mode, int relu, hipStream_t s, float* ypool, const float* xmask,
                        uint32_t* in_bits, uint32_t* out_bits, bool pooled_grad);

// winograd5.hip
bool winograd5_channels(int K, int N);
bool winograd5_takes(int H, int W, int K, int N);
int64_t winograd5_packed_floats(int Ci, int Co);
int64_t winograd5_workspace_floats(int B, int H, int W, int K, int N);
int winograd5_pack(const float* w_hwio, float* up5, int Ci, int Co, int kind, hipStream_t s);
int64_t winograd5_bits_words(int B, int H, int W, int C);
int winograd5_conv(const float* x, const float* U5, const float* aux0, const float* aux1, float* y, float* ws, int B,
                   int H, int W, int K, int N, int mode, int relu, int cus, hipStream_t s, uint32_t* in_bits);

typedef float f32x16 __attribute__((ext_vector_type(16)));

// ---- weights: U_k[ci][co] = (G g G^T)[k], packed [16][K/32][N][32] --------------------------------
// kind 0: GEMM K = Ci, N = Co, g = w[:, :, ci, co];  kind 1 (data gradient): K = Co, N = Ci,
// g[r][s] = w[2-r][2-s][ci][co] (taps flipped), k index = co, n index = ci.
__global__ void __launch_bounds__(256) winograd_pack_kernel(const float* __restrict__ w, float* __restrict__ up,
                                                            int Ci, int Co, int kind) {
  const int Kc = kind == 0 ? Ci : Co, Nc = kind == 0 ? Co : Ci;
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= (int64_t)Kc * Nc) return;
  const int n = (int)(gid % Nc), k = (int)(gid / Nc);
  float g[3][3];
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int s = 0; s < 3; ++s) {
      if (kind == 0) g[r][s] = w[((int64_t)(r * 3 + s) * Ci + k) * Co + n];
      else g[r][s] = w[((int64_t)((2 - r) * 3 + (2 - s)) * Ci + n) * Co + k];
    }
  // G = [[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]]
  float t[4][3];
#pragma unroll
  for (int s = 0; s < 3; ++s) {
    t[0][s] = g[0][s];
    t[1][s] = 0.5f * (g[0][s] + g[1][s] + g[2][s]);
    t[2][s] = 0.5f * (g[0][s] - g[1][s] + g[2][s]);
    t[3][s] = g[2][s];
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const float u[4] = {t[r][0], 0.5f * (t[r][0] + t[r][1] + t[r][2]), 0.5f * (t[r][0] - t[r][1] + t[r][2]), t[r][2]};
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int comp = r * 4 + s;
      up[(((int64_t)comp * (Kc / 32) + k / 32) * Nc + n) * 32 + (k & 31)] = u[s];
    }
  }
}

// ---- input transform: one thread = one tile x 4 channels --------------------------------------------
__global__ void __launch_bounds__(256) winograd_input_kernel(const float* __restrict__ x, float* __restrict__ V,
                                                             int B, int H, int W, int K, int TH, int TW) {
  const int K4 = K >> 2;
  const int64_t T = (int64_t)B * TH * TW;
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= T * K4) return;
  const int c4 = (int)(gid % K4);
  const int64_t tile = gid / K4;
  const int tx = (int)(tile % TW), ty = (int)((tile / TW) % TH), b = (int)(tile / ((int64_t)TW * TH));
  const int y0 = 2 * ty - 1, x0 = 2 * tx - 1;
  float4 d[4][4];
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int yy = y0 + r, xx = x0 + s;
      d[r][s] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (yy >= 0 && yy < H && xx >= 0 && xx < W)
        d[r][s] = *reinterpret_cast<const float4*>(x + (((int64_t)b * H + yy) * W + xx) * K + 4 * c4);
    }
  // B^T = [[1,0,-1,0],[0,1,1,0],[0,-1,1,0],[0,1,0,-1]]   V = B^T d B
#define NFS_F4(op, a, b) make_float4(a.x op b.x, a.y op b.y, a.z op b.z, a.w op b.w)
  float4 t[4][4];
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    t[0][s] = NFS_F4(-, d[0][s], d[2][s]);
    t[1][s] = NFS_F4(+, d[1][s], d[2][s]);
    t[2][s] = NFS_F4(-, d[2][s], d[1][s]);
    t[3][s] = NFS_F4(-, d[1][s], d[3][s]);
  }
  const int64_t comp_stride = T * K;
  float* vo = V + tile * K + 4 * c4;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const float4 o0 = NFS_F4(-, t[r][0], t[r][2]);
    const float4 o1 = NFS_F4(+, t[r][1], t[r][2]);
    const float4 o2 = NFS_F4(-, t[r][2], t[r][1]);
    const float4 o3 = NFS_F4(-, t[r][1], t[r][3]);
    *reinterpret_cast<float4*>(vo + (int64_t)(r * 4 + 0) * comp_stride) = o0;
    *reinterpret_cast<float4*>(vo + (int64_t)(r * 4 + 1) * comp_stride) = o1;
    *reinterpret_cast<float4*>(vo + (int64_t)(r * 4 + 2) * comp_stride) = o2;
    *reinterpret_cast<float4*>(vo + (int64_t)(r * 4 + 3) * comp_stride) = o3;
  }
}

// ---- F(4x4, 3x3): weights U_k = (G g G^T)[k], k = 0..35, packed [36][K/32][N][32] ----------------------
// G = [[1/4,0,0],[-1/6,-1/6,-1/6],[-1/6,1/6,-1/6],[1/24,1/12,1/6],[1/24,-1/12,1/6],[0,0,1]]
__device__ __forceinline__ void wg4_g(const float g0, const float g1, const float g2, float* u) {
  u[0] = 0.25f * g0;
  u[1] = (-1.f / 6.f) * (g0 + g1 + g2);
  u[2] = (-1.f / 6.f) * (g0 - g1 + g2);
  u[3] = (1.f / 24.f) * g0 + (1.f / 12.f) * g1 + (1.f / 6.f) * g2;
  u[4] = (1.f / 24.f) * g0 - (1.f / 12.f) * g1 + (1.f / 6.f) * g2;
  u[5] = g2;
}

__global__ void __launch_bounds__(256) winograd_pack4_kernel(const float* __restrict__ w, float* __restrict__ up,
                                                             int Ci, int Co, int kind) {
  const int Kc = kind == 0 ? Ci : Co, Nc = kind == 0 ? Co : Ci;
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= (int64_t)Kc * Nc) return;
  const int n = (int)(gid % Nc), k = (int)(gid / Nc);
  float g[3][3];
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int s = 0; s < 3; ++s) {
      if (kind == 0) g[r][s] = w[((int64_t)(r * 3 + s) * Ci + k) * Co + n];
      else g[r][s] = w[((int64_t)((2 - r) * 3 + (2 - s)) * Ci + n) * Co + k];
    }
  float t[6][3];
#pragma unroll
  for (int s = 0; s < 3; ++s) {
    float u[6];
    wg4_g(g[0][s], g[1][s], g[2][s], u);
#pragma unroll
    for (int r = 0; r < 6; ++r) t[r][s] = u[r];
  }
#pragma unroll
  for (int r = 0; r < 6; ++r) {
    float u[6];
    wg4_g(t[r][0], t[r][1], t[r][2], u);
#pragma unroll
    for (int q = 0; q < 6; ++q)
      up[(((int64_t)(r * 6 + q) * (Kc / 32) + k / 32) * Nc + n) * 32 + (k & 31)] = u[q];
  }
}

// input transform: one thread = one 6x6 patch x 2 channels (a wave covers 128 contiguous channels per pixel)
// POOLED: the operand is not stored -- it is the gradient coming through a 2x2 VALID average pool below a ReLU,
// d(y, x) = 0.25 * gpool[y/2, x/2] * (xmask[y, x] > 0) (0 outside the pooled area), formed on the fly
// (the separate avgpool adjoint kernel and its full-resolution round trip through HBM disappear).
// POOLED: 0 plain operand; 1 pooled gradient masked from the ReLU bit cache; 2 ... from the float output xmask.
// Every load is unconditional (clamped address, the value dropped by a select afterwards): a bounds test around a load
// is a branch, hipcc drains vmcnt at every join, and the 36 loads of a patch went out as seven dependent round trips
// (conv5_1 / conv4_4 at 8 views are a few hundred thousand threads: latency is all there is).
template <int POOLED>
__global__ void __launch_bounds__(256) winograd_input4_kernel(const float* __restrict__ x, float* __restrict__ V,
                                                              int B, int H, int W, int K, int TH, int TW,
                                                              const float* __restrict__ xmask,
                                                              uint32_t* __restrict__ bits) {
  // ReLU bit cache (word = one 4x4 tile x one channel pair, bit ((row*4 + col)*2 + channel) = value > 0):
  //   POOLED 0: `bits` (nullable) is WRITTEN with the mask of the tile's own 4x4 input pixels -- the forward pass of a
  //             layer records (x > 0) for its data gradient, which then never reads x again;
  //   POOLED 1: `bits` is READ instead of xmask (recorded by the forward output transform of this layer).
  const int K2 = K >> 1;
  const int64_t T = (int64_t)B * TH * TW;
  // 6x6 input patches of neighbouring tiles overlap by two pixels: give each XCD a contiguous range of tiles
  // (workgroups are dealt round-robin to the 8 XCDs), or every shared pixel is fetched from HBM into two L2s
  // (PMC: 2.0x / 2.5x the compulsory read bytes with the plain order)
  const unsigned per_xcd = gridDim.x / 8;
  const unsigned lb = (blockIdx.x % 8) * per_xcd + blockIdx.x / 8;
  const int64_t gid = (int64_t)lb * blockDim.x + threadIdx.x;
  if (gid >= T * K2) return;
  const int c2 = (int)(gid % K2);
  const int64_t tile = gid / K2;
  const int tx = (int)(tile % TW), ty = (int)((tile / TW) % TH), b = (int)(tile / ((int64_t)TW * TH));
  const int y0 = 4 * ty - 1, x0 = 4 * tx - 1;
  const int PH = H >> 1, PW = W >> 1;
  float2 d[6][6];         // d[s][r]: patch column s, row r
  [[maybe_unused]] float2 mk[6][6];
  [[maybe_unused]] uint32_t nb[3][3];      // POOLED 1: the words of the 3 x 3 tiles the 6 x 6 patch reaches into
  if (POOLED == 1) {
#pragma unroll
    for (int dy = 0; dy < 3; ++dy)
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) {
        const int ny = min(max(ty + dy - 1, 0), TH - 1), nx = min(max(tx + dx - 1, 0), TW - 1);
        nb[dy][dx] = bits[(((int64_t)b * TH + ny) * TW + nx) * K2 + c2];
      }
  }
#pragma unroll
  for (int s = 0; s < 6; ++s)
#pragma unroll
    for (int r = 0; r < 6; ++r) {
      const int yy = y0 + r, xx = x0 + s;
      if (!POOLED) {
        const int yc = min(max(yy, 0), H - 1), xc = min(max(xx, 0), W - 1);
        d[s][r] = *reinterpret_cast<const float2*>(x + (((int64_t)b * H + yc) * W + xc) * K + 2 * c2);
      } else {
        const int yc = min(max(yy, 0) >> 1, PH - 1), xc = min(max(xx, 0) >> 1, PW - 1);
        d[s][r] = *reinterpret_cast<const float2*>(x + (((int64_t)b * PH + yc) * PW + xc) * K + 2 * c2);
        if (POOLED == 2) {
          const int ym = min(max(yy, 0), H - 1), xm = min(max(xx, 0), W - 1);
          mk[s][r] = *reinterpret_cast<const float2*>(xmask + (((int64_t)b * H + ym) * W + xm) * K + 2 * c2);
        }
      }
    }
  float2 t[6][6];   // t[s][r]: column s after the vertical pass
  uint32_t word = 0u;
#pragma unroll
  for (int s = 0; s < 6; ++s) {
    const int xx = x0 + s;
#pragma unroll
    for (int r = 0; r < 6; ++r) {
      const int yy = y0 + r;
      float2 v = d[s][r];
      if (!POOLED) {
        const bool ok = yy >= 0 && yy < H && xx >= 0 && xx < W;
        v = make_float2(ok ? v.x : 0.f, ok ? v.y : 0.f);
        if (r >= 1 && r <= 4 && s >= 1 && s <= 4)
          word |= ((v.x > 0.f ? 1u : 0u) | (v.y > 0.f ? 2u : 0u)) << (((r - 1) * 4 + (s - 1)) * 2);
      } else {
        const bool ok = yy >= 0 && xx >= 0 && (yy >> 1) < PH && (xx >> 1) < PW;
        bool mx, my;
        if (POOLED == 1) {
          // patch row r: tile offset / row inside that tile (row 0 = last row of the tile above, 5 = first below)
          const int dy = r == 0 ? 0 : (r == 5 ? 2 : 1), ir = r == 0 ? 3 : (r == 5 ? 0 : r - 1);
          const int dx = s == 0 ? 0 : (s == 5 ? 2 : 1), is = s == 0 ? 3 : (s == 5 ? 0 : s - 1);
          const uint32_t wv = nb[dy][dx] >> ((ir * 4 + is) * 2);
          mx = wv & 1u;
          my = wv & 2u;
        } else {
          mx = mk[s][r].x > 0.f;
          my = mk[s][r].y > 0.f;
        }
        v = make_float2((ok && mx) ? 0.25f * v.x : 0.f, (ok && my) ? 0.25f * v.y : 0.f);
      }
      d[s][r] = v;
    }
    wg4_bt(d[s], t[s]);
  }
  if (!POOLED && bits) bits[gid] = word;
  const int64_t comp_stride = T * K;
  float* vo = V + tile * K + 2 * c2;
#pragma unroll
  for (int r = 0; r < 6; ++r) {
    const float2 row[6] = {t[0][r], t[1][r], t[2][r], t[3][r], t[4][r], t[5][r]};
    float2 o[6];
    wg4_bt(row, o);
#pragma unroll
    for (int q = 0; q < 6; ++q) *reinterpret_cast<float2*>(vo + (int64_t)(r * 6 + q) * comp_stride) = o[q];
  }
}

template <int MODE, int NSPLIT = 1>  // 0: y = relu?(Y + bias); 1: y = Y * (x_in > 0) + addend
__global__ void __launch_bounds__(256) winograd_output4_kernel(const float* __restrict__ M, const float* __restrict__ aux0,
                                                               const float* __restrict__ aux1, float* __restrict__ y,
                                                               int B, int H, int W, int N, int TH, int TW, int relu,
                                                               float* __restrict__ ypool, uint32_t* __restrict__ bits) {
  // NSPLIT: M holds the products in NSPLIT K parts, 36 * T * N floats apart (winograd_ksplit), summed here
  // ReLU bit cache (layout as in winograd_input4_kernel): MODE 0 WRITES the mask of its own output (read back by the
  // pooled data gradient of this layer), MODE 1 READS the mask of x_in instead of aux0 (both nullable)
  const int N2 = N >> 1;
  const int64_t T = (int64_t)B * TH * TW;
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= T * N2) return;
  const int c2 = (int)(gid % N2);
  const int64_t tile = gid / N2;
  const int tx = (int)(tile % TW), ty = (int)((tile / TW) % TH), b = (int)(tile / ((int64_t)TW * TH));
  const int64_t comp_stride = T * N;
  const float* mi = M + tile * N + 2 * c2;
  float2 pool[2][2] = {{make_float2(0.f, 0.f), make_float2(0.f, 0.f)}, {make_float2(0.f, 0.f), make_float2(0.f, 0.f)}};
  // data gradient: the 16 addend pairs are requested BEFORE the 36 components (clamped addresses, no branch around a
  // load; without an addend the loads read M, which is at least as large, and are ignored): one round trip, not two
  float2 adq[4][4];
  if (MODE == 1) {
    const float* ap = aux1 ? aux1 : M;
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int yc = min(4 * ty + a, H - 1), xc = min(4 * tx + c, W - 1);
        adq[a][c] = *reinterpret_cast<const float2*>(ap + (((int64_t)b * H + yc) * W + xc) * N + 2 * c2);
      }
  }
  float2 t[6][4];   // t[s][a]: column s after the vertical pass
#pragma unroll
  for (int s = 0; s < 6; ++s) {
    float2 m[6];
#pragma unroll
    for (int r = 0; r < 6; ++r) {
      m[r] = *reinterpret_cast<const float2*>(mi + (int64_t)(r * 6 + s) * comp_stride);
#pragma unroll
      for (int p = 1; p < NSPLIT; ++p) {
        const float2 q = *reinterpret_cast<const float2*>(mi + ((int64_t)p * 36 + r * 6 + s) * comp_stride);
        m[r].x += q.x; m[r].y += q.y;
      }
    }
    wg4_at(m, t[s]);
  }
  float2 bias = make_float2(0.f, 0.f);
  if (MODE == 0 && aux0) bias = *reinterpret_cast<const float2*>(aux0 + 2 * c2);
  uint32_t word = (MODE == 1 && bits) ? bits[gid] : 0u;
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    const int yy = 4 * ty + a;
    if (yy >= H) continue;
    const float2 row[6] = {t[0][a], t[1][a], t[2][a], t[3][a], t[4][a], t[5][a]};
    float2 o[4];
    wg4_at(row, o);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int xx = 4 * tx + c;
      if (xx >= W) continue;
      float2 v = o[c];
      const int64_t idx = (((int64_t)b * H + yy) * W + xx) * N + 2 * c2;
      if (MODE == 0) {
        v.x += bias.x; v.y += bias.y;
        if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); }
        word |= ((v.x > 0.f ? 1u : 0u) | (v.y > 0.f ? 2u : 0u)) << ((a * 4 + c) * 2);
      } else {
        // relu != 0 in this mode: the addend has NOT been through the ReLU mask yet (same mask: both are gradients
        // wrt the output of the layer below), so it is added first and masked with the rest
        const float2 ad = aux1 ? adq[a][c] : make_float2(0.f, 0.f);
        if (relu) { v.x += ad.x; v.y += ad.y; }
        if (bits) {
          const uint32_t wv = word >> ((a * 4 + c) * 2);
          v.x = (wv & 1u) ? v.x : 0.f; v.y = (wv & 2u) ? v.y : 0.f;
        } else if (aux0) {
          const float2 xin = *reinterpret_cast<const float2*>(aux0 + idx);
          v.x = xin.x > 0.f ? v.x : 0.f; v.y = xin.y > 0.f ? v.y : 0.f;
        }
        if (!relu) { v.x += ad.x; v.y += ad.y; }
      }
      if (MODE != 0 || y) *reinterpret_cast<float2*>(y + idx) = v;   // MODE 0: y is optional when only the pool is wanted
      if (MODE == 0) { pool[a >> 1][c >> 1].x += v.x; pool[a >> 1][c >> 1].y += v.y; }
    }
  }
  if (MODE == 0 && bits) bits[gid] = word;
  // slim.avg_pool2d [2,2] VALID of the layer output: a 4x4 output tile holds 2x2 complete pooling windows
  if (MODE == 0 && ypool) {
    const int PH = H >> 1, PW = W >> 1;
#pragma unroll
    for (int pa = 0; pa < 2; ++pa)
#pragma unroll
      for (int pc = 0; pc < 2; ++pc) {
        const int py = 2 * ty + pa, px = 2 * tx + pc;
        if (py < PH && px < PW)
          *reinterpret_cast<float2*>(ypool + (((int64_t)b * PH + py) * PW + px) * N + 2 * c2) =
              make_float2(0.25f * pool[pa][pc].x, 0.25f * pool[pa][pc].y);
      }
  }
}

// ---- the F(4x4) transforms with six waves per (tile, 128 channels) -------------------------------------------------------
// One view per GPU leaves the deep layers with 9-169 tiles: the one-thread-per-(tile, channel pair) kernels above are then a
// fraction of a wave per SIMD, each a chain of 36 loads -> 12 six-point transforms -> 36 stores.  Here a block of 6 waves
// takes one tile x 128 channels: wave s transforms patch column s into LDS, wave r then row r of the result (the scheme of
// winograd5_input7_kernel).  Same sums per value, same bit cache, same pooled output (summed in the same order).
template <int POOLED>   // 0 / 1 as in winograd_input4_kernel (2, the float mask, stays there)
__global__ void __launch_bounds__(384) winograd_input4w_kernel(const float* __restrict__ x, float* __restrict__ V, int B,
                                                               int H, int W, int K, int TH, int TW,
                                                               uint32_t* __restrict__ bits) {
  __shared__ float2 tl[6][6][64];                // [column s][row r][channel-pair lane] after the vertical pass
  __shared__ uint32_t wk[4][64];                 // the mask bits of patch columns 1..4
  const int K2 = K >> 1, kg = K >> 7;
  const int64_t T = (int64_t)B * TH * TW;
  const unsigned per_xcd = gridDim.x / 8;
  const unsigned lb = (blockIdx.x % 8) * per_xcd + blockIdx.x / 8;
  if ((int64_t)lb >= T * kg) return;
  const int64_t tile = lb / kg;
  const int lane = threadIdx.x, w = threadIdx.y, c2 = (int)(lb % kg) * 64 + lane;
  const int tx = (int)(tile % TW), ty = (int)((tile / TW) % TH), b = (int)(tile / ((int64_t)TW * TH));
  const int y0 = 4 * ty - 1, x0 = 4 * tx - 1;
  const int PH = H >> 1, PW = W >> 1;
  const int64_t gid = tile * K2 + c2;
  {
    const int s = w, xx = x0 + s;
    [[maybe_unused]] uint32_t nb[3];             // POOLED 1: the words of the three tiles this patch column reaches into
    const int dx = s == 0 ? 0 : (s == 5 ? 2 : 1), is = s == 0 ? 3 : (s == 5 ? 0 : s - 1);
    if (POOLED == 1) {
      const int nx = min(max(tx + dx - 1, 0), TW - 1);
#pragma unroll
      for (int dy = 0; dy < 3; ++dy) {
        const int ny = min(max(ty + dy - 1, 0), TH - 1);
        nb[dy] = bits[(((int64_t)b * TH + ny) * TW + nx) * K2 + c2];
      }
    }
    float2 d[6];
#pragma unroll
    for (int r = 0; r < 6; ++r) {
      const int yy = y0 + r;
      if (!POOLED) {
        const int yc = min(max(yy, 0), H - 1), xc = min(max(xx, 0), W - 1);
        d[r] = *reinterpret_cast<const float2*>(x + (((int64_t)b * H + yc) * W + xc) * K + 2 * c2);
      } else {
        const int yc = min(max(yy, 0) >> 1, PH - 1), xc = min(max(xx, 0) >> 1, PW - 1);
        d[r] = *reinterpret_cast<const float2*>(x + (((int64_t)b * PH + yc) * PW + xc) * K + 2 * c2);
      }
    }
    uint32_t word = 0u;
#pragma unroll
    for (int r = 0; r < 6; ++r) {
      const int yy = y0 + r;
      float2 v = d[r];
      if (!POOLED) {
        const bool ok = yy >= 0 && yy < H && xx >= 0 && xx < W;
        v = make_float2(ok ? v.x : 0.f, ok ? v.y : 0.f);
        if (r >= 1 && r <= 4) word |= ((v.x > 0.f ? 1u : 0u) | (v.y > 0.f ? 2u : 0u)) << (((r - 1) * 4) * 2);
      } else {
        const bool ok = yy >= 0 && xx >= 0 && (yy >> 1) < PH && (xx >> 1) < PW;
        const int dy = r == 0 ? 0 : (r == 5 ? 2 : 1), ir = r == 0 ? 3 : (r == 5 ? 0 : r - 1);
        const uint32_t wv = nb[dy] >> ((ir * 4 + is) * 2);
        v = make_float2((ok && (wv & 1u)) ? 0.25f * v.x : 0.f, (ok && (wv & 2u)) ? 0.25f * v.y : 0.f);
      }
      d[r] = v;
    }
    if (!POOLED && bits && s >= 1 && s <= 4) wk[s - 1][lane] = word << ((s - 1) * 2);
    float2 t[6];
    wg4_bt(d, t);
#pragma unroll
    for (int r = 0; r < 6; ++r) tl[s][r][lane] = t[r];
  }
  __syncthreads();
  if (!POOLED && bits && w == 5) bits[gid] = wk[0][lane] | wk[1][lane] | wk[2][lane] | wk[3][lane];
  const float2 row[6] = {tl[0][w][lane], tl[1][w][lane], tl[2][w][lane], tl[3][w][lane], tl[4][w][lane], tl[5][w][lane]};
  float2 o[6];
  wg4_bt(row, o);
  const int64_t comp_stride = T * K;
  float* vo = V + tile * K + 2 * c2 + (int64_t)(w * 6) * comp_stride;
#pragma unroll
  for (int q = 0; q < 6; ++q) *reinterpret_cast<float2*>(vo + (int64_t)q * comp_stride) = o[q];
}

template <int MODE, int NSPLIT>
__global__ void __launch_bounds__(384) winograd_output4w_kernel(const float* __restrict__ M, const float* __restrict__ aux0,
                                                                const float* __restrict__ aux1, float* __restrict__ y,
                                                                int B, int H, int W, int N, int TH, int TW, int relu,
                                                                float* __restrict__ ypool, uint32_t* __restrict__ bits) {
  __shared__ float2 tl[6][4][64];                // [column s][output row a][channel-pair lane]
  __shared__ float2 vs[4][4][64];                // MODE 0: the tile's outputs, for the pooled sums
  __shared__ uint32_t wk[4][64];
  const int N2 = N >> 1, ng = N >> 7;
  const int64_t T = (int64_t)B * TH * TW;
  const int64_t tile = blockIdx.x / ng;
  const int lane = threadIdx.x, w = threadIdx.y, c2 = (int)(blockIdx.x % ng) * 64 + lane;
  const int tx = (int)(tile % TW), ty = (int)((tile / TW) % TH), b = (int)(tile / ((int64_t)TW * TH));
  const int64_t comp_stride = T * N, gid = tile * N2 + c2;
  const int a = min(w, 3), yy = 4 * ty + a;
  // the second pass's operands first: row a of the addend and the mask word (clamped addresses, no branch around a load)
  float2 adq[4];
  uint32_t word = 0u;
  if (MODE == 1) {
    const float* ap = aux1 ? aux1 : M;
    const int yc = min(yy, H - 1);
#pragma unroll
    for (int c = 0; c < 4; ++c)
      adq[c] = *reinterpret_cast<const float2*>(ap + (((int64_t)b * H + yc) * W + min(4 * tx + c, W - 1)) * N + 2 * c2);
    if (bits) word = bits[gid];
  }
  {
    const float* mi = M + tile * N + 2 * c2 + (int64_t)w * comp_stride;       // column s = w: components 6 r + s
    float2 m[6];
#pragma unroll
    for (int r = 0; r < 6; ++r) {
      m[r] = *reinterpret_cast<const float2*>(mi + (int64_t)(r * 6) * comp_stride);
#pragma unroll
      for (int p = 1; p < NSPLIT; ++p) {
        const float2 q = *reinterpret_cast<const float2*>(mi + ((int64_t)p * 36 + r * 6) * comp_stride);
        m[r].x += q.x; m[r].y += q.y;
      }
    }
    float2 t[4];
    wg4_at(m, t);
#pragma unroll
    for (int q = 0; q < 4; ++q) tl[w][q][lane] = t[q];
  }
  __syncthreads();
  if (w < 4) {
    uint32_t wout = 0u;
    if (yy < H) {
      const float2 row[6] = {tl[0][a][lane], tl[1][a][lane], tl[2][a][lane], tl[3][a][lane], tl[4][a][lane], tl[5][a][lane]};
      float2 o[4];
      wg4_at(row, o);
      float2 bias = make_float2(0.f, 0.f);
      if (MODE == 0 && aux0) bias = *reinterpret_cast<const float2*>(aux0 + 2 * c2);
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int xx = 4 * tx + c;
        float2 v = o[c];
        if (xx < W) {
          const int64_t idx = (((int64_t)b * H + yy) * W + xx) * N + 2 * c2;
          if (MODE == 0) {
            v.x += bias.x; v.y += bias.y;
            if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); }
            wout |= ((v.x > 0.f ? 1u : 0u) | (v.y > 0.f ? 2u : 0u)) << ((a * 4 + c) * 2);
          } else {
            const float2 ad = aux1 ? adq[c] : make_float2(0.f, 0.f);
            if (relu) { v.x += ad.x; v.y += ad.y; }
            if (bits) {
              const uint32_t wv = word >> ((a * 4 + c) * 2);
              v.x = (wv & 1u) ? v.x : 0.f; v.y = (wv & 2u) ? v.y : 0.f;
            } else if (aux0) {
              const float2 xin = *reinterpret_cast<const float2*>(aux0 + idx);
              v.x = xin.x > 0.f ? v.x : 0.f; v.y = xin.y > 0.f ? v.y : 0.f;
            }
            if (!relu) { v.x += ad.x; v.y += ad.y; }
          }
          if (MODE != 0 || y) *reinterpret_cast<float2*>(y + idx) = v;
        }
        if (MODE == 0) vs[a][c][lane] = v;        // (outside the image: never part of a complete pooling window)
      }
    }
    if (MODE == 0) wk[a][lane] = wout;
  }
  if (MODE != 0) return;
  __syncthreads();
  if (bits && w == 5) bits[gid] = wk[0][lane] | wk[1][lane] | wk[2][lane] | wk[3][lane];
  if (ypool && w < 2) {                           // pooled row w of the tile: rows 2 w, 2 w + 1, summed in the order above
    const int PH = H >> 1, PW = W >> 1, py = 2 * ty + w;
#pragma unroll
    for (int pc = 0; pc < 2; ++pc) {
      const int px = 2 * tx + pc;
      if (py < PH && px < PW) {
        float2 p = make_float2(0.f, 0.f);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float2 v = vs[2 * w + (i >> 1)][2 * pc + (i & 1)][lane];
          p.x += v.x; p.y += v.y;
        }
        *reinterpret_cast<float2*>(ypool + (((int64_t)b * PH + py) * PW + px) * N + 2 * c2) = make_float2(0.25f * p.x, 0.25f * p.y);
      }
    }
  }
}

// ---- batched GEMMs on the f32 MFMA ------------------------------------------------------------------
// block = 4 waves (2 M x 2 N), tile 128 rows x BN columns, K in 32-wide chunks; both operand tiles are
// prefetched into registers one chunk ahead and double-buffered in LDS (36-float padded rows, b128 fragment
// reads, 4 consecutive k per lane feeding 4 MFMA steps -- the scheme of conv3x3_mfma_kernel).
constexpr int WG_KC = 32, WG_LS = 36, WG_XCDS = 8;


template <int BM, int BN, int NBUF>
__global__ void __launch_bounds__(256, 2) winograd_gemm_kernel(WgGemmArgs a) {
  constexpr int MT = BM / 64, NT = BN / 64;           // 32x32 MFMA tiles per wave (waves 2 x 2)
  constexpr int AJ = BM / 32, BJ = BN / 32;           // float4 per thread and chunk for the A / B tile
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As = smem;                                   // [NBUF][BM][36]
  float* Bs = smem + NBUF * BM * WG_LS;               // [NBUF][BN][36]
  const int t = threadIdx.x, lane = t & 63, wid = t >> 6;
  const int wm = wid >> 1, wn = wid & 1, i = lane & 31, h = lane >> 5;
  // XCD-aware order: the dispatcher deals consecutive workgroups round-robin to the 8 XCDs (each with its own
  // 4 MB L2).  Remap so that each XCD walks a CONTIGUOUS range of (batch, n, m) tiles: the ~64 workgroups an
  // XCD runs at a time then share one batch's A and B panels, which fit its L2.
  const int per_xcd = gridDim.x / WG_XCDS;
  const int logical = (blockIdx.x % WG_XCDS) * per_xcd + blockIdx.x / WG_XCDS;
  if (logical >= a.mt * a.nt * a.Z) return;
  const int comp = logical / (a.mt * a.nt);
  const int rem = logical - comp * (a.mt * a.nt);
  const int64_t m0 = (int64_t)(rem % a.mt) * BM;
  const int n0 = (rem / a.mt) * BN;
  const float* Vc = a.V + (int64_t)comp * a.T * a.K;
  const float* Uc = a.U + (int64_t)comp * a.b_batch;
  const int nchunks = a.K / WG_KC;

  // staging: thread t moves float4 #(t&7) of rows (t>>3) + 32 j
  const int q4 = 4 * (t & 7), r0 = t >> 3;
  const float* arow[AJ];
  bool aok[AJ];
#pragma unroll
  for (int j = 0; j < AJ; ++j) {
    const int64_t m = m0 + r0 + 32 * j;
    aok[j] = m < a.T;
    arow[j] = Vc + (aok[j] ? m : 0) * a.K + q4;
  }
  // B rows n0 + (t>>3) + 32 r, float4 #(t&7) of the 32 k of a chunk
  const float* brow = Uc + (int64_t)(n0 + r0) * a.b_row + q4;
  const int64_t brs = (int64_t)32 * a.b_row;
  float4 a0, a1, a2, a3, b0, b1, b2, b3;            // named registers: an indexed array would go to scratch
#define NFS_WG_LOAD(c_)                                                                         \
  {                                                                                             \
    a0 = a1 = a2 = a3 = make_float4(0.f, 0.f, 0.f, 0.f);                                        \
    if (aok[0]) a0 = *reinterpret_cast<const float4*>(arow[0] + (c_) * WG_KC);                  \
    if (aok[1]) a1 = *reinterpret_cast<const float4*>(arow[1] + (c_) * WG_KC);                  \
    if (AJ > 2) {                                                                               \
      if (aok[AJ - 2]) a2 = *reinterpret_cast<const float4*>(arow[AJ - 2] + (c_) * WG_KC);      \
      if (aok[AJ - 1]) a3 = *reinterpret_cast<const float4*>(arow[AJ - 1] + (c_) * WG_KC);      \
    }                                                                                           \
    const float* bn_ = brow + (int64_t)(c_) * a.b_chunk;                                        \
    b0 = *reinterpret_cast<const float4*>(bn_);                                                 \
    b1 = *reinterpret_cast<const float4*>(bn_ + brs);                                           \
    if (BJ > 2) {                                                                               \
      b2 = *reinterpret_cast<const float4*>(bn_ + 2 * brs);                                     \
      b3 = *reinterpret_cast<const float4*>(bn_ + 3 * brs);                                     \
    }                                                                                           \
  }
  NFS_WG_LOAD(0)

  int abase[MT], bbase[NT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) abase[mt] = (wm * (BM / 2) + mt * 32 + i) * WG_LS + 4 * h;
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) bbase[nt] = (wn * (BN / 2) + nt * 32 + i) * WG_LS + 4 * h;

  f32x16 acc[MT][NT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;

#ifdef NFS_ABLATE
  unsigned long long pt[6] = {0, 0, 0, 0, 0, 0};
  unsigned long long tk = a.prof ? clock64() : 0, t_begin = tk;
#define NFS_TICK(i_) if (a.prof) { const unsigned long long n_ = clock64(); pt[i_] += n_ - tk; tk = n_; }
#else
#define NFS_TICK(i_)
#endif
  for (int c = 0; c < nchunks; ++c) {
    float* Ac = As + (NBUF == 2 ? (c & 1) : 0) * BM * WG_LS;
    float* Bc = Bs + (NBUF == 2 ? (c & 1) : 0) * BN * WG_LS;
    if (NBUF == 1 && c > 0) __syncthreads();          // single buffer: everyone is done reading chunk c-1
    if (!NFS_DBG(a, 8)) {
      float* ad = Ac + r0 * WG_LS + q4;
      *reinterpret_cast<float4*>(ad) = a0;
      *reinterpret_cast<float4*>(ad + 32 * WG_LS) = a1;
      if (AJ > 2) {
        *reinterpret_cast<float4*>(ad + 64 * WG_LS) = a2;
        *reinterpret_cast<float4*>(ad + 96 * WG_LS) = a3;
      }
      float* bd = Bc + r0 * WG_LS + q4;
      *reinterpret_cast<float4*>(bd) = b0;
      *reinterpret_cast<float4*>(bd + 32 * WG_LS) = b1;
      if (BJ > 2) {
        *reinterpret_cast<float4*>(bd + 64 * WG_LS) = b2;
        *reinterpret_cast<float4*>(bd + 96 * WG_LS) = b3;
      }
    }
    NFS_TICK(0)                                          // wait for the chunk's loads + LDS staging writes
    if (!NFS_DBG(a, 16)) __syncthreads();    // buffer (c&1) visible; buffer (c+1)&1 was last read in iteration c-1
    NFS_TICK(1)                                          // barrier
    if (c + 1 < nchunks && !NFS_DBG(a, 4)) NFS_WG_LOAD(c + 1)
    NFS_TICK(2)                                          // issue of the next chunk's loads
    if (!NFS_DBG(a, 1))
#pragma unroll
    for (int s = 0; s < WG_KC / 8; ++s) {
      float af[MT][4], bf[NT][4];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const float4 x = *reinterpret_cast<const float4*>(Ac + abase[mt] + 8 * s);
        af[mt][0] = x.x; af[mt][1] = x.y; af[mt][2] = x.z; af[mt][3] = x.w;
      }
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const float4 bq = *reinterpret_cast<const float4*>(Bc + bbase[nt] + 8 * s);
        bf[nt][0] = bq.x; bf[nt][1] = bq.y; bf[nt][2] = bq.z; bf[nt][3] = bq.w;
      }
#pragma unroll
      for (int jj = 0; jj < 4; ++jj)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
          for (int mt = 0; mt < MT; ++mt)
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[mt][jj], bf[nt][jj], acc[mt][nt], 0, 0, 0);
    }
    NFS_TICK(3)                                          // fragment reads + MFMA issue
  }
#undef NFS_WG_LOAD

  // epilogue: transpose the tile through LDS, leave as float4 rows
  constexpr int OS = BN + 4;
  float* otile = smem;
  __syncthreads();
  NFS_TICK(4)                                            // drain of the last MFMAs + barrier
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = wm * (BM / 2) + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) otile[row * OS + wn * (BN / 2) + nt * 32 + i] = acc[mt][nt][r];
    }
  __syncthreads();
  float* Mc = a.M + (int64_t)comp * a.T * a.N;
  const float alpha = a.alpha * (a.alpha_dev ? a.alpha_dev[comp] : 1.f);
  constexpr int Q = BN / 4;
#pragma unroll
  for (int e = 0; e < (BM * Q) / 256; ++e) {
    const int f = t + 256 * e;
    const int row = f / Q, q = f - row * Q;
    const int64_t m = m0 + row;
    if (m >= a.T) continue;
    float4 v = *reinterpret_cast<const float4*>(otile + row * OS + 4 * q);
    v.x *= alpha; v.y *= alpha; v.z *= alpha; v.w *= alpha;
    const int64_t idx = m * a.N + n0 + 4 * q;
    if (a.mask) {
      const float4 mk = *reinterpret_cast<const float4*>(a.mask + (int64_t)comp * a.T * a.N + idx);
      v.x = mk.x > 0.f ? v.x : 0.f; v.y = mk.y > 0.f ? v.y : 0.f;
      v.z = mk.z > 0.f ? v.z : 0.f; v.w = mk.w > 0.f ? v.w : 0.f;
    }
    if (!NFS_DBG(a, 2) || v.x == 12345.f) *reinterpret_cast<float4*>(Mc + idx) = v;
  }
#ifdef NFS_ABLATE
  if (a.prof && lane == 0) {
    NFS_TICK(5)                                          // epilogue: LDS transpose + stores issued
    unsigned long long* o = a.prof + ((size_t)blockIdx.x * 4 + wid) * 8;
    for (int q = 0; q < 6; ++q) o[q] = pt[q];
    o[6] = t_begin;
    o[7] = tk;
  }
#endif
#undef NFS_TICK
}

// ---- the same GEMM with B straight from L2 into registers ("register-B" form) ---------------------------------------
// The filters never change, so they are also kept in the exact order the MFMA wants its B operand: for component z,
// 32-column tile nn and 8-deep k group kk one kilobyte [64 lanes][4] with element (lane, jj) = U_z[8 kk + 4 (lane>>5) + jj]
// [32 nn + (lane & 31)] -- one buffer_load_dwordx4 per lane feeds four k-steps.  Only A goes through LDS (half the
// staging writes, fragment reads and barrier-protected data of winograd_gemm_kernel).  Each B register is reloaded
// with the next chunk's group as soon as its MFMAs are issued: a prefetch distance of one whole chunk (64 MFMAs) with
// 8 float4 of registers.  A is staged with buffer loads too (rows beyond T come back as zeros: no branch around a load);
// the MFMAs are inline asm so that they accumulate in place and stay where they are put.
__device__ __forceinline__ float4 wg_ld4(__amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t soff) {
  return __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}

template <int BM, int BN>
__global__ void __launch_bounds__(256) winograd_gemm_rb_kernel(WgGemmArgs a) {
  constexpr int MT = BM / 64, NT = BN / 64;           // 32x32 MFMA tiles per wave (waves 2 x 2)
  constexpr int AJ = BM / 32;                         // float4 per thread and chunk for the A tile
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As = smem;                                   // [2][BM][36]
  const int t = threadIdx.x, lane = t & 63;
  const int wid = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wm = wid >> 1, wn = wid & 1, i = lane & 31, h = lane >> 5;
  const int per_xcd = gridDim.x / WG_XCDS;
  const int logical = (blockIdx.x % WG_XCDS) * per_xcd + blockIdx.x / WG_XCDS;   // XCD-aware order, see above
  if (logical >= a.mt * a.nt * a.Z) return;
  const int comp = logical / (a.mt * a.nt);
  const int rem = logical - comp * (a.mt * a.nt);
  const int64_t m0 = (int64_t)(rem % a.mt) * BM;
  const int n0 = (rem / a.mt) * BN;
  const int nchunks = a.K / WG_KC;
  const __amdgpu_buffer_rsrc_t a_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(a.V + (int64_t)comp * a.T * a.K), 0, (uint32_t)(a.T * a.K * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t b_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(a.Uq + (int64_t)comp * a.K * a.N), 0, (uint32_t)((int64_t)a.K * a.N * 4), 0x00020000);

  // A staging: thread t moves float4 #(t&7) of rows (t>>3) + 32 j
  const int q4 = 4 * (t & 7), r0 = t >> 3;
  uint32_t ao[AJ];
#pragma unroll
  for (int j = 0; j < AJ; ++j) {
    const int64_t m = m0 + r0 + 32 * j;
    ao[j] = m < a.T ? (uint32_t)((m * a.K + q4) * 4) : 0x80000000u;
  }
  // B fragments: column tile nn = (n0 + wn * BN/2) / 32 + nt, groups of 8 k
  const uint32_t bo = (uint32_t)lane * 16u;
  const uint32_t kgs = (uint32_t)(a.K / 8) * 1024u;   // bytes per column tile
  const uint32_t bt0 = (uint32_t)((n0 + wn * (BN / 2)) / 32) * kgs;
  float4 av[AJ], bq[NT][4];
#pragma unroll
  for (int j = 0; j < AJ; ++j) av[j] = wg_ld4(a_rsrc, ao[j], 0);
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
#pragma unroll
    for (int g = 0; g < 4; ++g) bq[nt][g] = wg_ld4(b_rsrc, bo, bt0 + nt * kgs + g * 1024u);

  int abase[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) abase[mt] = (wm * (BM / 2) + mt * 32 + i) * WG_LS + 4 * h;

  f32x16 acc[MT][NT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;

#pragma unroll 1
  for (int c = 0; c < nchunks; ++c) {
    float* Ac = As + (c & 1) * BM * WG_LS;
    {
      float* ad = Ac + r0 * WG_LS + q4;
#pragma unroll
      for (int j = 0; j < AJ; ++j) *reinterpret_cast<float4*>(ad + 32 * j * WG_LS) = av[j];
    }
    __syncthreads();                                   // buffer (c&1) visible; buffer (c+1)&1 was last read in iteration c-1
    const int cn = c + 1 < nchunks ? c + 1 : c;        // (the last iteration re-fetches its own chunk: no branch)
#pragma unroll
    for (int j = 0; j < AJ; ++j) av[j] = wg_ld4(a_rsrc, ao[j], (uint32_t)cn * (WG_KC * 4));
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      float af[MT][4];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const float4 x = *reinterpret_cast<const float4*>(Ac + abase[mt] + 8 * s);
        af[mt][0] = x.x; af[mt][1] = x.y; af[mt][2] = x.z; af[mt][3] = x.w;
      }
#pragma unroll
      for (int jj = 0; jj < 4; ++jj)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          const float b = jj == 0 ? bq[nt][s].x : jj == 1 ? bq[nt][s].y : jj == 2 ? bq[nt][s].z : bq[nt][s].w;
#pragma unroll
          for (int mt = 0; mt < MT; ++mt)
            asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(acc[mt][nt]) : "v"(af[mt][jj]), "v"(b));
        }
      // this group's registers take the next chunk's group s (one whole chunk of MFMAs ahead of its use)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) bq[nt][s] = wg_ld4(b_rsrc, bo, bt0 + nt * kgs + (uint32_t)(4 * cn + s) * 1024u);
      __builtin_amdgcn_sched_barrier(0);
    }
  }

  // epilogue: transpose the tile through LDS, leave as float4 rows -- one half (the rows of waves wm = 0, then wm = 1)
  // at a time, so that the tile buffer is no larger than the operand buffers and more blocks fit a CU
  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");   // (inline-asm MFMAs: the last one retires before an accumulator is read)
  constexpr int OS = BN + 4, HR = BM / 2;
  float* otile = smem;
  float* Mc = a.M + (int64_t)comp * a.T * a.N;
  constexpr int Q = BN / 4;
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    __syncthreads();
    if (wm == half) {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) otile[row * OS + wn * (BN / 2) + nt * 32 + i] = acc[mt][nt][r];
        }
    }
    __syncthreads();
#pragma unroll
    for (int e = 0; e < (HR * Q) / 256; ++e) {
      const int f = t + 256 * e;
      const int row = f / Q, q = f - row * Q;
      const int64_t m = m0 + half * HR + row;
      if (m >= a.T) continue;
      *reinterpret_cast<float4*>(Mc + m * a.N + n0 + 4 * q) = *reinterpret_cast<const float4*>(otile + row * OS + 4 * q);
    }
  }
}

// ---- register-B form on 16-row MFMA tiles ("rb16") --------------------------------------------------------------------
// v_mfma_f32_16x16x4_f32 runs at the same 64 flop / cycle / SIMD as the 32x32x2 form, and lets the row tile be a multiple
// of 16: 392 Winograd tiles (conv4_x, 8 views) are 5 x 80 rows instead of 7 x 64 = 448, 72 (conv5_1) are 80 instead of
// 128.  The four waves of a block sit side by side along N (NW16 column tiles of 16 each) and all read the block's BM =
// 16 MT16 rows of A from LDS; B as in winograd_gemm_rb_kernel, packed for this instruction: element (lane, s) of the
// kilobyte for (z, 16-column tile nn, 16-deep k group kk) is U_z[16 kk + 4 (lane>>4) + s][16 nn + (lane & 15)], i.e. MFMA
// step s multiplies k = s, 4 + s, 8 + s, 12 + s of the group -- the lane's A operand for the four steps is then one
// contiguous float4 of its LDS row.
typedef float f32x4 __attribute__((ext_vector_type(4)));

// One block's work on MT16 live row tiles starting at row m0 (a block of a taller grid tile whose last rows lie beyond T
// runs the instance for the row tiles that exist: no MFMA, LDS read or staging load is spent on rows of padding).
template <int MT16, int NW16>
__device__ __forceinline__ void winograd_gemm_rb16_block(const WgGemmArgs& a, float* smem, int comp, int64_t m0, int n0,
                                                         int split = 0) {
  constexpr int BM = 16 * MT16, BN = 64 * NW16;
  constexpr int BMP = (BM + 31) / 32 * 32;            // staged rows (a multiple of the 32 rows one pass of the block moves)
  constexpr int AJ = BMP / 32;
  float* As = smem;                                   // [2][BMP][36]
  const int t = threadIdx.x, lane = t & 63;
  const int wid = __builtin_amdgcn_readfirstlane(t >> 6);
  const int nchunks = a.K / WG_KC / a.ksplit, c0 = split * nchunks;     // this block's share of the K chunks
  const __amdgpu_buffer_rsrc_t a_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(a.V + (int64_t)comp * a.T * a.K), 0, (uint32_t)(a.T * a.K * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t b_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(a.Uq16 + (int64_t)comp * a.K * a.N), 0, (uint32_t)((int64_t)a.K * a.N * 4), 0x00020000);

  // A staging: thread t moves float4 #(t&7) of rows (t>>3) + 32 j; rows beyond the tile or beyond T read as zeros
  const int q4 = 4 * (t & 7), r0 = t >> 3;
  uint32_t ao[AJ];
#pragma unroll
  for (int j = 0; j < AJ; ++j) {
    const int r = r0 + 32 * j;
    const int64_t m = m0 + r;
    ao[j] = (r < BM && m < a.T) ? (uint32_t)((m * a.K + q4) * 4) : 0x80000000u;
  }
  // B fragment of (16-column tile nn, 16-deep k group kk): packed filters -> one kilobyte, lane-linear.  A symmetric
  // matrix needs no packing at all: B[k][n] = D[n][k], so the lane's four k (16 kk + 4 (lane>>4) + s) are four
  // consecutive floats of ROW 16 nn + (lane & 15) of the plain matrix.
  const uint32_t bo = a.symb ? (uint32_t)((lane & 15) * a.N + 4 * (lane >> 4)) * 4u : (uint32_t)lane * 16u;
  const uint32_t kgs = a.symb ? (uint32_t)(16 * a.N) * 4u : (uint32_t)(a.K / 16) * 1024u;   // bytes per 16-column tile
  const uint32_t gst = a.symb ? 64u : 1024u;                                               // ... per 16-deep k group
  const uint32_t bt0 = (uint32_t)((n0 + wid * 16 * NW16) / 16) * kgs;
  float4 av[AJ], bq[NW16][2];
#pragma unroll
  for (int j = 0; j < AJ; ++j) av[j] = wg_ld4(a_rsrc, ao[j], (uint32_t)c0 * (WG_KC * 4));
#pragma unroll
  for (int nt = 0; nt < NW16; ++nt)
#pragma unroll
    for (int g = 0; g < 2; ++g) bq[nt][g] = wg_ld4(b_rsrc, bo, bt0 + nt * kgs + (uint32_t)(2 * c0 + g) * gst);

  const int afrag = (lane & 15) * WG_LS + 4 * (lane >> 4);     // + 16 mt rows, + 16 g floats

  f32x4 acc[MT16][NW16];
#pragma unroll
  for (int mt = 0; mt < MT16; ++mt)
#pragma unroll
    for (int nt = 0; nt < NW16; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};

#pragma unroll 1
  for (int c = 0; c < nchunks; ++c) {
    float* Ac = As + (c & 1) * BMP * WG_LS;
    {
      float* ad = Ac + r0 * WG_LS + q4;
#pragma unroll
      for (int j = 0; j < AJ; ++j) *reinterpret_cast<float4*>(ad + 32 * j * WG_LS) = av[j];
    }
    __syncthreads();                                   // buffer (c&1) visible; buffer (c+1)&1 was last read in iteration c-1
    const int cn = c0 + (c + 1 < nchunks ? c + 1 : c); // (the last iteration re-fetches its own chunk: no branch)
#pragma unroll
    for (int j = 0; j < AJ; ++j) av[j] = wg_ld4(a_rsrc, ao[j], (uint32_t)cn * (WG_KC * 4));
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      float4 af[MT16];
#pragma unroll
      for (int mt = 0; mt < MT16; ++mt) af[mt] = *reinterpret_cast<const float4*>(Ac + afrag + 16 * mt * WG_LS + 16 * g);
#pragma unroll
      for (int ss = 0; ss < 4; ++ss)
#pragma unroll
        for (int nt = 0; nt < NW16; ++nt) {
          const float b = ss == 0 ? bq[nt][g].x : ss == 1 ? bq[nt][g].y : ss == 2 ? bq[nt][g].z : bq[nt][g].w;
#pragma unroll
          for (int mt = 0; mt < MT16; ++mt) {
            const float av_ = ss == 0 ? af[mt].x : ss == 1 ? af[mt].y : ss == 2 ? af[mt].z : af[mt].w;
            asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[mt][nt]) : "v"(av_), "v"(b));
          }
        }
#pragma unroll
      for (int nt = 0; nt < NW16; ++nt) bq[nt][g] = wg_ld4(b_rsrc, bo, bt0 + nt * kgs + (uint32_t)(2 * cn + g) * gst);
      __builtin_amdgcn_sched_barrier(0);
    }
  }

  // epilogue: the tile through LDS (C layout: lane = column l & 15, rows 4 (l >> 4) + r), out as float4 rows; at most
  // five row tiles (80 rows) per pass, so that a tall tile does not need a tall buffer
  // (256-column tiles, NW16 = 4: two row tiles per pass keep the buffer at 33 KB, so that blocks still share a CU)
  constexpr int OS = BN + 4, EPMAX = NW16 >= 4 ? 2 : 5, EP = MT16 < EPMAX ? MT16 : EPMAX, NPASS = (MT16 + EP - 1) / EP;
  // The inline-asm MFMAs are opaque to the compiler's hazard recogniser: nothing may read an accumulator until the last
  // MFMA has retired (8 passes).  The barrier below used to be reached behind the loop's outstanding loads; a kernel
  // with nothing outstanding (a round-5 experiment with one row tile) read stale sums.
  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
  float* otile = smem;
  float* Mc = a.M + ((int64_t)split * a.Z + comp) * a.T * a.N;
  const float alpha = a.alpha * (a.alpha_dev ? a.alpha_dev[comp] : 1.f);
  constexpr int Q = BN / 4;
#pragma unroll
  for (int pass = 0; pass < NPASS; ++pass) {
    __syncthreads();
#pragma unroll
    for (int mt = 0; mt < MT16; ++mt)
      if (mt / EP == pass)
#pragma unroll
        for (int nt = 0; nt < NW16; ++nt)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            otile[(16 * (mt % EP) + 4 * (lane >> 4) + r) * OS + wid * 16 * NW16 + 16 * nt + (lane & 15)] = acc[mt][nt][r];
    __syncthreads();
    const int rows = 16 * ((pass + 1) * EP <= MT16 ? EP : MT16 - pass * EP);
    for (int f = t; f < rows * Q; f += 256) {
      const int row = f / Q, q = f - row * Q;
      const int64_t m = m0 + 16 * EP * pass + row;
      if (m >= a.T) continue;
      float4 v = *reinterpret_cast<const float4*>(otile + row * OS + 4 * q);
      v.x *= alpha; v.y *= alpha; v.z *= alpha; v.w *= alpha;
      const int64_t idx = m * a.N + n0 + 4 * q;
      if (a.mask) {
        const float4 mk = *reinterpret_cast<const float4*>(a.mask + (int64_t)comp * a.T * a.N + idx);
        v.x = mk.x > 0.f ? v.x : 0.f; v.y = mk.y > 0.f ? v.y : 0.f;
        v.z = mk.z > 0.f ? v.z : 0.f; v.w = mk.w > 0.f ? v.w : 0.f;
      }
      *reinterpret_cast<float4*>(Mc + idx) = v;
    }
  }
}

// The grid tiles the rows in blocks of 16 MT16; the last block of a component runs the instance for the row tiles it
// really has (up to five: the taller tiles keep their zero-padded rows, their share of padding is small)
template <int MT16, int NW16>
__device__ __forceinline__ void winograd_gemm_rb16_tile(const WgGemmArgs& a, float* smem, int block, int nblocks) {
  const int per_xcd = nblocks / WG_XCDS;
  const int logical = (block % WG_XCDS) * per_xcd + block / WG_XCDS;   // XCD-aware order, see above
  const int per_split = a.mt * a.nt * a.Z;
  if (logical >= per_split * a.ksplit) return;
  const int split = logical / per_split;                 // (the K parts of a tile sit a whole grid apart: different CUs)
  const int lg = logical - split * per_split;
  const int comp = lg / (a.mt * a.nt);
  const int rem = lg - comp * (a.mt * a.nt);
  const int64_t m0 = (int64_t)(rem % a.mt) * (16 * MT16);
  const int n0 = (rem / a.mt) * (64 * NW16);
  if constexpr (MT16 <= 5) {
    const int64_t left = (a.T - m0 + 15) / 16;
    const int live = left < MT16 ? (int)left : MT16;
    if (live == MT16) winograd_gemm_rb16_block<MT16, NW16>(a, smem, comp, m0, n0, split);
    else if (live == 1) winograd_gemm_rb16_block<1, NW16>(a, smem, comp, m0, n0, split);
    else if (MT16 > 2 && live == 2) winograd_gemm_rb16_block<(MT16 > 2 ? 2 : 1), NW16>(a, smem, comp, m0, n0, split);
    else if (MT16 > 3 && live == 3) winograd_gemm_rb16_block<(MT16 > 3 ? 3 : 1), NW16>(a, smem, comp, m0, n0, split);
    else if (MT16 > 4 && live == 4) winograd_gemm_rb16_block<(MT16 > 4 ? 4 : 1), NW16>(a, smem, comp, m0, n0, split);
  } else {
    winograd_gemm_rb16_block<MT16, NW16>(a, smem, comp, m0, n0, split);
  }
}

template <int MT16, int NW16>
__global__ void __launch_bounds__(256) winograd_gemm_rb16_kernel(WgGemmArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  winograd_gemm_rb16_tile<MT16, NW16>(a, smem, (int)blockIdx.x, (int)gridDim.x);
}

// Several batched GEMMs in one launch (the Gram gradients of all style layers): block -> (problem, its block within the
// problem); ustart are multiples of 8, so the XCD deal restarts with every problem.
constexpr int WG_MAXG = 8;
struct WgGroupArgs {
  WgGemmArgs g[WG_MAXG];
  int ustart[WG_MAXG + 1];
  int n;
};
template <int MT16, int NW16>
__global__ void __launch_bounds__(256) winograd_gemm_rb16_group_kernel(WgGroupArgs G) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  int p = 0;
  while (p + 1 < G.n && (int)blockIdx.x >= G.ustart[p + 1]) ++p;
  winograd_gemm_rb16_tile<MT16, NW16>(G.g[p], smem, (int)blockIdx.x - G.ustart[p], G.ustart[p + 1] - G.ustart[p]);
}

// ---- the 16-row register-B form in split-limb arithmetic ("rb16s") ------------------------------------------------------
// The same tiling as winograd_gemm_rb16_kernel on v_mfma_f32_16x16x32_bf16: every float32 operand is written exactly as
// three bf16 limbs (round-to-nearest at each level, see the split-limb notes further down) and the six leading limb
// products of a 16 x 16 x 32 block take 6 x 16 cycles where the f32-input MFMA needs 8 x 32 (2.67 x).  Where the limbs
// are made is what the round-2 kernel (winograd_gemm_split_kernel, both operands through LDS, filters as limb planes
// from memory: 6 bytes per element) got wrong for these shapes:
//   * B stays the float32 fragment pack of the rb16 kernel (4 bytes per element from HBM -- the deep layers stream
//     51 MB of filters per launch and are within 2 x of that bound at the bf16 rate) and is split IN REGISTERS by the
//     wave that loaded it: 8 NW16 values per lane and chunk, amortised over the wave's MT16 row tiles;
//   * A is split ONCE per block while it is staged (each thread splits the float4 it moves) into three LDS planes laid
//     out in fragment order, so that a lane's 8 k values of a limb are one ds_read_b128.
// The k index a lane position stands for is the rb16 pack's: position j of lane group q is k = 16 (j / 4) + 4 q + j % 4
// of the 32-deep chunk, for A and B alike (the MFMA pairs positions, whatever k they are called).
typedef __bf16 bf16x8s __attribute__((ext_vector_type(8)));
#ifndef NFS_RB16S_PF
#define NFS_RB16S_PF 2
#endif
// timing-only ablations of the rb16s kernel (tools/variant_sweep.sh -DNFS_RB16S_ABL=<mask>; WRONG RESULTS by construction,
// never in the product build): 1 no MFMAs, 2 no limb split of B, 4 no limb split / LDS staging of A, 8 no operand loads
// after the prologue, 16 no epilogue stores
#ifndef NFS_RB16S_ABL
#define NFS_RB16S_ABL 0
#endif
constexpr int WS_RB = 80;            // bytes per LDS row of one limb plane: 32 bf16 + 16 pad (conflict-free b128 reads)

typedef float rb16s_f2 __attribute__((ext_vector_type(2)));
typedef __bf16 rb16s_b2 __attribute__((ext_vector_type(2)));
// two floats -> their hi / mid / lo limbs, each pair packed in one dword (one v_cvt_pk_bf16_f32 per level)
__device__ __forceinline__ void rb16s_split2(float x0, float x1, unsigned& h, unsigned& m, unsigned& l) {
  const rb16s_f2 x = {x0, x1};
  const rb16s_b2 bh = __builtin_convertvector(x, rb16s_b2);
  const rb16s_f2 r1 = x - __builtin_convertvector(bh, rb16s_f2);
  const rb16s_b2 bm = __builtin_convertvector(r1, rb16s_b2);
  const rb16s_f2 r2 = r1 - __builtin_convertvector(bm, rb16s_f2);
  const rb16s_b2 bl = __builtin_convertvector(r2, rb16s_b2);
  h = __builtin_bit_cast(unsigned, bh);
  m = __builtin_bit_cast(unsigned, bm);
  l = __builtin_bit_cast(unsigned, bl);
}
// 8 floats (two float4 of the fragment pack) -> the three limb fragments
__device__ __forceinline__ void rb16s_split8(const float4& a, const float4& b, bf16x8s& H, bf16x8s& M, bf16x8s& L) {
  uint4 hv, mv, lv;
  rb16s_split2(a.x, a.y, hv.x, mv.x, lv.x); rb16s_split2(a.z, a.w, hv.y, mv.y, lv.y);
  rb16s_split2(b.x, b.y, hv.z, mv.z, lv.z); rb16s_split2(b.z, b.w, hv.w, mv.w, lv.w);
  H = __builtin_bit_cast(bf16x8s, hv); M = __builtin_bit_cast(bf16x8s, mv); L = __builtin_bit_cast(bf16x8s, lv);
}

// B once and for all: the fragment pack split at PACK time into three limb planes, [Z][N/16][K/32][3][64 lanes][8 bf16]
// (6 bytes per filter value instead of 4).  Splitting B in registers costs every block ~44 VALU instructions per chunk
// and 64 columns, repeated by each of the T / BM row blocks that share the columns (10 times at conv3_x and 8 views):
// 4-5 us of a 40-us launch (profiles/r05_rb16s_ablation.txt, mask 2).  NFS_RB16S_PRE=0 builds the in-register form.
#ifndef NFS_RB16S_PRE
#define NFS_RB16S_PRE 1
#endif
__global__ void __launch_bounds__(256) winograd_pack_limbs16_kernel(const float4* __restrict__ uq16, uint4* __restrict__ ub,
                                                                    int64_t total) {
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;     // (z, n tile, 32-deep chunk) x lane
  if (gid >= total) return;
  const int64_t grp = gid >> 6;
  const int lane = (int)(gid & 63);
  const float4 a = uq16[(2 * grp) * 64 + lane], b = uq16[(2 * grp + 1) * 64 + lane];
  bf16x8s H, M, L;
  rb16s_split8(a, b, H, M, L);
  ub[(3 * grp) * 64 + lane] = __builtin_bit_cast(uint4, H);
  ub[(3 * grp + 1) * 64 + lane] = __builtin_bit_cast(uint4, M);
  ub[(3 * grp + 2) * 64 + lane] = __builtin_bit_cast(uint4, L);
}

template <int MT16, int NW16, bool PRE>
__device__ __forceinline__ void winograd_gemm_rb16s_block(const WgGemmArgs& a, unsigned char* smem, int comp, int64_t m0,
                                                          int n0, int split) {
  constexpr int BM = 16 * MT16, BN = 64 * NW16;
  constexpr int BMP = (BM + 31) / 32 * 32, AJ = BMP / 32;
  constexpr int PLANE = BMP * WS_RB, BUF = 3 * PLANE;          // bytes: [2][3 limbs][BMP rows][80]
  constexpr int G = MT16 <= 5 ? MT16 : (MT16 == 7 ? 4 : 5);   // row tiles whose A fragments are live together
  const int t = threadIdx.x, lane = t & 63;
  const int wid = __builtin_amdgcn_readfirstlane(t >> 6);
  const int nchunks = a.K / WG_KC / a.ksplit, c0 = split * nchunks;
  const __amdgpu_buffer_rsrc_t a_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(a.V + (int64_t)comp * a.T * a.K), 0, (uint32_t)(a.T * a.K * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t b_rsrc =
      PRE ? __builtin_amdgcn_make_buffer_rsrc(
                const_cast<unsigned char*>(static_cast<const unsigned char*>(a.Ub16)) + (int64_t)comp * a.K * a.N * 6, 0,
                (uint32_t)((int64_t)a.K * a.N * 6), 0x00020000)
          : __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.Uq16 + (int64_t)comp * a.K * a.N), 0,
                                              (uint32_t)((int64_t)a.K * a.N * 4), 0x00020000);
  // A staging: thread t moves float4 #(t & 7) (k = 4 (t & 7) ...) of rows (t >> 3) + 32 j; in fragment order those four
  // values are positions 8 q + 4 half .. + 3 with q = (t & 3), half = (t >> 2) & 1: 8 bytes at 16 q + 8 half
  const int q4 = 4 * (t & 7), r0 = t >> 3;
  const int a_st = r0 * WS_RB + 16 * (t & 3) + 8 * ((t >> 2) & 1);
  uint32_t ao[AJ];
#pragma unroll
  for (int j = 0; j < AJ; ++j) {
    const int r = r0 + 32 * j;
    const int64_t m = m0 + r;
    ao[j] = (r < BM && m < a.T) ? (uint32_t)((m * a.K + q4) * 4) : 0x80000000u;
  }
  const uint32_t bo = (uint32_t)lane * 16u;
  constexpr int NB = PRE ? 3 : 2;                                 // 16-byte loads per lane, chunk and 16 columns
  const uint32_t kgs = (uint32_t)(a.K / 16) * (PRE ? 1536u : 1024u);
  const uint32_t bt0 = (uint32_t)((n0 + wid * 16 * NW16) / 16) * kgs;
  // Two register sets, two chunks in flight per wave: at the split-limb rate a chunk's MFMAs take ~0.4 us, a round trip
  // to HBM 1-2 us, and the accumulators leave room for two blocks per CU only -- with one chunk ahead (the rb16 kernel's
  // distance) every chunk waited for its operands (measured: 40 us where the MFMAs need 12)
  constexpr int PF = NFS_RB16S_PF;                // register sets = chunks in flight per wave
  static_assert(PF >= 2 && PF <= 4, "the K loop below is written for 2-4 register sets");
  float4 av[PF][AJ], bq[PF][NW16][NB];
#pragma unroll
  for (int st = 0; st < PF; ++st) {
    const int cc = c0 + (st < nchunks ? st : nchunks - 1);
#pragma unroll
    for (int j = 0; j < AJ; ++j) av[st][j] = wg_ld4(a_rsrc, ao[j], (uint32_t)cc * (WG_KC * 4));
#pragma unroll
    for (int nt = 0; nt < NW16; ++nt)
#pragma unroll
      for (int g = 0; g < NB; ++g) bq[st][nt][g] = wg_ld4(b_rsrc, bo, bt0 + nt * kgs + (uint32_t)(NB * cc + g) * 1024u);
  }

  const int afrag = (lane & 15) * WS_RB + 16 * (lane >> 4);     // + 16 mt rows, + plane
  f32x4 acc[MT16][NW16];
#pragma unroll
  for (int mt = 0; mt < MT16; ++mt)
#pragma unroll
    for (int nt = 0; nt < NW16; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};

#define NFS_RB16S_STEP(ST, C)                                                                                      \
  {                                                                                                                \
    unsigned char* Ac = smem + ((C) & 1) * BUF;                                                                    \
    if (!(NFS_RB16S_ABL & 4)) _Pragma("unroll") for (int j = 0; j < AJ; ++j) {                                     \
      uint2 hv, mv, lv;                                                                                            \
      if (NFS_RB16S_ABL & 32) {                                                                                    \
        hv = make_uint2(__float_as_uint(av[ST][j].x), __float_as_uint(av[ST][j].y));                               \
        mv = make_uint2(__float_as_uint(av[ST][j].z), __float_as_uint(av[ST][j].w)); lv = hv;                      \
      } else {                                                                                                     \
        rb16s_split2(av[ST][j].x, av[ST][j].y, hv.x, mv.x, lv.x);                                                  \
        rb16s_split2(av[ST][j].z, av[ST][j].w, hv.y, mv.y, lv.y);                                                  \
      }                                                                                                            \
      unsigned char* d_ = Ac + a_st + 32 * j * WS_RB;                                                              \
      *reinterpret_cast<uint2*>(d_) = hv;                                                                          \
      *reinterpret_cast<uint2*>(d_ + PLANE) = mv;                                                                  \
      *reinterpret_cast<uint2*>(d_ + 2 * PLANE) = lv;                                                              \
    }                                                                                                              \
    __syncthreads();       /* buffer (C & 1) visible; the other one was last read in the step before */            \
    const int cn = c0 + ((C) + PF < nchunks ? (C) + PF : nchunks - 1);   /* (past the end: a harmless re-fetch) */  \
    if (!(NFS_RB16S_ABL & 8))                                                                                      \
      _Pragma("unroll") for (int j = 0; j < AJ; ++j) av[ST][j] = wg_ld4(a_rsrc, ao[j], (uint32_t)cn * (WG_KC * 4)); \
    bf16x8s bf[NW16][3];                                                                                           \
    _Pragma("unroll") for (int nt = 0; nt < NW16; ++nt) {                                                          \
      if (NFS_RB16S_ABL & 2) {                                                                                     \
        bf[nt][0] = __builtin_bit_cast(bf16x8s, bq[ST][nt][0]); bf[nt][1] = __builtin_bit_cast(bf16x8s, bq[ST][nt][1]); \
        bf[nt][2] = bf[nt][0];                                                                                     \
      } else if (PRE) {                                                                                            \
        _Pragma("unroll") for (int p = 0; p < 3; ++p) bf[nt][p] = __builtin_bit_cast(bf16x8s, bq[ST][nt][p % NB]);  \
      } else rb16s_split8(bq[ST][nt][0], bq[ST][nt][1], bf[nt][0], bf[nt][1], bf[nt][2]);                          \
    }                                                                                                              \
    if (!(NFS_RB16S_ABL & 8))                                                                                      \
      _Pragma("unroll") for (int nt = 0; nt < NW16; ++nt)                                                          \
        _Pragma("unroll") for (int g = 0; g < NB; ++g)                                                             \
          bq[ST][nt][g] = wg_ld4(b_rsrc, bo, bt0 + nt * kgs + (uint32_t)(NB * cn + g) * 1024u);                    \
    _Pragma("unroll") for (int mg = 0; mg < MT16; mg += G) {                                                       \
      bf16x8s af[G][3];                                                                                            \
      _Pragma("unroll") for (int gi = 0; gi < G; ++gi)                                                             \
        if (mg + gi < MT16)                                                                                        \
          _Pragma("unroll") for (int p = 0; p < 3; ++p)                                                            \
            af[gi][p] = *reinterpret_cast<const bf16x8s*>(Ac + p * PLANE + afrag + 16 * (mg + gi) * WS_RB);        \
      /* limb product outermost (smallest terms first: hi lo, lo hi, mid mid | hi mid, mid hi | hi hi), tiles      \
         inner: consecutive MFMAs go to different accumulators */                                                  \
      _Pragma("unroll") for (int lp = 0; lp < 6; ++lp) {                                                           \
        const int pa = lp == 0 ? 0 : lp == 1 ? 2 : lp == 2 ? 1 : lp == 3 ? 0 : lp == 4 ? 1 : 0;                    \
        const int pb = lp == 0 ? 2 : lp == 1 ? 0 : lp == 2 ? 1 : lp == 3 ? 1 : lp == 4 ? 0 : 0;                    \
        _Pragma("unroll") for (int nt = 0; nt < NW16; ++nt)                                                        \
          _Pragma("unroll") for (int gi = 0; gi < G; ++gi)                                                         \
            if (mg + gi < MT16 && !(NFS_RB16S_ABL & 1))                                                            \
              acc[mg + gi][nt] =                                                                                   \
                  __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[gi][pa], bf[nt][pb], acc[mg + gi][nt], 0, 0, 0);      \
      }                                                                                                            \
    }                                                                                                              \
  }
#pragma unroll 1
  for (int c = 0; c < nchunks; c += PF) {
    NFS_RB16S_STEP(0, c)
    if (c + 1 < nchunks) NFS_RB16S_STEP(1, c + 1)
    if (PF > 2 && c + 2 < nchunks) NFS_RB16S_STEP(PF > 2 ? 2 : 0, c + 2)
    if (PF > 3 && c + 3 < nchunks) NFS_RB16S_STEP(PF > 3 ? 3 : 0, c + 3)
  }
#undef NFS_RB16S_STEP

  // epilogue: as winograd_gemm_rb16_block (same C layout)
  constexpr int OS = BN + 4, EPMAX = NW16 >= 4 ? 2 : 5, EP = MT16 < EPMAX ? MT16 : EPMAX, NPASS = (MT16 + EP - 1) / EP;
  float* otile = reinterpret_cast<float*>(smem);
  float* Mc = a.M + ((int64_t)split * a.Z + comp) * a.T * a.N;
  constexpr int Q = BN / 4;
#pragma unroll
  for (int pass = 0; pass < NPASS; ++pass) {
    __syncthreads();
#pragma unroll
    for (int mt = 0; mt < MT16; ++mt)
      if (mt / EP == pass)
#pragma unroll
        for (int nt = 0; nt < NW16; ++nt)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            otile[(16 * (mt % EP) + 4 * (lane >> 4) + r) * OS + wid * 16 * NW16 + 16 * nt + (lane & 15)] = acc[mt][nt][r];
    __syncthreads();
    const int rows = 16 * ((pass + 1) * EP <= MT16 ? EP : MT16 - pass * EP);
    for (int f = t; f < rows * Q; f += 256) {
      const int row = f / Q, q = f - row * Q;
      const int64_t m = m0 + 16 * EP * pass + row;
      if (m >= a.T) continue;
      const float4 ov = *reinterpret_cast<const float4*>(otile + row * OS + 4 * q);
      if (!(NFS_RB16S_ABL & 16) || ov.x == 12345.678f) *reinterpret_cast<float4*>(Mc + m * a.N + n0 + 4 * q) = ov;
    }
  }
}

// Blocks per CU the register allocation aims at for the instances of up to NFS_RB16S_OCC_TILES accumulator tiles per wave: with 3 the
// 80 x 128 tile drops from 196 to 144 VGPRs without a spill (48 x 128: 144 -> 112) and a third block sits in another phase
// while one is in its MFMAs -- the VALU / load / barrier phases of a chunk hide under a neighbour's matrix work instead of
// queueing behind the block's own (fixed tiles: 515.5 -> 488.1 us for the 14 launches of tools/split_gemm_bench.py; the
// 8-view step 2.487 -> 2.450 ms; 4 spills the 80 x 128 instance: 763 us)
#ifndef NFS_RB16S_OCC
#define NFS_RB16S_OCC 3
#endif
#ifndef NFS_RB16S_OCC_TILES
#define NFS_RB16S_OCC_TILES 14        // (10 -> 14, i.e. the 112 x 128 tile too: 486 -> 482 us)
#endif
template <int MT16, int NW16, bool PRE>
__global__ void __launch_bounds__(256, (MT16 * NW16 <= NFS_RB16S_OCC_TILES ? NFS_RB16S_OCC : 1)) winograd_gemm_rb16s_kernel(WgGemmArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_s[];
  const int nblocks = (int)gridDim.x, block = (int)blockIdx.x;
  const int per_xcd = nblocks / WG_XCDS;
  const int logical = (block % WG_XCDS) * per_xcd + block / WG_XCDS;   // XCD-aware order, as the rb16 kernel
  const int per_split = a.mt * a.nt * a.Z;
  if (logical >= per_split * a.ksplit) return;
  const int split = logical / per_split;
  const int lg = logical - split * per_split;
  const int comp = lg / (a.mt * a.nt);
  const int rem = lg - comp * (a.mt * a.nt);
  const int64_t m0 = (int64_t)(rem % a.mt) * (16 * MT16);
  const int n0 = (rem / a.mt) * (64 * NW16);
  if constexpr (MT16 <= 5) {
    const int64_t left = (a.T - m0 + 15) / 16;
    const int live = left < MT16 ? (int)left : MT16;
    if (live == MT16) winograd_gemm_rb16s_block<MT16, NW16, PRE>(a, smem_s, comp, m0, n0, split);
    else if (live == 1) winograd_gemm_rb16s_block<1, NW16, PRE>(a, smem_s, comp, m0, n0, split);
    else if (MT16 > 2 && live == 2) winograd_gemm_rb16s_block<(MT16 > 2 ? 2 : 1), NW16, PRE>(a, smem_s, comp, m0, n0, split);
    else if (MT16 > 3 && live == 3) winograd_gemm_rb16s_block<(MT16 > 3 ? 3 : 1), NW16, PRE>(a, smem_s, comp, m0, n0, split);
    else if (MT16 > 4 && live == 4) winograd_gemm_rb16s_block<(MT16 > 4 ? 4 : 1), NW16, PRE>(a, smem_s, comp, m0, n0, split);
  } else {
    winograd_gemm_rb16s_block<MT16, NW16, PRE>(a, smem_s, comp, m0, n0, split);
  }
}

// Round 6, built / measured / removed (profiles/r06_rb16d_ab.txt, the kernel text in profiles/r06_rb16d_kernel.hip.txt, commit
// a7c1c05): A as three bf16 limb planes in fragment order moved global -> LDS by `buffer_load_dwordx4 ... lds` (no VALU split,
// no ds_write, three LDS buffers, XOR-swizzled 64-byte rows).  Bit-identical to rb16s; the GEMM alone 1.4-4.9 us faster per
// launch (conv3_x 41.6 -> 37.5, conv4_2 39.6 -> 37.9, conv5_1 +0.9), ~40 us per step -- against ~200 MB more V per step
// (6 bytes per element written by 17 input transforms that run at the bandwidth already: ~50 us).  hipcc also waits for
// EVERY transfer in flight before a fragment read that might alias it (it cannot tell the three buffers apart), so the
// transfer of chunk c + 2 never overlaps the multiplies of chunk c; getting that needs the loads of the K loop in inline
// asm with hand-counted waits on a fully unrolled loop.  Priced at 2-3 us more per launch: still not above the transforms'
// extra bytes.
// U [Z][K/32][N][32] -> Uq16 [Z][N/16][K/16][64][4]
__global__ void __launch_bounds__(256) winograd_pack_frag16_kernel(const float* __restrict__ up, float* __restrict__ uq,
                                                                   int K, int N, int64_t total) {
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= total) return;
  const int n = (int)(gid % N), k = (int)((gid / N) % K), z = (int)(gid / ((int64_t)N * K));
  const float u = up[(((int64_t)z * (K / 32) + k / 32) * N + n) * 32 + (k & 31)];
  const int kk = k >> 4, kq = (k >> 2) & 3, ss = k & 3;
  uq[((((int64_t)z * (N / 16) + n / 16) * (K / 16) + kk) * 64 + kq * 16 + (n & 15)) * 4 + ss] = u;
}

// U [Z][K/32][N][32] -> Uq [Z][N/32][K/8][64][4]
__global__ void __launch_bounds__(256) winograd_pack_frag_kernel(const float* __restrict__ up, float* __restrict__ uq,
                                                                 int K, int N, int64_t total) {
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= total) return;
  const int n = (int)(gid % N), k = (int)((gid / N) % K), z = (int)(gid / ((int64_t)N * K));
  const float u = up[(((int64_t)z * (K / 32) + k / 32) * N + n) * 32 + (k & 31)];
  // MFMA step jj of group kk pairs k = 8 kk + jj (lanes 0-31) with k = 8 kk + 4 + jj (lanes 32-63), as the A fragments do
  const int kk = k >> 3, hh = (k >> 2) & 1, jj = k & 3;
  uq[((((int64_t)z * (N / 32) + n / 32) * (K / 8) + kk) * 64 + hh * 32 + (n & 31)) * 4 + jj] = u;
}

// ---- split-limb arithmetic (float32-equivalent products on the bf16 matrix pipe) --------------------------------------
// Every float32 operand is written EXACTLY as the sum of three bf16 limbs, x = hi + mid + lo (round-to-nearest at each
// level: |mid| <= 2^-9 |x|, |lo| <= 2^-18 |x|; 3 x 8 significand bits cover float32's 24, the remainders x - hi and
// x - hi - mid are exact in float32).  A product of two limbs has a 16-bit significand, i.e. is exact in float32, and the
// bf16 MFMA accumulates in float32.  So
//     a * b = hi hi + (hi mid + mid hi) + (mid mid + hi lo + lo hi) + [mid lo + lo mid + lo lo]
// and the six leading limb products carry a*b to within 2^-26 |a b| (the dropped bracket) -- below float32's own
// rounding unit 2^-24: the accuracy class of the float32 FMA chain of the f32-input MFMA (measured against a float64
// GEMM: tests/test_ops_gpu.py::test_split_limb_gemm_is_float32_accurate).  The kernel is winograd_gemm_rb16s_kernel
// above.  (Round 2's form -- both operands through LDS, filters as limb planes from memory, 32x32x16 tiles of 64 rows --
// was slower than the f32-input kernels it was meant to beat and left the tree in round 5: DESIGN_HISTORY.md.)

// ---- output transform + layer epilogue: one thread = one tile x 4 channels -----------------------------
template <int MODE>  // 0: y = relu?(Y + bias); 1: y = Y * (x_in > 0) + addend
__global__ void __launch_bounds__(256) winograd_output_kernel(const float* __restrict__ M, const float* __restrict__ aux0,
                                                              const float* __restrict__ aux1, float* __restrict__ y,
                                                              int B, int H, int W, int N, int TH, int TW, int relu) {
  const int N4 = N >> 2;
  const int64_t T = (int64_t)B * TH * TW;
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= T * N4) return;
  const int c4 = (int)(gid % N4);
  const int64_t tile = gid / N4;
  const int tx = (int)(tile % TW), ty = (int)((tile / TW) % TH), b = (int)(tile / ((int64_t)TW * TH));
  const int64_t comp_stride = T * N;
  const float* mi = M + tile * N + 4 * c4;
  float4 m[4][4];
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int s = 0; s < 4; ++s) m[r][s] = *reinterpret_cast<const float4*>(mi + (int64_t)(r * 4 + s) * comp_stride);
  // A^T = [[1,1,1,0],[0,1,-1,-1]]    Y = A^T m A
  float4 t[2][4];
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    t[0][s] = make_float4(m[0][s].x + m[1][s].x + m[2][s].x, m[0][s].y + m[1][s].y + m[2][s].y,
                          m[0][s].z + m[1][s].z + m[2][s].z, m[0][s].w + m[1][s].w + m[2][s].w);
    t[1][s] = make_float4(m[1][s].x - m[2][s].x - m[3][s].x, m[1][s].y - m[2][s].y - m[3][s].y,
                          m[1][s].z - m[2][s].z - m[3][s].z, m[1][s].w - m[2][s].w - m[3][s].w);
  }
#pragma unroll
  for (int a = 0; a < 2; ++a) {
    const int yy = 2 * ty + a;
    if (yy >= H) continue;
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const int xx = 2 * tx + c;
      if (xx >= W) continue;
      float4 v;
      if (c == 0) v = make_float4(t[a][0].x + t[a][1].x + t[a][2].x, t[a][0].y + t[a][1].y + t[a][2].y,
                                  t[a][0].z + t[a][1].z + t[a][2].z, t[a][0].w + t[a][1].w + t[a][2].w);
      else v = make_float4(t[a][1].x - t[a][2].x - t[a][3].x, t[a][1].y - t[a][2].y - t[a][3].y,
                           t[a][1].z - t[a][2].z - t[a][3].z, t[a][1].w - t[a][2].w - t[a][3].w);
      const int64_t idx = (((int64_t)b * H + yy) * W + xx) * N + 4 * c4;
      if (MODE == 0) {
        if (aux0) {
          const float4 bb = *reinterpret_cast<const float4*>(aux0 + 4 * c4);
          v.x += bb.x; v.y += bb.y; v.z += bb.z; v.w += bb.w;
        }
        if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
      } else {
        if (aux0) {
          const float4 xin = *reinterpret_cast<const float4*>(aux0 + idx);
          v.x = xin.x > 0.f ? v.x : 0.f; v.y = xin.y > 0.f ? v.y : 0.f;
          v.z = xin.z > 0.f ? v.z : 0.f; v.w = xin.w > 0.f ? v.w : 0.f;
        }
        if (aux1) {
          const float4 ad = *reinterpret_cast<const float4*>(aux1 + idx);
          v.x += ad.x; v.y += ad.y; v.z += ad.z; v.w += ad.w;
        }
      }
      *reinterpret_cast<float4*>(y + idx) = v;
    }
  }
}

// ---- host side -----------------------------------------------------------------------------------------------
// ---- optional per-launch event timing of the GEMM kernel (nfs_gemm_timer) ---------------------------------
struct GemmTimerRec { hipEvent_t e0, e1; double flops; int split = 0; double bytes = 0.0; };   // split: a split-limb (bf16 MFMA) launch; bytes: V + U + M
static std::atomic<bool> g_timer_on{false};
static std::vector<GemmTimerRec> g_timer_recs;
static std::mutex g_timer_mu;

template <int BM, int BN, int NBUF>
static void launch_gemm_variant(const WgGemmArgs& a, hipStream_t s) {
  const size_t oper = NBUF * (BM + BN) * WG_LS, tile = BM * (BN + 4);
  const size_t lds = (oper > tile ? oper : tile) * sizeof(float);
  static std::once_flag attr_once;   // (one set per kernel instance, safe from several host threads)
  if (lds > 65536) std::call_once(attr_once, [&] {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(winograd_gemm_kernel<BM, BN, NBUF>),
                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); });
  const int total = a.mt * a.nt * a.Z, grid = (total + WG_XCDS - 1) / WG_XCDS * WG_XCDS;
  GemmTimerRec rec{nullptr, nullptr, 2.0 * a.Z * (double)a.T * a.K * a.N};
  const bool timed = g_timer_on && hipEventCreate(&rec.e0) == hipSuccess && hipEventCreate(&rec.e1) == hipSuccess;
  if (timed) (void)hipEventRecord(rec.e0, s);
  hipLaunchKernelGGL((winograd_gemm_kernel<BM, BN, NBUF>), dim3(grid), dim3(256), lds, s, a);
  if (timed) {
    (void)hipEventRecord(rec.e1, s);
    std::lock_guard<std::mutex> lk(g_timer_mu);
    g_timer_recs.push_back(rec);
  }
}

// Tile choice.  All co-resident blocks of a CU share its 4 MFMA pipes, so a CU's time is (number of tiles it
// processes) x (tile work); blocks are dealt to CU slots as they free up.  cost = [full rounds x blocks/CU +
// ceil(left-over blocks / CUs)] x BM x BN, with M padded to BM: small T (deep layers, few views per GPU) wants
// BM = 64, a grid that just misses a round boundary wants the other aspect ratio.
static void pick_gemm_tile(int64_t T, int N, int Z, int cus, int* bm_out, int* bn_out) {
  static const int force_bm = [] { const char* e = getenv("NFS_GEMM_BM"); return e ? atoi(e) : 0; }();
  static const int force_bn = [] { const char* e = getenv("NFS_GEMM_BN"); return e ? atoi(e) : 0; }();
  double best = 1e300;
  *bm_out = 128; *bn_out = 64;
  const int bms[2] = {128, 64}, bns[2] = {64, 128};
  for (int bi = 0; bi < 2; ++bi)
    for (int ni = 0; ni < 2; ++ni) {
      const int bm = bms[bi], bn = bns[ni];
      if (N % bn) continue;
      if (force_bm && bm != force_bm) continue;
      if (force_bn && bn != force_bn && N % force_bn == 0) continue;
      const size_t oper = 2 * (bm + bn) * WG_LS, tile = (size_t)bm * (bn + 4);
      const size_t lds = (oper > tile ? oper : tile) * sizeof(float);
      int bpc = (int)(160 * 1024 / lds);
      const int vg = (bm * bn >= 128 * 128) ? 3 : 4;      // waves/SIMD the register budget allows
      if (bpc > vg) bpc = vg;
      const int64_t blocks = ((T + bm - 1) / bm) * (N / bn) * Z;
      const int64_t slots = (int64_t)cus * bpc, full = blocks / slots, rem = blocks % slots;
      double cost = (double)(full * bpc + (rem + cus - 1) / cus) * bm * bn;
      cost *= 1.0 + 8.0 / (bm < bn ? bm : bn);            // smaller tiles: more operand traffic per flop
      if (cost < best) { best = cost; *bm_out = bm; *bn_out = bn; }
    }
}

// 0: float32-input MFMA (v_mfma_f32_32x32x2_f32); 1: split-limb form on the bf16 MFMA (float32-equivalent, see
// winograd_gemm_split_kernel).  Process-wide setting (nfs_gemm_mode); NFS_GEMM_MODE presets it.
static std::atomic<int> g_gemm_mode{[] { const char* e = getenv("NFS_GEMM_MODE"); return e ? (atoi(e) == 0 ? 0 : 1) : 1; }()};

template <int BM, int BN>
static void launch_gemm_rb(const WgGemmArgs& a, hipStream_t s) {
  const size_t oper = 2 * BM * WG_LS, tile = (BM / 2) * (BN + 4);
  const size_t lds = (oper > tile ? oper : tile) * sizeof(float);
  static std::once_flag attr_once;   // (one set per kernel instance, safe from several host threads)
  if (lds > 65536) std::call_once(attr_once, [&] {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(winograd_gemm_rb_kernel<BM, BN>),
                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); });
  const int total = a.mt * a.nt * a.Z, grid = (total + WG_XCDS - 1) / WG_XCDS * WG_XCDS;
  GemmTimerRec rec{nullptr, nullptr, 2.0 * a.Z * (double)a.T * a.K * a.N};
  const bool timed = g_timer_on && hipEventCreate(&rec.e0) == hipSuccess && hipEventCreate(&rec.e1) == hipSuccess;
  if (timed) (void)hipEventRecord(rec.e0, s);
  hipLaunchKernelGGL((winograd_gemm_rb_kernel<BM, BN>), dim3(grid), dim3(256), lds, s, a);
  if (timed) {
    (void)hipEventRecord(rec.e1, s);
    std::lock_guard<std::mutex> lk(g_timer_mu);
    g_timer_recs.push_back(rec);
  }
}

template <int MT16, int NW16>
static void launch_gemm_rb16(const WgGemmArgs& a, hipStream_t s) {
  constexpr int BM = 16 * MT16, BN = 64 * NW16, BMP = (BM + 31) / 32 * 32;
  constexpr int EPMAX = NW16 >= 4 ? 2 : 5;
  const size_t oper = 2 * BMP * WG_LS, tile = 16 * (MT16 < EPMAX ? MT16 : EPMAX) * (BN + 4);
  const size_t lds = (oper > tile ? oper : tile) * sizeof(float);
  static std::once_flag attr_once;
  if (lds > 65536) std::call_once(attr_once, [&] {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(winograd_gemm_rb16_kernel<MT16, NW16>),
                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); });
  const int total = a.mt * a.nt * a.Z * a.ksplit, grid = (total + WG_XCDS - 1) / WG_XCDS * WG_XCDS;
  GemmTimerRec rec{nullptr, nullptr, 2.0 * a.Z * (double)a.T * a.K * a.N};
  const bool timed = g_timer_on && hipEventCreate(&rec.e0) == hipSuccess && hipEventCreate(&rec.e1) == hipSuccess;
  if (timed) (void)hipEventRecord(rec.e0, s);
  hipLaunchKernelGGL((winograd_gemm_rb16_kernel<MT16, NW16>), dim3(grid), dim3(256), lds, s, a);
  if (timed) {
    (void)hipEventRecord(rec.e1, s);
    std::lock_guard<std::mutex> lk(g_timer_mu);
    g_timer_recs.push_back(rec);
  }
}

// Which form of B a launch reads: the limb planes (6 bytes per filter value, no split in the kernel) from NFS_RB16S_PRE_ROWS
// rows on, the float32 fragment pack split in registers (4 bytes per value) below -- a launch of a few dozen rows (one or
// two views per GPU) is bound by its filter stream, 51 MB at conv4_x, and the planes would make that 77 (one view 0.958 ->
// 0.997 ms, configs[1] 0.604 -> 0.654 with planes everywhere).  The limbs are the same numbers either way: bit-identical
// results, so the choice may depend on T (the environment variable of the same name moves the threshold, read once:
// tests/test_ops_gpu.py runs both forms on one input in two processes).  NFS_RB16S_PRE=0 builds: never the planes.
#ifndef NFS_RB16S_PRE_ROWS
#define NFS_RB16S_PRE_ROWS 128
#endif
template <int MT16, int NW16, bool PRE>
static void launch_gemm_rb16s_pre(const WgGemmArgs& a, hipStream_t s) {
  constexpr int BM = 16 * MT16, BN = 64 * NW16, BMP = (BM + 31) / 32 * 32;
  constexpr int EPMAX = NW16 >= 4 ? 2 : 5;
  const size_t oper = (size_t)2 * 3 * BMP * WS_RB, tile = (size_t)16 * (MT16 < EPMAX ? MT16 : EPMAX) * (BN + 4) * sizeof(float);
  const size_t lds = oper > tile ? oper : tile;
  static std::once_flag attr_once;
  if (lds > 65536) std::call_once(attr_once, [&] {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(winograd_gemm_rb16s_kernel<MT16, NW16, PRE>),
                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); });
  const int total = a.mt * a.nt * a.Z * a.ksplit, grid = (total + WG_XCDS - 1) / WG_XCDS * WG_XCDS;
  GemmTimerRec rec{nullptr, nullptr, 2.0 * a.Z * (double)a.T * a.K * a.N};
  const bool timed = g_timer_on && hipEventCreate(&rec.e0) == hipSuccess && hipEventCreate(&rec.e1) == hipSuccess;
  if (timed) (void)hipEventRecord(rec.e0, s);
  hipLaunchKernelGGL((winograd_gemm_rb16s_kernel<MT16, NW16, PRE>), dim3(grid), dim3(256), lds, s, a);
  if (timed) {
    rec.split = 1;
    rec.bytes = a.Z * (4.0 * a.T * a.K + (PRE ? 6.0 : 4.0) * a.K * a.N + 4.0 * a.T * a.N * a.ksplit);   // V, filters, M
    (void)hipEventRecord(rec.e1, s);
    std::lock_guard<std::mutex> lk(g_timer_mu);
    g_timer_recs.push_back(rec);
  }
}
template <int MT16, int NW16>
static void launch_gemm_rb16s(const WgGemmArgs& a, hipStream_t s) {
  static const int64_t pre_rows = [] { const char* e = getenv("NFS_RB16S_PRE_ROWS"); return e ? atoll(e) : (int64_t)NFS_RB16S_PRE_ROWS; }();
  if (NFS_RB16S_PRE && a.Ub16 && a.T >= pre_rows) launch_gemm_rb16s_pre<MT16, NW16, true>(a, s);
  else launch_gemm_rb16s_pre<MT16, NW16, false>(a, s);
}

// the register-B kernel takes the plain Winograd GEMMs (packed filters, no mask / scale) with 32-bit operand offsets
// row tiles of the 16-row form: 80 (5 MFMA tiles), 48, 112, 208 (e.g. the 200 rows of an F(5x5) layer at 8 views)
static const int kRb16Rows[4] = {80, 48, 112, 208};
static bool rb16_rows_ok(int bm) { return bm == 80 || bm == 48 || bm == 112 || bm == 208; }
// rows the 16-row form executes with row tile bm: the 80- and 48-row blocks run only the 16-row tiles that exist (their
// last block is ragged), the taller ones keep whole zero-padded blocks
static int64_t rb16_rows_executed(int64_t T, int bm) { return bm <= 80 ? (T + 15) / 16 * 16 : (T + bm - 1) / bm * bm; }
static int64_t rb16_best_rows(int64_t T, int* bm_out) {
  int64_t best = -1;
  for (int i = 0; i < 4; ++i) {
    const int64_t p = rb16_rows_executed(T, kRb16Rows[i]);
    if (best < 0 || p < best) { best = p; if (bm_out) *bm_out = kRb16Rows[i]; }
  }
  return best;
}

// ... the 16-row form also scales and masks (the Gram gradient runs on it, its symmetric D read in place)
static bool gemm_rb16_applies(const WgGemmArgs& a) {
  static const bool off = [] { const char* e = getenv("NFS_GEMM_RB"); return e && atoi(e) == 0; }();
  return !off && a.Uq16 && a.T * a.K * 4 < ((int64_t)1 << 31) && (int64_t)a.K * a.N * 4 < ((int64_t)1 << 31);
}
static bool gemm_rb_applies(const WgGemmArgs& a) {
  static const bool off = [] { const char* e = getenv("NFS_GEMM_RB"); return e && atoi(e) == 0; }();
  return !off && a.Uq && !a.mask && !a.alpha_dev && a.alpha == 1.f && a.T * a.K * 4 < ((int64_t)1 << 31) &&
         (int64_t)a.K * a.N * 4 < ((int64_t)1 << 31);
}

void winograd_pack_frag16(const float* up, float* uq, int K, int N, int64_t total, hipStream_t s) {
  hipLaunchKernelGGL(winograd_pack_frag16_kernel, dim3(blocks_for(total, 256)), dim3(256), 0, s, up, uq, K, N, total);
}

void winograd_pack_limbs16(const float* uq16, float* ub16, int K, int N, int Z, hipStream_t s) {
  const int64_t total = (int64_t)Z * (N / 16) * (K / 32) * 64;
  hipLaunchKernelGGL(winograd_pack_limbs16_kernel, dim3(blocks_for(total, 256)), dim3(256), 0, s,
                     reinterpret_cast<const float4*>(uq16), reinterpret_cast<uint4*>(ub16), total);
}

static unsigned long long* g_gemm_prof = nullptr;        // NFS_ABLATE builds only (nfs_gemm_prof)

static void launch_gemm_tile(WgGemmArgs a, int Z, int bm, int bn, hipStream_t s, int variant = 0) {
  a.prof = g_gemm_prof;
  static const int nbuf_env = [] { const char* e = getenv("NFS_GEMM_NBUF"); return e ? atoi(e) : 0; }();
  const int nbuf = nbuf_env == 1 ? 1 : 2;
  a.mt = (int)((a.T + bm - 1) / bm);
  a.nt = a.N / bn;
  a.Z = Z;
#ifdef NFS_ABLATE
  // timing-only ablations, -DNFS_ABLATE builds only (wrong results by construction): 1 no MFMA / operand reads, 2 no
  // stores of C, 4 no global loads after the first chunk, 8 no LDS staging, 16 no barrier in the K loop
  static const int dbg = getenv("NFS_GEMM_DBG") ? atoi(getenv("NFS_GEMM_DBG")) : 0;
  a.dbg = dbg;
#endif
  // (a split-limb choice that meets a launch with a mask / scale / symmetric operand runs the same tile on the f32-input
  // MFMA: never the generic kernel with the 16-row form's tile height)
  if (variant == 3 && (a.mask || a.alpha_dev || a.alpha != 1.f || a.symb)) variant = 2;
  if (variant == 3 && gemm_rb16_applies(a) && rb16_rows_ok(bm) && a.N % bn == 0) {
    // the 16-row register-B form in split-limb arithmetic (mode 1): 64- or 128-column tiles
    a.mt = (int)((a.T + bm - 1) / bm);
    if (bn > 128 && bm == 208) { bn = 128; a.nt = a.N / 128; }
    if (bm == 80) { if (bn == 256) launch_gemm_rb16s<5, 4>(a, s); else if (bn == 128) launch_gemm_rb16s<5, 2>(a, s); else launch_gemm_rb16s<5, 1>(a, s); }
    else if (bm == 48) { if (bn == 256) launch_gemm_rb16s<3, 4>(a, s); else if (bn == 128) launch_gemm_rb16s<3, 2>(a, s); else launch_gemm_rb16s<3, 1>(a, s); }
    else if (bm == 112) { if (bn == 256) launch_gemm_rb16s<7, 4>(a, s); else if (bn == 128) launch_gemm_rb16s<7, 2>(a, s); else launch_gemm_rb16s<7, 1>(a, s); }
    else { if (bn == 128) launch_gemm_rb16s<13, 2>(a, s); else launch_gemm_rb16s<13, 1>(a, s); }
    return;
  }
  if (variant == 2 && gemm_rb16_applies(a) && rb16_rows_ok(bm) && a.N % bn == 0) {
    a.mt = (int)((a.T + bm - 1) / bm);
    // (256-column tiles for the 80- / 48- / 112-row forms: four times the MFMA work per block between its prologue
    // and its epilogue; a tuner candidate where N % 256 == 0)
    if (bm == 80) { if (bn == 256) launch_gemm_rb16<5, 4>(a, s); else if (bn == 128) launch_gemm_rb16<5, 2>(a, s); else launch_gemm_rb16<5, 1>(a, s); }
    else if (bm == 48) { if (bn == 256) launch_gemm_rb16<3, 4>(a, s); else if (bn == 128) launch_gemm_rb16<3, 2>(a, s); else launch_gemm_rb16<3, 1>(a, s); }
    else if (bm == 112) { if (bn == 256) launch_gemm_rb16<7, 4>(a, s); else if (bn == 128) launch_gemm_rb16<7, 2>(a, s); else launch_gemm_rb16<7, 1>(a, s); }
    else {
      if (bn == 256) { bn = 128; a.nt = a.N / 128; }          // (no 256-column instance of the 208-row form: 208 accumulators)
      if (bn == 128) launch_gemm_rb16<13, 2>(a, s); else launch_gemm_rb16<13, 1>(a, s);
    }
    return;
  }
  if (variant == 1 && gemm_rb_applies(a)) {
    if (bm == 128 && bn == 128) launch_gemm_rb<128, 128>(a, s);
    else if (bm == 128) launch_gemm_rb<128, 64>(a, s);
    else if (bn == 128) launch_gemm_rb<64, 128>(a, s);
    else launch_gemm_rb<64, 64>(a, s);
    return;
  }
  // K = 64 (two chunks): nothing to double-buffer; a single LDS buffer doubles the co-resident blocks of this
  // bandwidth-bound shape (conv1_2: 0.103 -> 0.093 ms)
  if (nbuf == 2 && (a.K > 64 || nbuf_env == 2)) {
    if (bm == 128 && bn == 128) launch_gemm_variant<128, 128, 2>(a, s);
    else if (bm == 128) launch_gemm_variant<128, 64, 2>(a, s);
    else if (bn == 128) launch_gemm_variant<64, 128, 2>(a, s);
    else launch_gemm_variant<64, 64, 2>(a, s);
  } else {
    if (bm == 128 && bn == 128) launch_gemm_variant<128, 128, 1>(a, s);
    else if (bm == 128) launch_gemm_variant<128, 64, 1>(a, s);
    else if (bn == 128) launch_gemm_variant<64, 128, 1>(a, s);
    else launch_gemm_variant<64, 64, 1>(a, s);
  }
}

// Tile shape per problem: the planner's model (pick_gemm_tile) misses by up to 15 % on single layers (it knows nothing
// about L2 behaviour), so the first launch of every (T, K, N, Z, arithmetic) shape times the four candidates on the
// device (two runs each, HIP events, the launch's own operands -- every candidate computes the identical result, the
// k order does not depend on the tile shape) and the fastest is remembered for the process.  Not while a stream
// capture is in progress, not while the GEMM timer brackets launches, and not with NFS_GEMM_BM / BN / NFS_GEMM_TUNE=0.
struct GemmKey {
  int64_t T; int K, N, Z, mode;
  bool operator<(const GemmKey& o) const {
    return std::tie(T, K, N, Z, mode) < std::tie(o.T, o.K, o.N, o.Z, o.mode);
  }
};
static std::map<GemmKey, std::tuple<int, int, int>> g_tile_cache;   // (BM, BN, kernel variant: 0 LDS-B, 1 register-B)
static std::mutex g_tile_mu;

int winograd_ksplit(int64_t T, int K) {
  static const int forced = [] { const char* e = getenv("NFS_GEMM_KSPLIT"); return e ? atoi(e) : 0; }();
  static const int tmax = [] { const char* e = getenv("NFS_GEMM_KSPLIT_T"); return e ? atoi(e) : 64; }();
  static const int kmin = [] { const char* e = getenv("NFS_GEMM_KSPLIT_K"); return e ? atoi(e) : 512; }();
  int ks = forced > 0 ? (forced > 2 ? 2 : forced) : (T <= tmax && K >= kmin ? 2 : 1);    // (the output transforms sum 1 or 2)
  while (ks > 1 && (K / WG_KC) % ks) ks >>= 1;
  return ks < 1 ? 1 : ks;
}

static int launch_batched_gemm(WgGemmArgs a, int Z, int cus, hipStream_t s) {
  static const bool tune = [] {
    const char* e = getenv("NFS_GEMM_TUNE");
    return !(e && atoi(e) == 0) && !getenv("NFS_GEMM_BM") && !getenv("NFS_GEMM_BN");
  }();
  // NFS_GEMM_RB=2 (with NFS_GEMM_BM / BN or NFS_GEMM_TUNE=0): always the register-B kernel where it applies (tests)
  // NFS_GEMM_RB=3 NFS_GEMM_BM=80|48 NFS_GEMM_BN=128|64: always the 16-row form
  static const int force_rb = [] { const char* e = getenv("NFS_GEMM_RB"); const int v = e ? atoi(e) : 0; return v == 2 ? 1 : v == 3 ? 2 : 0; }();
  int bm, bn, variant = force_rb;
  pick_gemm_tile(a.T, a.N, Z, cus, &bm, &bn);
  // Which MFMA the GEMM runs on is decided by the shape alone (never by a measurement: the two instructions sum k in
  // different groupings, so their results differ in the last bit): the 16-row form wherever it executes no more rows
  // than the best 32-row tiling (NFS_GEMM_ROWS16_PCT, default 100: at equal rows it measured 3-5 % faster).  Within a family every candidate computes the identical result, and the tuner
  // measures.
  static const int rows16_pct = [] { const char* e = getenv("NFS_GEMM_ROWS16_PCT"); return e ? atoi(e) : 100; }();
  int bm16 = 80;
  const int64_t pad32 = (a.T + 63) / 64 * 64, pad16 = rb16_best_rows(a.T, &bm16);
  static const bool bm_forced = getenv("NFS_GEMM_BM") != nullptr;
  // mode 1: a plain product takes the split-limb instance of the 16-row form (variant 3); the Gram gradient (mask /
  // scale / symmetric B) stays on the f32-input MFMA
  const bool plain = !a.mask && !a.alpha_dev && a.alpha == 1.f && !a.symb;
  const int mode = g_gemm_mode;
  const int v16 = (mode == 1 && plain) ? 3 : 2;
  const bool rows16 = gemm_rb16_applies(a) && force_rb != 1 && !bm_forced &&
                      pad16 * 100 <= pad32 * rows16_pct;
  // K parts (winograd_ksplit: by shape alone) only on the 16-row register-B form of a plain product (no mask / scale in
  // the epilogue: those apply to the complete sum)
  a.ksplit = (rows16 && !a.mask && !a.alpha_dev && a.alpha == 1.f && !a.symb) ? winograd_ksplit(a.T, a.K) : 1;
  if (rows16 && !tune) { variant = v16; bm = bm16; bn = a.N % 128 == 0 ? 128 : 64; }
  if (force_rb == 2) {
    static const int fbm = [] { const char* e = getenv("NFS_GEMM_BM"); return e ? atoi(e) : 80; }();
    static const int fbn = [] { const char* e = getenv("NFS_GEMM_BN"); return e ? atoi(e) : 128; }();
    if (rb16_rows_ok(fbm) && a.N % fbn == 0 && gemm_rb16_applies(a)) { bm = fbm; bn = fbn; } else variant = 0;
  }
  if (tune) {
    // (the arithmetic of a cached choice is part of the key: in mode 1 a plain product runs variant 3, a Gram gradient of
    // the same (T, K, N, Z) variant 2 -- one must never inherit the other's entry)
    const GemmKey key{a.T, a.K, a.N, Z, mode * 4 + (plain ? 2 : 0) + (a.mask ? 1 : 0)};
    if (rows16) { variant = v16; bm = bm16; bn = a.N % 128 == 0 ? 128 : 64; }   // (capture / timer: no trial)
    std::unique_lock<std::mutex> lk(g_tile_mu);
    auto it = g_tile_cache.find(key);
    if (it != g_tile_cache.end()) {
      bm = std::get<0>(it->second); bn = std::get<1>(it->second); variant = std::get<2>(it->second);
    } else {
      hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
      const bool capturing = hipStreamIsCapturing(s, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone;
      hipEvent_t e0, e1;
      if (!capturing && !g_timer_on && hipEventCreate(&e0) == hipSuccess && hipEventCreate(&e1) == hipSuccess) {
        float best = 1e30f;
        auto trial = [&](int cbm, int cbn, int var) {
          launch_gemm_tile(a, Z, cbm, cbn, s, var);                     // warm (L2, instruction cache)
          // the fastest of three pairs: one pair alone mis-ranked candidates 5-10 % apart when something else (a
          // profiler, another process) touched the host or the GPU during the trial
          for (int rep = 0; rep < 3; ++rep) {
            (void)hipEventRecord(e0, s);
            launch_gemm_tile(a, Z, cbm, cbn, s, var);
            launch_gemm_tile(a, Z, cbm, cbn, s, var);
            (void)hipEventRecord(e1, s);
            float ms = 1e30f;
            if (hipEventSynchronize(e1) != hipSuccess || hipEventElapsedTime(&ms, e0, e1) != hipSuccess) ms = 1e30f;
            if (ms < best) { best = ms; bm = cbm; bn = cbn; variant = var; }
          }
        };
        if (rows16) {                                                   // 16-row tiles: 80 / 48 rows x 128 / 64 columns
          for (int i = 0; i < 4; ++i) {                                  // row tiles that pad no worse than 1.2 x the best
            const int64_t p = rb16_rows_executed(a.T, kRb16Rows[i]);
            if (p * 10 > pad16 * 12) continue;
            static const bool bn256 = [] { const char* e = getenv("NFS_GEMM_BN256"); return !(e && atoi(e) == 0); }();
            for (int cbn = bn256 ? 256 : 128; cbn >= 64; cbn /= 2)
              if (a.N % cbn == 0 && !(cbn == 256 && kRb16Rows[i] == 208)) trial(kRb16Rows[i], cbn, v16);
          }
        } else {
          const int cand[4][2] = {{64, 64}, {64, 128}, {128, 64}, {128, 128}};
          const int nvar = (g_gemm_mode == 0 && gemm_rb_applies(a)) ? 2 : 1;
          for (int var = 0; var < nvar; ++var)
            for (int c = 0; c < 4; ++c)
              if (a.N % cand[c][1] == 0) trial(cand[c][0], cand[c][1], var);
        }
        (void)hipEventDestroy(e0);
        (void)hipEventDestroy(e1);
        g_tile_cache[key] = std::make_tuple(bm, bn, variant);
        static const bool log = getenv("NFS_GEMM_TUNE_LOG") != nullptr;
        if (log)
          fprintf(stderr, "gemm tuner: Z=%d T=%lld K=%d N=%d mask=%d ksplit=%d -> %d x %d (variant %d), %.1f us / launch\n",
                  Z, (long long)a.T, a.K, a.N, a.mask ? 1 : 0, a.ksplit, bm, bn, variant, 500.f * best);
        return a.ksplit;                                              // the result is already in place
      }
    }
  }
  if (variant != 2 && variant != 3) a.ksplit = 1;                     // (only the rb16 kernels know about K parts)
  launch_gemm_tile(a, Z, bm, bn, s, variant);
  return a.ksplit;
}

int winograd_launch_batched_gemm(const WgGemmArgs& a, int Z, int cus, hipStream_t s) { return launch_batched_gemm(a, Z, cus, s); }

// dF[b] = alpha_b * F[b] @ D[b] (D symmetric, so row n of D serves as column n), optional (F > 0) mask
int gram_bwd_gemm(const float* F, const float* Dm, float* dF, int B, int HW, int C, float alpha, const float* alpha_dev,
                  int relu_mask, int cus, hipStream_t s) {
  WgGemmArgs a{F, Dm, dF, (int64_t)HW, C, C, (int64_t)C * C, 32, C, alpha, alpha_dev, relu_mask ? F : nullptr};
  a.Uq16 = Dm;               // rb16 reads the symmetric D in place (row n as column n)
  a.symb = 1;
  launch_batched_gemm(a, B, cus, s);
  return check_launch("gram_bwd_gemm");
}

// The Gram gradients of n layers as ONE launch of the 16-row register-B GEMM (80 x 64 tiles: the common denominator of
// C = 64 ... 512), problems ordered deep K first (long tiles first, the two-chunk tiles of the 64-channel layer fill the
// tail).  Same per-tile arithmetic as gram_bwd_gemm on the rb16 kernel: bit-identical results.
int gram_bwd_gemm_group(const float* const* F, const float* const* Dm, float* const* dF, const int* HW, const int* C,
                        const float* alpha, const int* relu_mask, int n, int B, hipStream_t s) {
  if (n < 1 || n > WG_MAXG) return -1;
  int order[WG_MAXG];
  for (int i = 0; i < n; ++i) order[i] = i;
  for (int i = 1; i < n; ++i)
    for (int j = i; j > 0 && C[order[j]] > C[order[j - 1]]; --j) { const int t = order[j]; order[j] = order[j - 1]; order[j - 1] = t; }
  WgGroupArgs G;
  G.n = n;
  int ub = 0;
  double flops = 0;
  for (int k = 0; k < n; ++k) {
    const int l = order[k];
    WgGemmArgs a{F[l], Dm[l], dF[l], (int64_t)HW[l], C[l], C[l], (int64_t)C[l] * C[l], 32, C[l], alpha[l], nullptr,
                 relu_mask[l] ? F[l] : nullptr};
    a.Uq16 = Dm[l];
    a.symb = 1;
    if (!gemm_rb16_applies(a) || C[l] % 64) return -1;
    a.mt = (int)((a.T + 79) / 80);
    a.nt = a.N / 64;
    a.Z = B;
    G.g[k] = a;
    G.ustart[k] = ub;
    ub += (a.mt * a.nt * a.Z + WG_XCDS - 1) / WG_XCDS * WG_XCDS;
    flops += 2.0 * B * (double)a.T * a.K * a.N;
  }
  G.ustart[n] = ub;
  constexpr int BMP = 96;
  const size_t oper = 2 * BMP * WG_LS, tile = 16 * 5 * (64 + 4);
  const size_t lds = (oper > tile ? oper : tile) * sizeof(float);
  GemmTimerRec rec{nullptr, nullptr, flops};
  const bool timed = g_timer_on && hipEventCreate(&rec.e0) == hipSuccess && hipEventCreate(&rec.e1) == hipSuccess;
  if (timed) (void)hipEventRecord(rec.e0, s);
  hipLaunchKernelGGL((winograd_gemm_rb16_group_kernel<5, 1>), dim3(ub), dim3(256), lds, s, G);
  if (timed) {
    (void)hipEventRecord(rec.e1, s);
    std::lock_guard<std::mutex> lk(g_timer_mu);
    g_timer_recs.push_back(rec);
  }
  return check_launch("gram_bwd_gemm_group");
}

// ---- Winograd host side (called from vgg.hip) ------------------------------------------------------------
int winograd_tile() {
  static const int m = [] { const char* e = getenv("NFS_WINOGRAD_TILE"); return (e && atoi(e) == 2) ? 2 : 4; }();
  return m;
}

// Which form a (non-pooled) conv call takes: 1 the single-kernel path (narrow layers), 2 F(5x5) (deep layers whose
// image F(4x4) would pad heavily), 0 the three-kernel F(4x4) / F(2x2).  A function of the shapes alone.
int winograd_path(int B, int H, int W, int K, int N) {
  if (winograd_tile() != 4) return 0;
  if (winograd_fusable(K, N) && winograd_fused_takes(B, H, W, K, N)) return 1;
  return winograd5_takes(H, W, K, N) ? 2 : 0;
}

int64_t winograd_workspace_floats(int B, int H, int W, int K, int N) {
  const int m = winograd_tile();
  const int64_t T = (int64_t)B * ((H + m - 1) / m) * ((W + m - 1) / m);
  // (M once per K part of the GEMM: winograd_ksplit)
  const int64_t f4 = (m + 2) * (m + 2) * T * ((int64_t)K + (int64_t)N * winograd_ksplit(T, K)),
                f5 = m == 4 ? winograd5_workspace_floats(B, H, W, K, N) : 0;
  return f4 > f5 ? f4 : f5;
}

// 36 floats per (ci, co): room for either tile size,
// the filters twice more in MFMA fragment order (register-B GEMM kernels, 32x32x2 and 16x16x4 forms -- the 16x16x4 pack
// also feeds the split-limb kernel, which splits it in registers),
// and, for the layers the single-kernel path takes (winograd_fused.hip), the filters in its fragment order
int64_t winograd_packed_floats(int Ci, int Co) {
  return (int64_t)(36 + 36 + 36 + 54) * Ci * Co + winograd_fused_packed_floats(Ci, Co) + winograd5_packed_floats(Ci, Co);
}

int winograd_pack(const float* w_hwio, float* up, int Ci, int Co, int kind, hipStream_t s) {
  const int64_t n = (int64_t)Ci * Co;
  if (winograd_tile() == 4) {
    hipLaunchKernelGGL(winograd_pack4_kernel, dim3(blocks_for(n, 256)), dim3(256), 0, s, w_hwio, up, Ci, Co, kind);
    const int Kc = kind == 0 ? Ci : Co, Nc = kind == 0 ? Co : Ci;
    hipLaunchKernelGGL(winograd_pack_frag_kernel, dim3(blocks_for(36 * n, 256)), dim3(256), 0, s, up, up + 36 * n, Kc, Nc,
                       36 * n);
    hipLaunchKernelGGL(winograd_pack_frag16_kernel, dim3(blocks_for(36 * n, 256)), dim3(256), 0, s, up, up + 72 * n, Kc,
                       Nc, 36 * n);
    if (winograd_fusable(Kc, Nc))
      if (int e = winograd_pack_fused(up, up + 108 * n, Kc, Nc, s)) return e;
    if (winograd5_channels(Kc, Nc))       // the F(5x5) filters, their fragment order and its limb planes
      if (int e = winograd5_pack(w_hwio, up + 108 * n + winograd_fused_packed_floats(Ci, Co), Ci, Co, kind, s)) return e;
    // the limb planes of the 16 x 16 fragment pack (split-limb GEMM), behind everything else
    winograd_pack_limbs16(up + 72 * n, up + 108 * n + winograd_fused_packed_floats(Ci, Co) + winograd5_packed_floats(Ci, Co),
                          Kc, Nc, 36, s);
  } else
    hipLaunchKernelGGL(winograd_pack_kernel, dim3(blocks_for(n, 256)), dim3(256), 0, s, w_hwio, up, Ci, Co, kind);
  return check_launch("winograd_pack");
}

// x [B,H,W,K] -> y [B,H,W,N]; U packed by winograd_pack; ws >= winograd_workspace_floats
// ypool (mode 0, nullable): also write the 2x2 average pool of y (y itself is then optional).  pooled_grad (mode 1): x is a POOLED gradient
// [B,H/2,W/2,K] that reaches the conv through the ReLU of xmask [B,H,W,K] (see winograd_input4_kernel<true>).
// ReLU bit cache (m == 4 only; see winograd_input4_kernel): mode 0 writes in_bits (mask of x) and out_bits (mask of y),
// mode 1 reads in_bits instead of aux0 (the x_in mask of the result) and out_bits instead of xmask.  All nullable.
int winograd_conv(const float* x, const float* U, const float* aux0, const float* aux1, float* y, float* ws, int B,
                  int H, int W, int K, int N, int mode, int relu, int cus, hipStream_t s, float* ypool,
                  const float* xmask, uint32_t* in_bits, uint32_t* out_bits, bool pooled_grad) {
  const int m = winograd_tile(), comps = (m + 2) * (m + 2);
  // narrow layers: one kernel, no V / M round trip
  if (m == 4 && winograd_fusable(K, N) && winograd_fused_takes(B, H, W, K, N) && (!pooled_grad || mode == 1))
    return winograd_fused_conv(x, U + (int64_t)108 * K * N, aux0, aux1, y, B, H, W, K, N, mode, relu, s, ypool, xmask,
                               in_bits, out_bits, pooled_grad);
  // deep layers with heavily padded F(4x4) tilings: F(5x5) (no pooling fused there; its ReLU bit cache has its own
  // layout, sized by nfs_conv3x3_relu_bits_words for exactly the layers that come here)
  if (m == 4 && !pooled_grad && !ypool && !out_bits && y && winograd5_takes(H, W, K, N))
    return winograd5_conv(x, U + (int64_t)108 * K * N + winograd_fused_packed_floats(K, N), aux0, aux1, y, ws, B, H, W, K, N,
                          mode, relu, cus, s, in_bits);
  const int TH = (H + m - 1) / m, TW = (W + m - 1) / m;
  const int64_t T = (int64_t)B * TH * TW;
  float* V = ws;
  float* M = ws + comps * T * K;
  const dim3 ig((blocks_for(T * (K / 2), 256) + 7) / 8 * 8);
  // few tiles: the six-wave transforms (NFS_W4_WAVES6_MAX (tile, channel) items per launch; 0: never)
  static const int64_t waves6_max = [] { const char* e = getenv("NFS_W4_WAVES6_MAX"); return e ? atoll(e) : (int64_t)65536; }();
  const bool in6 = m == 4 && K % 128 == 0 && T * K <= waves6_max, out6 = m == 4 && N % 128 == 0 && T * N <= waves6_max;
  const dim3 ig6((unsigned)((T * (K / 128) + 7) / 8 * 8));
  if (in6 && pooled_grad && mode == 1 && out_bits)
    hipLaunchKernelGGL(winograd_input4w_kernel<1>, ig6, dim3(64, 6), 0, s, x, V, B, H, W, K, TH, TW, out_bits);
  else if (in6 && !pooled_grad)
    hipLaunchKernelGGL(winograd_input4w_kernel<0>, ig6, dim3(64, 6), 0, s, x, V, B, H, W, K, TH, TW,
                       mode == 0 ? in_bits : (uint32_t*)nullptr);
  else
  if (m == 4 && pooled_grad && mode == 1 && out_bits)   // pooled data gradient: the mask of the layer's own output from the bit cache ...
    hipLaunchKernelGGL(winograd_input4_kernel<1>, ig, dim3(256), 0, s, x, V, B, H, W, K, TH, TW, xmask, out_bits);
  else if (m == 4 && pooled_grad)                       // ... or from the float output
    hipLaunchKernelGGL(winograd_input4_kernel<2>, ig, dim3(256), 0, s, x, V, B, H, W, K, TH, TW, xmask,
                       (uint32_t*)nullptr);
  else if (m == 4)
    hipLaunchKernelGGL(winograd_input4_kernel<0>, ig, dim3(256), 0, s, x, V, B, H, W, K, TH, TW,
                       (const float*)nullptr, mode == 0 ? in_bits : nullptr);
  else
    hipLaunchKernelGGL(winograd_input_kernel, dim3(blocks_for(T * (K / 4), 256)), dim3(256), 0, s, x, V, B, H, W, K, TH,
                       TW);
  WgGemmArgs a{V, U, M, T, K, N, (int64_t)K * N, (int64_t)N * 32, 32, 1.f, nullptr, nullptr};
  if (m == 4) {
    a.Uq = U + (int64_t)36 * K * N;
    a.Uq16 = U + (int64_t)72 * K * N;
    a.Ub16 = U + (int64_t)108 * K * N + winograd_fused_packed_floats(K, N) + winograd5_packed_floats(K, N);
  }
  const int nsplit = launch_batched_gemm(a, comps, cus, s);
  if (m == 4) {
    const unsigned ob = blocks_for(T * (N / 2), 256);
    uint32_t* ib = aux0 ? in_bits : nullptr;                              // a mask only where the caller asks for one
    if (nsplit != 1 && nsplit != 2) {
      set_error("winograd_conv: unsupported number of K parts");
      return NFS_EINVAL;
    }
    if (out6) {
      const dim3 og6((unsigned)(T * (N / 128)));
#define NFS_W4_OUT6(MODE_, NS_, POOL_, BITS_)                                                                         \
      hipLaunchKernelGGL((winograd_output4w_kernel<MODE_, NS_>), og6, dim3(64, 6), 0, s, M, aux0, aux1, y, B, H, W, N, TH, TW, \
                         relu, POOL_, BITS_)
      if (mode == 0 && nsplit == 1) NFS_W4_OUT6(0, 1, ypool, out_bits);
      else if (mode == 0) NFS_W4_OUT6(0, 2, ypool, out_bits);
      else if (nsplit == 1) NFS_W4_OUT6(1, 1, (float*)nullptr, ib);
      else NFS_W4_OUT6(1, 2, (float*)nullptr, ib);
#undef NFS_W4_OUT6
    } else
    if (mode == 0 && nsplit == 1)
      hipLaunchKernelGGL((winograd_output4_kernel<0, 1>), dim3(ob), dim3(256), 0, s, M, aux0, aux1, y, B, H, W, N, TH, TW, relu,
                         ypool, out_bits);
    else if (mode == 0)
      hipLaunchKernelGGL((winograd_output4_kernel<0, 2>), dim3(ob), dim3(256), 0, s, M, aux0, aux1, y, B, H, W, N, TH, TW, relu,
                         ypool, out_bits);
    else if (nsplit == 1)
      hipLaunchKernelGGL((winograd_output4_kernel<1, 1>), dim3(ob), dim3(256), 0, s, M, aux0, aux1, y, B, H, W, N, TH, TW, relu,
                         (float*)nullptr, ib);
    else
      hipLaunchKernelGGL((winograd_output4_kernel<1, 2>), dim3(ob), dim3(256), 0, s, M, aux0, aux1, y, B, H, W, N, TH, TW, relu,
                         (float*)nullptr, ib);
  } else {
    const unsigned ob = blocks_for(T * (N / 4), 256);
    if (mode == 0)
      hipLaunchKernelGGL(winograd_output_kernel<0>, dim3(ob), dim3(256), 0, s, M, aux0, aux1, y, B, H, W, N, TH, TW, relu);
    else
      hipLaunchKernelGGL(winograd_output_kernel<1>, dim3(ob), dim3(256), 0, s, M, aux0, aux1, y, B, H, W, N, TH, TW, relu);
  }
  return check_launch("winograd_conv");
}

}  // namespace nfs

extern "C" {
#ifdef NFS_ABLATE
// measurement builds only: per-wave phase cycle sums of winograd_gemm_kernel, 8 x uint64 per (block, wave):
// [loads+staging wait, barrier, load issue, fragments+MFMA, drain, epilogue, t_begin, t_end]
int nfs_gemm_prof(void* buf) { nfs::g_gemm_prof = reinterpret_cast<unsigned long long*>(buf); return 0; }
#endif
int nfs_gemm_mode(int mode) {
  const int prev = nfs::g_gemm_mode.load();
  if (mode == 0 || mode == 1) nfs::g_gemm_mode.store(mode);
  return prev;
}

int nfs_gemm_timer(int enable) {
  std::lock_guard<std::mutex> lk(nfs::g_timer_mu);
  nfs::g_timer_on = enable != 0;
  return NFS_OK;
}

// which: 0 every record, 1 only the split-limb launches, 2 only the f32-input launches; the records read are removed
static int gemm_timer_read(double* ms_total, double* flops_total, long long* launches, int which, const char* who,
                           double* bytes_total = nullptr) {
  if (!ms_total || !flops_total || !launches) { nfs::set_error("%s: null pointer", who); return NFS_EINVAL; }
  if (hipDeviceSynchronize() != hipSuccess) {
    nfs::set_error("%s: device synchronise failed", who);
    return NFS_ELAUNCH;
  }
  std::lock_guard<std::mutex> lk(nfs::g_timer_mu);
  double ms = 0.0, fl = 0.0, by = 0.0;
  long long n = 0;
  std::vector<nfs::GemmTimerRec> keep;
  for (auto& r : nfs::g_timer_recs) {
    if ((which == 1 && !r.split) || (which == 2 && r.split)) { keep.push_back(r); continue; }
    float t = 0.f;
    if (hipEventElapsedTime(&t, r.e0, r.e1) == hipSuccess) { ms += t; fl += r.flops; by += r.bytes; }
    (void)hipEventDestroy(r.e0);
    (void)hipEventDestroy(r.e1);
    ++n;
  }
  *ms_total = ms;
  *flops_total = fl;
  *launches = n;
  if (bytes_total) *bytes_total = by;
  nfs::g_timer_recs.swap(keep);
  return NFS_OK;
}

int nfs_gemm_timer_read(double* ms_total, double* flops_total, long long* launches) {
  return gemm_timer_read(ms_total, flops_total, launches, 0, "nfs_gemm_timer_read");
}

int nfs_gemm_timer_read_kind(int split_limb, double* ms_total, double* flops_total, long long* launches,
                             double* bytes_total) {
  return gemm_timer_read(ms_total, flops_total, launches, split_limb ? 1 : 2, "nfs_gemm_timer_read_kind", bytes_total);
}
}

