// A6, deep layers whose image side is a multiple of 5 or close to one (25 x 25, 50 x 50): Winograd F(5x5, 3x3) in float32.
//
//   Y = A^T [ (G g G^T) (.) (B^T d B) ] A     per 5 x 5 output tile from a 7 x 7 input patch: 49 products per 25 outputs
//
// (1.96 multiplies per output against F(4x4)'s 2.25), and -- what matters more here -- no tile padding: a 25 x 25 image is
// 5 x 5 tiles of 5 where F(4x4) needs 7 x 7 tiles of 4 (28 x 28: 20 % of its multiplies fall on padding), 50 x 50 is 10 x 10
// against 13 x 13.  49 T5 against 36 T4 rows of GEMM work: 0.69 x at 25 x 25, 0.81 x at 50 x 50; V and M shrink alike.
// Interpolation points 0, +-1, +-2, 1/2, inf (exact rational matrices below, Cook-Toom).  Float32 error against a float64
// convolution, 512 input channels, post-ReLU activations: 4.8e-6 relative where F(4x4) has 2.3e-6 and the direct form
// 7e-7 (tests/test_ops_gpu.py states the budget).  Which form a layer takes is a static function of its shape
// (winograd5_takes), never of a measurement.
//
//   winograd5_input_kernel   x [B,H,W,K]               -> V [49][T][K]      T = B * ceil(H/5) * ceil(W/5) tiles
//   (batched GEMMs: winograd_launch_batched_gemm, Z = 49 -- the 16-row register-B kernel for these row counts)
//   winograd5_output_kernel  M [49][T][N]              -> y [B,H,W,N]   + bias / ReLU (fwd) or ReLU mask / addend (dgrad)
//
// No fused pooling on this path (a 5 x 5 tile does not hold whole 2 x 2 windows): pooled layers stay on F(4x4).  The ReLU
// bit cache has its own layout here -- two words per (5 x 5 tile, channel pair): bit 2 p + c of the 64 = (x > 0) of pixel p,
// channel c -- written by the forward input transform (whose thread holds exactly those 50 values), read by the data
// gradient's output transform (same tiling, one 8-byte load per thread) instead of x_in.
#include "common.h"
#include "winograd_gemm.h"

namespace nfs {

// G (7 x 3) = {-1/2,0,0} {-1/3,-1/3,-1/3} {1/9,-1/9,1/9} {1/36,1/18,1/9} {-1/60,1/30,-1/15} {32/45,16/45,8/45} {0,0,1}
__device__ __forceinline__ void w5_g(const float g0, const float g1, const float g2, float* u) {
  u[0] = -0.5f * g0;
  u[1] = (-1.f / 3.f) * (g0 + g1 + g2);
  u[2] = (1.f / 9.f) * (g0 - g1 + g2);
  u[3] = (1.f / 36.f) * g0 + (1.f / 18.f) * g1 + (1.f / 9.f) * g2;
  u[4] = (-1.f / 60.f) * g0 + (1.f / 30.f) * g1 - (1.f / 15.f) * g2;
  u[5] = (32.f / 45.f) * g0 + (16.f / 45.f) * g1 + (8.f / 45.f) * g2;
  u[6] = g2;
}

// U_z[ci][co] = (G g G^T)[z], z = 7 r + q, packed [49][K/32][N][32] (the layout of winograd_pack4_kernel); kind as there
__global__ void __launch_bounds__(256) winograd5_pack_kernel(const float* __restrict__ w, float* __restrict__ up, int Ci,
                                                             int Co, int kind) {
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
  float t[7][3];
#pragma unroll
  for (int s = 0; s < 3; ++s) {
    float u[7];
    w5_g(g[0][s], g[1][s], g[2][s], u);
#pragma unroll
    for (int r = 0; r < 7; ++r) t[r][s] = u[r];
  }
#pragma unroll
  for (int r = 0; r < 7; ++r) {
    float u[7];
    w5_g(t[r][0], t[r][1], t[r][2], u);
#pragma unroll
    for (int q = 0; q < 7; ++q) up[(((int64_t)(r * 7 + q) * (Kc / 32) + k / 32) * Nc + n) * 32 + (k & 31)] = u[q];
  }
}

// B^T (7 x 7) = {-2,4,5/2,-5,-1/2,1,0} {0,2,-2,-9/2,1/2,1,0} {0,-2,6,-7/2,-3/2,1,0} {0,1,-3/2,-2,3/2,1,0}
//               {0,-1,5/2,0,-5/2,1,0} {0,4,0,-5,0,1,0} {0,-2,4,5/2,-5,-1/2,1}
// A^T (5 x 7) = {1,1,1,1,1,1,0} {0,1,-1,2,-2,1/2,0} {0,1,1,4,4,1/4,0} {0,1,-1,8,-8,1/8,0} {0,1,1,16,16,1/16,1}
// (written for one float; a thread owns ONE channel of a tile: the deep layers have few tiles -- 200 at 25 x 25 and 8 views --
// and two channels per thread, as in the F(4x4) transforms, leave the chip with less than one wave per SIMD)
// (no FMA contraction in the two transforms, as in wg4_bt / wg4_at: every kernel that inlines them rounds alike)
__device__ __forceinline__ void w5_bt(const float* d, float* o) {
#pragma clang fp contract(off)
  o[0] = -2.f * d[0] + 4.f * d[1] + 2.5f * d[2] - 5.f * d[3] - 0.5f * d[4] + d[5];
  o[1] = 2.f * d[1] - 2.f * d[2] - 4.5f * d[3] + 0.5f * d[4] + d[5];
  o[2] = -2.f * d[1] + 6.f * d[2] - 3.5f * d[3] - 1.5f * d[4] + d[5];
  o[3] = d[1] - 1.5f * d[2] - 2.f * d[3] + 1.5f * d[4] + d[5];
  o[4] = -d[1] + 2.5f * d[2] - 2.5f * d[4] + d[5];
  o[5] = 4.f * d[1] - 5.f * d[3] + d[5];
  o[6] = -2.f * d[1] + 4.f * d[2] + 2.5f * d[3] - 5.f * d[4] - 0.5f * d[5] + d[6];
}
__device__ __forceinline__ void w5_at(const float* m, float* o) {
#pragma clang fp contract(off)
  const float s12 = m[1] + m[2], d12 = m[1] - m[2], s34 = m[3] + m[4], d34 = m[3] - m[4];
  o[0] = m[0] + s12 + s34 + m[5];
  o[1] = d12 + 2.f * d34 + 0.5f * m[5];
  o[2] = s12 + 4.f * s34 + 0.25f * m[5];
  o[3] = d12 + 8.f * d34 + 0.125f * m[5];
  o[4] = s12 + 16.f * s34 + 0.0625f * m[5] + m[6];
}

// input transform: one thread = one 7 x 7 patch x 1 channel (a wave covers 64 contiguous channels per pixel); a contiguous
// range of tiles per XCD, as in winograd_input4_kernel.  bits (nullable): two words per (tile, channel PAIR) -- the mask of
// the tile's own 5 x 5 pixels, bit 2 p + (channel & 1); the two lanes of a pair combine theirs with one __shfl_xor.
__global__ void __launch_bounds__(256) winograd5_input_kernel(const float* __restrict__ x, float* __restrict__ V, int B,
                                                              int H, int W, int K, int TH, int TW,
                                                              uint32_t* __restrict__ bits) {
  const int64_t T = (int64_t)B * TH * TW;
  const unsigned per_xcd = gridDim.x / 8;
  const unsigned lb = (blockIdx.x % 8) * per_xcd + blockIdx.x / 8;
  const int64_t gid = (int64_t)lb * blockDim.x + threadIdx.x;
  if (gid >= T * K) return;
  const int c = (int)(gid % K);
  const int64_t tile = gid / K;
  const int tx = (int)(tile % TW), ty = (int)((tile / TW) % TH), b = (int)(tile / ((int64_t)TW * TH));
  const int y0 = 5 * ty - 1, x0 = 5 * tx - 1;
  // all 49 loads go out before anything waits: clamped addresses, the value outside the image dropped by a select
  // (a bounds test around a load is a branch, and hipcc drains vmcnt at the join: three dependent round trips before)
  float dv[7][7];  // dv[s][r]: patch column s, row r
#pragma unroll
  for (int s = 0; s < 7; ++s)
#pragma unroll
    for (int r = 0; r < 7; ++r) {
      const int yc = min(max(y0 + r, 0), H - 1), xc = min(max(x0 + s, 0), W - 1);
      dv[s][r] = x[(((int64_t)b * H + yc) * W + xc) * K + c];
    }
  float t[7][7];   // t[s][r]: column s after the vertical pass
  unsigned long long mask = 0ull;
#pragma unroll
  for (int s = 0; s < 7; ++s) {
    float d[7];
    const int xx = x0 + s;
#pragma unroll
    for (int r = 0; r < 7; ++r) {
      const int yy = y0 + r;
      d[r] = (yy >= 0 && yy < H && xx >= 0 && xx < W) ? dv[s][r] : 0.f;
      if (r >= 1 && r <= 5 && s >= 1 && s <= 5)        // the tile's own pixels (zeros outside the image: bit 0)
        mask |= (unsigned long long)(d[r] > 0.f ? 1u : 0u) << (2 * ((r - 1) * 5 + (s - 1)));
    }
    w5_bt(d, t[s]);
  }
  if (bits) {
    const uint32_t lo = (uint32_t)mask, hi = (uint32_t)(mask >> 32);
    const uint32_t plo = __shfl_xor(lo, 1, 64), phi = __shfl_xor(hi, 1, 64);       // the odd channel of the pair
    if (!(c & 1)) *reinterpret_cast<uint2*>(bits + gid) = make_uint2(lo | (plo << 1), hi | (phi << 1));   // 2 * (gid / 2)
  }
  const int64_t comp_stride = T * K;
  float* vo = V + tile * K + c;
#pragma unroll
  for (int r = 0; r < 7; ++r) {
    const float row[7] = {t[0][r], t[1][r], t[2][r], t[3][r], t[4][r], t[5][r], t[6][r]};
    float o[7];
    w5_bt(row, o);
#pragma unroll
    for (int q = 0; q < 7; ++q) vo[(int64_t)(r * 7 + q) * comp_stride] = o[q];
  }
}

// output transform + layer epilogue: one thread = one 5 x 5 output tile x 1 channel
template <int MODE, int NSPLIT = 1>  // 0: y = relu?(Y + bias); 1: y = (Y [+ addend if relu]) * (x_in > 0) [+ addend if !relu]
__global__ void __launch_bounds__(256) winograd5_output_kernel(const float* __restrict__ M, const float* __restrict__ aux0,
                                                               const float* __restrict__ aux1, float* __restrict__ y,
                                                               int B, int H, int W, int N, int TH, int TW, int relu,
                                                               const uint32_t* __restrict__ bits) {
  // NSPLIT: M holds the products in NSPLIT K parts, 49 * T * N floats apart (winograd_ksplit), summed here (a template
  // parameter: as a run-time loop the 49 loads of a column no longer went out together -- +12 us on a 6-us kernel)
  const int64_t T = (int64_t)B * TH * TW;
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= T * N) return;
  const int c = (int)(gid % N);
  const int64_t tile = gid / N;
  const int tx = (int)(tile % TW), ty = (int)((tile / TW) % TH), b = (int)(tile / ((int64_t)TW * TH));
  const int64_t comp_stride = T * N;
  const float* mi = M + tile * N + c;
  // data gradient: the 25 addend values are requested BEFORE the 49 components (clamped addresses, no branch around a
  // load), so that the two round trips to memory overlap -- the small deep layers are latency-bound here
  float ad[5][5];
  if (MODE == 1) {
    const float* ap = aux1 ? aux1 : M;      // (no addend: the loads still go out -- M is at least as large -- and are ignored)
#pragma unroll
    for (int a = 0; a < 5; ++a)
#pragma unroll
      for (int cc = 0; cc < 5; ++cc) {
        const int yc = min(5 * ty + a, H - 1), xc = min(5 * tx + cc, W - 1);
        ad[a][cc] = ap[(((int64_t)b * H + yc) * W + xc) * N + c];
      }
  }
  float t[7][5];   // t[s][a]: column s after the vertical pass
#pragma unroll
  for (int s = 0; s < 7; ++s) {
    float m[7];
#pragma unroll
    for (int r = 0; r < 7; ++r) {
      m[r] = mi[(int64_t)(r * 7 + s) * comp_stride];
#pragma unroll
      for (int p = 1; p < NSPLIT; ++p) m[r] += mi[((int64_t)p * 49 + r * 7 + s) * comp_stride];
    }
    w5_at(m, t[s]);
  }
  const float bias = (MODE == 0 && aux0) ? aux0[c] : 0.f;
  unsigned long long mask = 0ull;
  if (MODE == 1 && bits) {
    const uint2 mw = *reinterpret_cast<const uint2*>(bits + (gid & ~(int64_t)1));
    mask = (((unsigned long long)mw.y << 32) | mw.x) >> (c & 1);
  }
#pragma unroll
  for (int a = 0; a < 5; ++a) {
    const int yy = 5 * ty + a;
    if (yy >= H) continue;
    const float row[7] = {t[0][a], t[1][a], t[2][a], t[3][a], t[4][a], t[5][a], t[6][a]};
    float o[5];
    w5_at(row, o);
#pragma unroll
    for (int cc = 0; cc < 5; ++cc) {
      const int xx = 5 * tx + cc;
      if (xx >= W) continue;
      float v = o[cc];
      const int64_t idx = (((int64_t)b * H + yy) * W + xx) * N + c;
      if (MODE == 0) {
        v += bias;
        if (relu) v = fmaxf(v, 0.f);
      } else {
        const float adv = aux1 ? ad[a][cc] : 0.f;
        if (relu) v += adv;                      // addend not yet through the mask: add first
        if (bits) v = ((mask >> (2 * (a * 5 + cc))) & 1ull) ? v : 0.f;
        else if (aux0) v = aux0[idx] > 0.f ? v : 0.f;
        if (!relu) v += adv;
      }
      y[idx] = v;
    }
  }
}

// ---- the same two transforms with seven waves per (tile, 64 channels) ---------------------------------------------------
// The one-thread-per-(tile, channel) kernels above put 1.5-3 waves on a SIMD at 8 views (200-800 tiles) and a fraction of
// one at one view, each wave a chain of 49 loads -> 14 seven-point transforms -> 49 stores.  Here a block of 7 waves takes
// one tile x 64 channels: wave s transforms patch column s (7 loads, one transform) into LDS, then wave r transforms row r
// of the result (7 LDS reads, one transform, 7 stores): seven times the waves, a seventh of the chain.  Same sums per value.
__global__ void __launch_bounds__(448) winograd5_input7_kernel(const float* __restrict__ x, float* __restrict__ V, int B,
                                                               int H, int W, int K, int TH, int TW,
                                                               uint32_t* __restrict__ bits) {
  __shared__ float tl[7][7][64];                 // [column s][row r][channel lane] after the vertical pass
  __shared__ unsigned long long mk[5][64];       // the mask bits of patch columns 1..5
  const int64_t T = (int64_t)B * TH * TW;
  const unsigned per_xcd = gridDim.x / 8;
  const unsigned lb = (blockIdx.x % 8) * per_xcd + blockIdx.x / 8;       // a contiguous range of tiles per XCD
  const int kg = K / 64;
  if ((int64_t)lb >= T * kg) return;
  const int64_t tile = lb / kg;
  const int lane = threadIdx.x, w = threadIdx.y, c = (int)(lb % kg) * 64 + lane;
  const int tx = (int)(tile % TW), ty = (int)((tile / TW) % TH), b = (int)(tile / ((int64_t)TW * TH));
  const int y0 = 5 * ty - 1, x0 = 5 * tx - 1;
  {
    const int xx = x0 + w, xc = min(max(xx, 0), W - 1);
    float d[7];
#pragma unroll
    for (int r = 0; r < 7; ++r) {
      const int yc = min(max(y0 + r, 0), H - 1);
      d[r] = x[(((int64_t)b * H + yc) * W + xc) * K + c];
    }
    unsigned long long mask = 0ull;
#pragma unroll
    for (int r = 0; r < 7; ++r) {
      const int yy = y0 + r;
      d[r] = (yy >= 0 && yy < H && xx >= 0 && xx < W) ? d[r] : 0.f;
      if (r >= 1 && r <= 5) mask |= (unsigned long long)(d[r] > 0.f ? 1u : 0u) << (2 * ((r - 1) * 5));
    }
    if (bits && w >= 1 && w <= 5) mk[w - 1][lane] = mask << (2 * (w - 1));
    float t[7];
    w5_bt(d, t);
#pragma unroll
    for (int r = 0; r < 7; ++r) tl[w][r][lane] = t[r];
  }
  __syncthreads();
  const int64_t gid = tile * K + c;
  if (bits && w == 6) {
    const unsigned long long mask = mk[0][lane] | mk[1][lane] | mk[2][lane] | mk[3][lane] | mk[4][lane];
    const uint32_t lo = (uint32_t)mask, hi = (uint32_t)(mask >> 32);
    const uint32_t plo = __shfl_xor(lo, 1, 64), phi = __shfl_xor(hi, 1, 64);
    if (!(c & 1)) *reinterpret_cast<uint2*>(bits + gid) = make_uint2(lo | (plo << 1), hi | (phi << 1));
  }
  const float row[7] = {tl[0][w][lane], tl[1][w][lane], tl[2][w][lane], tl[3][w][lane], tl[4][w][lane], tl[5][w][lane], tl[6][w][lane]};
  float o[7];
  w5_bt(row, o);
  const int64_t comp_stride = T * K;
  float* vo = V + gid + (int64_t)(w * 7) * comp_stride;
#pragma unroll
  for (int q = 0; q < 7; ++q) vo[(int64_t)q * comp_stride] = o[q];
}

template <int MODE, int NSPLIT>
__global__ void __launch_bounds__(448) winograd5_output7_kernel(const float* __restrict__ M, const float* __restrict__ aux0,
                                                                const float* __restrict__ aux1, float* __restrict__ y,
                                                                int B, int H, int W, int N, int TH, int TW, int relu,
                                                                const uint32_t* __restrict__ bits) {
  __shared__ float tl[7][5][64];                 // [column s][output row a][channel lane]
  const int64_t T = (int64_t)B * TH * TW;
  const int ng = N / 64;
  const int64_t tile = blockIdx.x / ng;
  const int lane = threadIdx.x, w = threadIdx.y, c = (int)(blockIdx.x % ng) * 64 + lane;
  const int tx = (int)(tile % TW), ty = (int)((tile / TW) % TH), b = (int)(tile / ((int64_t)TW * TH));
  const int64_t comp_stride = T * N, gid = tile * N + c;
  // the operands of the second pass go out first (row a = min(w, 4) of the addend: no branch around a load)
  const int a = min(w, 4), yy = 5 * ty + a;
  float ad[5];
  uint2 mw = make_uint2(0u, 0u);
  if (MODE == 1) {
    const float* ap = aux1 ? aux1 : M;
    const int yc = min(yy, H - 1);
#pragma unroll
    for (int cc = 0; cc < 5; ++cc) ad[cc] = ap[(((int64_t)b * H + yc) * W + min(5 * tx + cc, W - 1)) * N + c];
    if (bits) mw = *reinterpret_cast<const uint2*>(bits + (gid & ~(int64_t)1));
  }
  {
    const float* mi = M + gid + (int64_t)w * comp_stride;         // column s = w: components 7 r + s
    float m[7];
#pragma unroll
    for (int r = 0; r < 7; ++r) {
      m[r] = mi[(int64_t)(r * 7) * comp_stride];
#pragma unroll
      for (int p = 1; p < NSPLIT; ++p) m[r] += mi[((int64_t)p * 49 + r * 7) * comp_stride];
    }
    float t[5];
    w5_at(m, t);
#pragma unroll
    for (int q = 0; q < 5; ++q) tl[w][q][lane] = t[q];
  }
  __syncthreads();
  if (w >= 5 || yy >= H) return;
  const float row[7] = {tl[0][a][lane], tl[1][a][lane], tl[2][a][lane], tl[3][a][lane], tl[4][a][lane], tl[5][a][lane], tl[6][a][lane]};
  float o[5];
  w5_at(row, o);
  const float bias = (MODE == 0 && aux0) ? aux0[c] : 0.f;
  const unsigned long long mask = ((((unsigned long long)mw.y << 32) | mw.x) >> (c & 1)) >> (2 * (a * 5));
#pragma unroll
  for (int cc = 0; cc < 5; ++cc) {
    const int xx = 5 * tx + cc;
    if (xx >= W) continue;
    float v = o[cc];
    const int64_t idx = (((int64_t)b * H + yy) * W + xx) * N + c;
    if (MODE == 0) {
      v += bias;
      if (relu) v = fmaxf(v, 0.f);
    } else {
      const float adv = aux1 ? ad[cc] : 0.f;
      if (relu) v += adv;
      if (bits) v = ((mask >> (2 * cc)) & 1ull) ? v : 0.f;
      else if (aux0) v = aux0[idx] > 0.f ? v : 0.f;
      if (!relu) v += adv;
    }
    y[idx] = v;
  }
}

// ---- host side ------------------------------------------------------------------------------------------------
// F(5x5) where it executes at most 0.9 x the GEMM rows of F(4x4) (49 tiles5 vs 36 tiles4) and both channel counts are
// >= 128 (the narrower layers have their own single-kernel path, and their images are large: the padding saved is small)
bool winograd5_channels(int K, int N) {
  static const bool off = [] { const char* e = getenv("NFS_WG5"); return e && atoi(e) == 0; }();
  return !off && K >= 128 && N >= 128 && K % 64 == 0 && N % 64 == 0;
}
bool winograd5_takes(int H, int W, int K, int N) {
  if (!winograd5_channels(K, N) || H < 5 || W < 5) return false;
  const int64_t r5 = (int64_t)49 * ((H + 4) / 5) * ((W + 4) / 5), r4 = (int64_t)36 * ((H + 3) / 4) * ((W + 3) / 4);
  return r5 * 10 <= r4 * 9;
}

// 49 + 49 floats per (ci, co) and the limb planes of the second 49 (1.5 x: 73.5)
int64_t winograd5_packed_floats(int Ci, int Co) { return winograd5_channels(Ci, Co) ? (int64_t)98 * Ci * Co + (int64_t)147 * Ci * Co / 2 : 0; }

int64_t winograd5_workspace_floats(int B, int H, int W, int K, int N) {
  if (!winograd5_takes(H, W, K, N)) return 0;
  const int64_t T = (int64_t)B * ((H + 4) / 5) * ((W + 4) / 5);
  return (int64_t)49 * T * ((int64_t)K + (int64_t)N * winograd_ksplit(T, K));     // (M once per K part of the GEMM)
}

// up5: [49][K/32][N][32] followed by the 16x16x4 fragment order of the same
int winograd5_pack(const float* w_hwio, float* up5, int Ci, int Co, int kind, hipStream_t s) {
  const int64_t n = (int64_t)Ci * Co;
  const int Kc = kind == 0 ? Ci : Co, Nc = kind == 0 ? Co : Ci;
  hipLaunchKernelGGL(winograd5_pack_kernel, dim3(blocks_for(n, 256)), dim3(256), 0, s, w_hwio, up5, Ci, Co, kind);
  winograd_pack_frag16(up5, up5 + 49 * n, Kc, Nc, 49 * n, s);
  winograd_pack_limbs16(up5 + 49 * n, up5 + 98 * n, Kc, Nc, 49, s);
  return check_launch("winograd5_pack");
}

// x [B,H,W,K] -> y [B,H,W,N]; U5 from winograd5_pack; ws >= winograd5_workspace_floats.  mode / relu / aux as winograd_conv.
int64_t winograd5_bits_words(int B, int H, int W, int C) { return (int64_t)B * ((H + 4) / 5) * ((W + 4) / 5) * C; }

int winograd5_conv(const float* x, const float* U5, const float* aux0, const float* aux1, float* y, float* ws, int B,
                   int H, int W, int K, int N, int mode, int relu, int cus, hipStream_t s, uint32_t* in_bits) {
  const int TH = (H + 4) / 5, TW = (W + 4) / 5;
  const int64_t T = (int64_t)B * TH * TW;
  float* V = ws;
  float* M = ws + 49 * T * K;
  // the seven-wave transforms where a launch has at most this many (tile, channel) items (NFS_W5_WAVES7_MAX; 0: never)
  static const int64_t waves7_max = [] { const char* e = getenv("NFS_W5_WAVES7_MAX"); return e ? atoll(e) : (int64_t)65536; }();
  // forward: record the mask of x (the layer's own data gradient reads it); data gradient: read the mask of x_in
  if (T * K <= waves7_max) {
    const dim3 ig((unsigned)((T * (K / 64) + 7) / 8 * 8));
    hipLaunchKernelGGL(winograd5_input7_kernel, ig, dim3(64, 7), 0, s, x, V, B, H, W, K, TH, TW, mode == 0 ? in_bits : nullptr);
  } else {
    const dim3 ig((blocks_for(T * K, 256) + 7) / 8 * 8);
    hipLaunchKernelGGL(winograd5_input_kernel, ig, dim3(256), 0, s, x, V, B, H, W, K, TH, TW,
                       mode == 0 ? in_bits : nullptr);
  }
  WgGemmArgs a{V, U5, M, T, K, N, (int64_t)K * N, (int64_t)N * 32, 32, 1.f, nullptr, nullptr};
  a.Uq16 = U5 + (int64_t)49 * K * N;
  a.Ub16 = U5 + (int64_t)98 * K * N;
  const int nsplit = winograd_launch_batched_gemm(a, 49, cus, s);
  const unsigned ob = blocks_for(T * N, 256);
  if (nsplit != 1 && nsplit != 2) {
    set_error("winograd5_conv: unsupported number of K parts");
    return NFS_EINVAL;
  }
  const uint32_t* ib = aux0 ? in_bits : nullptr;
  if (T * N <= waves7_max) {
    const dim3 og((unsigned)(T * (N / 64)));
#define NFS_W5_OUT7(MODE_, NS_, BITS_)                                                                                \
    hipLaunchKernelGGL((winograd5_output7_kernel<MODE_, NS_>), og, dim3(64, 7), 0, s, M, aux0, aux1, y, B, H, W, N, TH, TW,  \
                       relu, BITS_)
    if (mode == 0 && nsplit == 1) NFS_W5_OUT7(0, 1, (const uint32_t*)nullptr);
    else if (mode == 0) NFS_W5_OUT7(0, 2, (const uint32_t*)nullptr);
    else if (nsplit == 1) NFS_W5_OUT7(1, 1, ib);
    else NFS_W5_OUT7(1, 2, ib);
#undef NFS_W5_OUT7
    return check_launch("winograd5_conv");
  }
  if (mode == 0 && nsplit == 1)
    hipLaunchKernelGGL((winograd5_output_kernel<0, 1>), dim3(ob), dim3(256), 0, s, M, aux0, aux1, y, B, H, W, N, TH, TW, relu,
                       (const uint32_t*)nullptr);
  else if (mode == 0)
    hipLaunchKernelGGL((winograd5_output_kernel<0, 2>), dim3(ob), dim3(256), 0, s, M, aux0, aux1, y, B, H, W, N, TH, TW, relu,
                       (const uint32_t*)nullptr);
  else if (nsplit == 1)
    hipLaunchKernelGGL((winograd5_output_kernel<1, 1>), dim3(ob), dim3(256), 0, s, M, aux0, aux1, y, B, H, W, N, TH, TW, relu, ib);
  else
    hipLaunchKernelGGL((winograd5_output_kernel<1, 2>), dim3(ob), dim3(256), 0, s, M, aux0, aux1, y, B, H, W, N, TH, TW, relu, ib);
  return check_launch("winograd5_conv");
}

}  // namespace nfs
