// A6: VGG-19 3x3 SAME convolution (+bias+ReLU) and its data gradient as an implicit
// GEMM on the gfx950 f32 MFMA (v_mfma_f32_32x32x2_f32: exact f32, 157 TFLOP/s peak,
// the same peak as the f32 VALU but reachable with one wave per SIMD), vgg.py:44-48,89-108.
//
// GEMM view (NHWC):  M = output pixels, N = output channels, K = 9 taps x input channels.
//   block  = 256 threads = 4 waves (2 M x 2 N), tile 128 pixel slots x BN channels
//   M tile = TH x TW spatial patch of the row-stacked batch (rows of all images stacked;
//            TH*TW <= 128, any TW: MFMA rows just enumerate the patch's pixels), so the
//            (TH+2)x(TW+2) halo patch is staged in LDS ONCE per 32-channel chunk and
//            re-used by all 9 taps (9x fewer global reads than im2col);
//   K loop = 32-channel chunks x 9 taps; per tap the [BN][32] weight slab is staged
//            through registers into a double-buffered LDS tile (prefetch of tap t+1 is
//            in flight while tap t is multiplied);
//   LDS rows are padded to 36 floats so the ds_read_b128 operand fetches are
//   bank-conflict-free (36*n mod 64 hits 16 distinct 16-B slots per 16-lane group).
//   Each lane fetches 4 consecutive k (one b128) and feeds 4 MFMA steps with them: the
//   k order inside a chunk is permuted identically for A and B, which a sum permits.
// Weights are frozen, so they are packed once on the device into [chunk][tap][N][32].
#include "common.h"

#include <mutex>
#include <stdlib.h>

namespace nfs {

// winograd.hip
int64_t winograd_workspace_floats(int B, int H, int W, int K, int N);
int64_t winograd_packed_floats(int Ci, int Co);
int winograd_path(int B, int H, int W, int K, int N);
int winograd_tile();
int64_t winograd5_bits_words(int B, int H, int W, int C);
int winograd_pack(const float* w_hwio, float* up, int Ci, int Co, int kind, hipStream_t s);
int winograd_conv(const float* x, const float* U, const float* aux0, const float* aux1, float* y, float* ws, int B,
                  int H, int W, int K, int N, int mode, int relu, int cus, hipStream_t s, float* ypool,
                  const float* xmask, uint32_t* in_bits, uint32_t* out_bits, bool pooled_grad);
// layers with >= 64 channels on both sides take the Winograd path (F(4x4,3x3): 4x fewer MFMA flops); only the
// 3-channel conv1_1 stays direct.  NFS_WINOGRAD_MIN_CH raises the threshold (timing comparisons).
static inline bool winograd_eligible(int K, int N) {
  static const int min_ch = [] { const char* e = getenv("NFS_WINOGRAD_MIN_CH"); return e ? atoi(e) : 64; }();
  return K >= min_ch && N >= min_ch && K % 32 == 0 && N % 64 == 0;
}

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int KC = 32;        // channels per K chunk
constexpr int LDS_STRIDE = 36;  // padded row (floats)
constexpr int BM = 128;
constexpr int PATCH_MAX = 192;  // max (TH+2)*(TW+2): 6 pixel rows of 32 per staging pass

struct ConvArgs {
  const float* x;       // [B,H,W,Kc]  input of the GEMM (fwd: activations, dgrad: gy)
  const float* wp;      // packed [Kc/32][9][Nc][32]
  const float* aux0;    // fwd: bias [Nc] (nullable); dgrad: x_in [B,H,W,Nc] for the ReLU mask (nullable)
  const float* aux1;    // dgrad: addend [B,H,W,Nc] (nullable)
  float* y;             // [B,H,W,Nc]
  float* ws;            // split-K partial sums [ksplit][B*H*W][Nc] (ksplit > 1)
  int B, H, W, Kc, Nc;
  int TH, TW, tiles_c;
  int relu, ksplit;
  int dbg;              // timing ablations only (env NFS_CONV_DBG): 1 no fragment reads, 2 no barriers, 4 no W staging
};

// y = epilogue(acc): MODE 0 bias + ReLU, MODE 1 ReLU mask of the layer below + style-gradient addend
template <int MODE>
__device__ __forceinline__ float conv_epilogue(const ConvArgs& a, float v, int n, int64_t idx) {
  if (MODE == 0) {
    if (a.aux0) v += a.aux0[n];
    if (a.relu) v = fmaxf(v, 0.f);
  } else {
    if (a.aux0) v = a.aux0[idx] > 0.f ? v : 0.f;
    if (a.aux1) v += a.aux1[idx];
  }
  return v;
}

template <int BN, int MODE>
__global__ void __launch_bounds__(256, 2) conv3x3_mfma_kernel(ConvArgs a) {
  constexpr int NT = BN / 64;  // 32-wide N tiles per wave
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* patch = smem;                                  // [PATCH_MAX][36]
  float* wl = smem + PATCH_MAX * LDS_STRIDE;            // [2][BN][36]
  // rowpix sits behind max(operand buffers, staged output tile) so that the epilogue tile cannot overwrite it
  constexpr int OPER_FLOATS = PATCH_MAX * LDS_STRIDE + 2 * BN * LDS_STRIDE;
  constexpr int TILE_FLOATS = BM * (BN + 4);
  int* rowpix = reinterpret_cast<int*>(smem + (OPER_FLOATS > TILE_FLOATS ? OPER_FLOATS : TILE_FLOATS));  // [128]

  const int t = threadIdx.x;
  const int lane = t & 63, wid = t >> 6;
  const int wm = wid >> 1, wn = wid & 1;
  const int i = lane & 31, h = lane >> 5;
  const int tr = blockIdx.x / a.tiles_c, tc = blockIdx.x - tr * a.tiles_c;
  const int n0 = blockIdx.y * BN;
  const int TH = a.TH, TW = a.TW, PW = TW + 2;
  const int PP = (TH + 2) * PW;
  const int rows_total = a.B * a.H;
  const int r0 = tr * TH, c0 = tc * TW;

  // row -> global pixel table for the epilogue
  if (t < BM) {
    int pix = -1;
    if (t < TH * TW) {
      const int ty = t / TW, tx = t - ty * TW;
      const int r = r0 + ty, c = c0 + tx;
      if (r < rows_total && c < a.W) pix = r * a.W + c;
    }
    rowpix[t] = pix;
  }

  // per-lane A/B fragment bases
  int abase[2];
  bool up_ok[2], dn_ok[2];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) {
    int m = wm * 64 + mt * 32 + i;
    if (m >= TH * TW) m = 0;
    const int ty = m / TW, tx = m - ty * TW;
    abase[mt] = (ty * PW + tx) * LDS_STRIDE + 4 * h;
    const int yy = (r0 + ty) % a.H;
    up_ok[mt] = yy > 0;           // input row y-1 belongs to the same image
    dn_ok[mt] = yy < a.H - 1;     // input row y+1 belongs to the same image
  }
  int bbase[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) bbase[nt] = (wn * (BN / 2) + nt * 32 + i) * LDS_STRIDE + 4 * h;

  f32x16 acc[2][NT];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;

  // split-K: this block multiplies chunks [cb, ce) of the K dimension
  const int nchunks = a.Kc / KC;
  const int cb = (int)((int64_t)blockIdx.z * nchunks / a.ksplit);
  const int ce = (int)((int64_t)(blockIdx.z + 1) * nchunks / a.ksplit);
  const int it0 = cb * 9, it1 = ce * 9;

  // halo patch: thread t stages float4 #(t&7) of patch pixels (t>>3) + 32 j, j < 6, through named
  // registers (prefetched one chunk = 9 taps ahead).  Offsets are chunk-independent.
  const int q4 = 4 * (t & 7);
  int64_t poff[6];
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    const int pp = (t >> 3) + 32 * j;
    const int pr = pp / PW, pc = pp - pr * PW;
    const int srow = r0 + pr - 1, xc = c0 + pc - 1;
    const bool ok = pp < PP && srow >= 0 && srow < rows_total && xc >= 0 && xc < a.W;
    poff[j] = ok ? ((int64_t)srow * a.W + xc) * a.Kc + q4 : -1;
  }
  float4 p0, p1, p2, p3, p4, p5;
#define NFS_LOAD_PATCH(chunk_)                                                                   \
  {                                                                                              \
    const float* xb = a.x + (chunk_) * KC;                                                       \
    p0 = p1 = p2 = p3 = p4 = p5 = make_float4(0.f, 0.f, 0.f, 0.f);                               \
    if (poff[0] >= 0) p0 = *reinterpret_cast<const float4*>(xb + poff[0]);                       \
    if (poff[1] >= 0) p1 = *reinterpret_cast<const float4*>(xb + poff[1]);                       \
    if (poff[2] >= 0) p2 = *reinterpret_cast<const float4*>(xb + poff[2]);                       \
    if (poff[3] >= 0) p3 = *reinterpret_cast<const float4*>(xb + poff[3]);                       \
    if (poff[4] >= 0) p4 = *reinterpret_cast<const float4*>(xb + poff[4]);                       \
    if (poff[5] >= 0) p5 = *reinterpret_cast<const float4*>(xb + poff[5]);                       \
  }
  NFS_LOAD_PATCH(cb)

  // weight slab (chunk, tap) = it: rows n0..n0+BN of [Nc][32]; thread t owns float4 #(t + 256 r).
  // Named registers (not an array): an indexed private array here ends up in scratch memory.
  constexpr int WREG = BN / 32;
  const float4* wp4 = reinterpret_cast<const float4*>(a.wp) + (int64_t)n0 * (KC / 4) + t;
  const int64_t slab4 = (int64_t)a.Nc * (KC / 4);
  float4 w0, w1, w2, w3;
  {
    const float4* wn = wp4 + (int64_t)it0 * slab4;
    w0 = wn[0]; w1 = wn[256];
    if (WREG > 2) { w2 = wn[512]; w3 = wn[768]; }
  }

  float af0[4] = {0, 0, 0, 0}, af1[4] = {0, 0, 0, 0}, bf[NT][4] = {};
  for (int it = it0; it < it1; ++it) {
    const int chunk = it / 9, tap = it - chunk * 9;
    if (tap == 0) {
      if (!NFS_DBG(a, 2)) __syncthreads();  // every wave is done with the previous chunk's patch
      float* pdst = patch + (t >> 3) * LDS_STRIDE + q4;
      *reinterpret_cast<float4*>(pdst) = p0;
      *reinterpret_cast<float4*>(pdst + 32 * LDS_STRIDE) = p1;
      *reinterpret_cast<float4*>(pdst + 64 * LDS_STRIDE) = p2;
      *reinterpret_cast<float4*>(pdst + 96 * LDS_STRIDE) = p3;
      *reinterpret_cast<float4*>(pdst + 128 * LDS_STRIDE) = p4;
      *reinterpret_cast<float4*>(pdst + 160 * LDS_STRIDE) = p5;
    }
    float* wcur = wl + (it & 1) * BN * LDS_STRIDE;
    if (!NFS_DBG(a, 4) || it == it0) {
      float* wdst = wcur + (t >> 3) * LDS_STRIDE + q4;   // f = t + 256 r -> row (f>>3) = (t>>3) + 32 r
      *reinterpret_cast<float4*>(wdst) = w0;
      *reinterpret_cast<float4*>(wdst + 32 * LDS_STRIDE) = w1;
      if (WREG > 2) {
        *reinterpret_cast<float4*>(wdst + 64 * LDS_STRIDE) = w2;
        *reinterpret_cast<float4*>(wdst + 96 * LDS_STRIDE) = w3;
      }
    }
    if (!NFS_DBG(a, 2)) __syncthreads();
    if (it + 1 < it1 && !NFS_DBG(a, 4)) {
      const float4* wn = wp4 + (int64_t)(it + 1) * slab4;
      w0 = wn[0]; w1 = wn[256];
      if (WREG > 2) { w2 = wn[512]; w3 = wn[768]; }
    }
    if (tap == 0 && chunk + 1 < ce) NFS_LOAD_PATCH(chunk + 1)   // consumed 9 taps later

    const int dy = tap / 3, dx = tap - dy * 3;
    const int tapoff = (dy * PW + dx) * LDS_STRIDE;
    const bool z0 = (dy == 0 && !up_ok[0]) || (dy == 2 && !dn_ok[0]);
    const bool z1 = (dy == 0 && !up_ok[1]) || (dy == 2 && !dn_ok[1]);
#pragma unroll
    for (int c = 0; c < KC / 8; ++c) {
      if (!NFS_DBG(a, 1) || it == it0) {
      float4 a0 = *reinterpret_cast<const float4*>(patch + abase[0] + tapoff + 8 * c);
      float4 a1 = *reinterpret_cast<const float4*>(patch + abase[1] + tapoff + 8 * c);
      if (z0) a0 = make_float4(0.f, 0.f, 0.f, 0.f);
      if (z1) a1 = make_float4(0.f, 0.f, 0.f, 0.f);
      af0[0] = a0.x; af0[1] = a0.y; af0[2] = a0.z; af0[3] = a0.w;
      af1[0] = a1.x; af1[1] = a1.y; af1[2] = a1.z; af1[3] = a1.w;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const float4 bq = *reinterpret_cast<const float4*>(wcur + bbase[nt] + 8 * c);
        bf[nt][0] = bq.x; bf[nt][1] = bq.y; bf[nt][2] = bq.z; bf[nt][3] = bq.w;
      }
      }
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          acc[0][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(af0[jj], bf[nt][jj], acc[0][nt], 0, 0, 0);
          acc[1][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(af1[jj], bf[nt][jj], acc[1][nt], 0, 0, 0);
        }
      }
    }
  }
#undef NFS_LOAD_PATCH

  // epilogue.  The C/D layout (col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)) gives 4-byte stores at a
  // row stride: measured store-issue-bound (~1.1 TB/s, up to 70 us fixed per launch).  So the tile is transposed
  // through LDS (the operand buffers are free now) and leaves as float4 rows: 8x fewer, 16-byte store instructions.
  constexpr int OS = BN + 4;                               // padded row stride of the staged tile
  float* otile = smem;                                     // [128][OS] aliases patch + weight buffers
  __syncthreads();                                         // every wave is done reading operands
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = wm * 64 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) otile[row * OS + wn * (BN / 2) + nt * 32 + i] = acc[mt][nt][r];
    }
  __syncthreads();
  float* outp = a.ksplit > 1 ? a.ws + (int64_t)blockIdx.z * a.B * a.H * a.W * a.Nc : a.y;
  constexpr int Q = BN / 4;                                // float4 per row
#pragma unroll
  for (int e = 0; e < (BM * Q) / 256; ++e) {
    const int f = t + 256 * e;
    const int row = f / Q, q = f - row * Q;
    const int pix = rowpix[row];
    if (pix < 0) continue;
    float4 v = *reinterpret_cast<const float4*>(otile + row * OS + 4 * q);
    const int n = n0 + 4 * q;
    const int64_t idx = (int64_t)pix * a.Nc + n;
    if (a.ksplit == 1) {
      if (MODE == 0) {
        if (a.aux0) {
          const float4 bb = *reinterpret_cast<const float4*>(a.aux0 + n);
          v.x += bb.x; v.y += bb.y; v.z += bb.z; v.w += bb.w;
        }
        if (a.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
      } else {
        if (a.aux0) {
          const float4 xin = *reinterpret_cast<const float4*>(a.aux0 + idx);
          v.x = xin.x > 0.f ? v.x : 0.f; v.y = xin.y > 0.f ? v.y : 0.f;
          v.z = xin.z > 0.f ? v.z : 0.f; v.w = xin.w > 0.f ? v.w : 0.f;
        }
        if (a.aux1) {
          const float4 ad = *reinterpret_cast<const float4*>(a.aux1 + idx);
          v.x += ad.x; v.y += ad.y; v.z += ad.z; v.w += ad.w;
        }
      }
    }
    *reinterpret_cast<float4*>(outp + idx) = v;
  }
}

// split-K second pass: y = epilogue(sum_s ws[s]); fixed summation order => deterministic
template <int MODE>
__global__ void __launch_bounds__(256) conv_reduce_kernel(ConvArgs a, int64_t mn) {
  const int64_t i4 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (i4 >= mn) return;
  float4 s = *reinterpret_cast<const float4*>(a.ws + i4);
  for (int k = 1; k < a.ksplit; ++k) {
    const float4 v = *reinterpret_cast<const float4*>(a.ws + (int64_t)k * mn + i4);
    s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
  }
  const int n = (int)(i4 % a.Nc);
  float4 o;
  o.x = conv_epilogue<MODE>(a, s.x, n, i4);
  o.y = conv_epilogue<MODE>(a, s.y, n + 1, i4 + 1);
  o.z = conv_epilogue<MODE>(a, s.z, n + 2, i4 + 2);
  o.w = conv_epilogue<MODE>(a, s.w, n + 3, i4 + 3);
  *reinterpret_cast<float4*>(a.y + i4) = o;
}

// ---- first layer (Ci = 3): direct VALU kernels; 0.5 % of the FLOPs -------------------------
// fwd: thread = (pixel, 4 consecutive output channels); weights HWIO [9][3][Co] in LDS.
// ---- conv1_1 (Ci = 3, Co = 64) on the f32 MFMA ------------------------------------------------------------------
// Implicit GEMM with M = an 8 x 16 pixel tile, N = 64, K = 27 (14 k-steps of 2, the last half-step zero): the
// 3-channel halo patch is tiny (540 floats) and lives in LDS, a lane builds its A operand by indexing it, the 27 x 64
// weights sit in 28 registers per lane; the tile leaves through LDS as float4 rows.  The VALU version (one thread per
// output float4, 27 global loads + 108 FMAs) was bound by vector-memory issue at 55 us for an 82 MB output.
constexpr int C3F_TH = 8, C3F_TW = 16, C3F_PW = C3F_TW + 2, C3F_OS = 68;
__global__ void __launch_bounds__(256) conv3x3_c3co64_fwd_kernel(const float* __restrict__ x,
                                                                 const float* __restrict__ w,      // [27][64]
                                                                 const float* __restrict__ bias,
                                                                 float* __restrict__ y, int B, int H, int W, int relu,
                                                                 int tiles_x, int tiles_y) {
  __shared__ float patch[(C3F_TH + 2) * C3F_PW * 3 + 4];
  __shared__ __attribute__((aligned(16))) float otile[C3F_TH * C3F_TW * C3F_OS];
  const int t = threadIdx.x, lane = t & 63, wid = t >> 6, i = lane & 31, h = lane >> 5;
  float bfr[14][2];                                  // the lane's 28 weights: gathered once, used for every tile it walks
#pragma unroll
  for (int s2 = 0; s2 < 14; ++s2) {
    const int k = 2 * s2 + h;
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) bfr[s2][nt] = k < 27 ? w[k * 64 + nt * 32 + i] : 0.f;
  }
  float bv2[2];
#pragma unroll
  for (int nt = 0; nt < 2; ++nt) bv2[nt] = bias ? bias[nt * 32 + i] : 0.f;
  const int ntiles_all = B * tiles_x * tiles_y;
  for (int tile_id = blockIdx.x; tile_id < ntiles_all; tile_id += gridDim.x) {
  const int tx = tile_id % tiles_x, ty = (tile_id / tiles_x) % tiles_y, b = tile_id / (tiles_x * tiles_y);
  const int y0 = ty * C3F_TH, x0 = tx * C3F_TW;
  for (int e = t; e < (C3F_TH + 2) * C3F_PW * 3; e += 256) {
    const int r = e / (C3F_PW * 3), rem = e - r * (C3F_PW * 3);
    const int c = rem / 3, ci = rem - c * 3;
    const int gy_ = y0 - 1 + r, gx_ = x0 - 1 + c;
    float v = 0.f;
    if (gy_ >= 0 && gy_ < H && gx_ >= 0 && gx_ < W) v = x[(((int64_t)b * H + gy_) * W + gx_) * 3 + ci];
    patch[e] = v;
  }
  __syncthreads();
  const int m = wid * 32 + i;                       // pixel of this lane's A row
  const int base = ((m / C3F_TW) * C3F_PW + (m % C3F_TW)) * 3;
  f32x16 acc[2];
#pragma unroll
  for (int nt = 0; nt < 2; ++nt)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[nt][r] = 0.f;
#pragma unroll
  for (int s2 = 0; s2 < 14; ++s2) {
    const int k = 2 * s2 + h;                       // k = (ky*3 + kx)*3 + ci
    const int ky = k / 9, kx = (k / 3) % 3, ci = k % 3;
    const float av = k < 27 ? patch[base + (ky * C3F_PW + kx) * 3 + ci] : 0.f;
    acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bfr[s2][0], acc[0], 0, 0, 0);
    acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bfr[s2][1], acc[1], 0, 0, 0);
  }
#pragma unroll
  for (int nt = 0; nt < 2; ++nt) {
    const float bv = bv2[nt];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = wid * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
      float v = acc[nt][r] + bv;
      if (relu) v = fmaxf(v, 0.f);
      otile[row * C3F_OS + nt * 32 + i] = v;
    }
  }
  __syncthreads();
  for (int f = t; f < C3F_TH * C3F_TW * 16; f += 256) {
    const int row = f >> 4, q = f & 15;
    const int gy_ = y0 + row / C3F_TW, gx_ = x0 + row % C3F_TW;
    if (gy_ < H && gx_ < W)
      *reinterpret_cast<float4*>(y + (((int64_t)b * H + gy_) * W + gx_) * 64 + 4 * q) =
          *reinterpret_cast<const float4*>(otile + row * C3F_OS + 4 * q);
  }
  __syncthreads();                                    // patch and otile are rewritten by the next tile
  }
}

__global__ void __launch_bounds__(256) conv3x3_c3_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                             const float* __restrict__ bias, float* __restrict__ y,
                                                             int B, int H, int W, int Co, int relu) {
  extern __shared__ __attribute__((aligned(16))) float sw[];  // [27][Co]
  for (int k = threadIdx.x; k < 27 * Co; k += blockDim.x) sw[k] = w[k];
  __syncthreads();
  const int CQ = Co >> 2;
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= (int64_t)B * H * W * CQ) return;
  const int cq = (int)(gid % CQ);
  const int64_t pix = gid / CQ;
  const int xx = (int)(pix % W);
  const int yy = (int)((pix / W) % H);
  float4 acc = bias ? *reinterpret_cast<const float4*>(bias + 4 * cq) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int dy = -1; dy <= 1; ++dy) {
    if (yy + dy < 0 || yy + dy >= H) continue;
#pragma unroll
    for (int dx = -1; dx <= 1; ++dx) {
      if (xx + dx < 0 || xx + dx >= W) continue;
      const float* xp = x + (pix + (int64_t)dy * W + dx) * 3;
      const int tap = (dy + 1) * 3 + dx + 1;
#pragma unroll
      for (int ci = 0; ci < 3; ++ci) {
        const float xv = xp[ci];
        const float4 wv = *reinterpret_cast<const float4*>(sw + (tap * 3 + ci) * Co + 4 * cq);
        acc.x += xv * wv.x; acc.y += xv * wv.y; acc.z += xv * wv.z; acc.w += xv * wv.w;
      }
    }
  }
  if (relu) { acc.x = fmaxf(acc.x, 0.f); acc.y = fmaxf(acc.y, 0.f); acc.z = fmaxf(acc.z, 0.f); acc.w = fmaxf(acc.w, 0.f); }
  *reinterpret_cast<float4*>(y + pix * Co + 4 * cq) = acc;
}

// dgrad to the 3-channel image: thread = pixel; packed [tap'][Co][3] (taps pre-flipped) in LDS.
// conv1_1 data gradient (Co = 64 -> 3 channels) on the f32 MFMA, in two phases inside one block:
//   T[q][tap*3+ci] = sum_co gy[q][co] * wd[tap][co][ci]     GEMM, M = the 10 x 18 halo'd pixels of an 8 x 16 tile
//                                                           (192 rows), K = 64, N = 27 (-> 32)
//   gx[p][ci]      = sum_tap T[p + d(tap)][tap*3+ci]        9-tap gather of T from LDS
// A fragments come straight from global memory as float4 (K permuted so that a lane owns 4 consecutive channels
// for 4 consecutive k-steps: lane (i, h) of step 4j+u multiplies channel 8j + 4h + u), the 64 x 27 weights sit
// in 32 registers per lane.  The VALU version (one thread per pixel, 144 float4 loads at a 256-byte lane stride)
// took 118 us for an 82 MB input.
// (round 4: 14 x 14 pixel tiles -- their 16 x 16 halo'd pixels are exactly 8 M tiles, two per wave; the 8 x 16 tiles before
// had 6, i.e. two waves with twice the MFMA chain of the others, and a halo of 1.5 x instead of 1.31 x.  The kernel is
// bound by its 32-deep dependent MFMA chains, not by HBM: 2.0 GFLOP executed for 8 x 200 x 200)
constexpr int C3D_TH = 14, C3D_TW = 14, C3D_PH = C3D_TH + 2, C3D_PW = C3D_TW + 2, C3D_M = 256, C3D_TS = 33;
__global__ void __launch_bounds__(256) conv3x3_c3co64_dgrad_kernel(const float* __restrict__ gy,
                                                                   const float* __restrict__ wd,   // [9][64][3]
                                                                   float* __restrict__ gx, int B, int H, int W,
                                                                   int tiles_x, int tiles_y) {
  __shared__ float Tt[C3D_M * C3D_TS];
  const int t = threadIdx.x, lane = t & 63, wid = t >> 6, i = lane & 31, h = lane >> 5;
  // B operand: column n = i (tap = n / 3, ci = n % 3; columns 27..31 are zero), rows k = 8j + 4h + u
  float bw[8][4];
#pragma unroll
  for (int j = 0; j < 8; ++j)
#pragma unroll
    for (int u = 0; u < 4; ++u) bw[j][u] = i < 27 ? wd[((i / 3) * 64 + 8 * j + 4 * h + u) * 3 + i % 3] : 0.f;
  // a block walks several pixel tiles with the 32 weight registers it gathered once (they were two thirds of a one-tile
  // block's vector-memory instructions)
  const int ntiles_all = B * tiles_x * tiles_y;
  for (int tile_id = blockIdx.x; tile_id < ntiles_all; tile_id += gridDim.x) {
  const int tx = tile_id % tiles_x, ty = (tile_id / tiles_x) % tiles_y, b = tile_id / (tiles_x * tiles_y);
  const int y0 = ty * C3D_TH, x0 = tx * C3D_TW;
  // M tiles of 32 halo'd pixels: two per wave (8 tiles = the 256 rows of the 16 x 16 halo'd tile)
  constexpr int ntile = C3D_PH * C3D_PW / 128;
  static_assert(ntile * 128 == C3D_M && C3D_PH * C3D_PW == C3D_M, "the halo'd tile must fill whole M tiles");
  const int tile0 = ntile * wid;
  for (int mt = 0; mt < ntile; ++mt) {
    // (both M tiles' loads up front and their MFMA chains interleaved was measured: 36.5 us against 34.1 for this form)
    const int q = (tile0 + mt) * 32 + i;                         // halo'd pixel index of this lane's A row
    const int qy = y0 - 1 + q / C3D_PW, qx = x0 - 1 + q % C3D_PW;
    const bool ok = qy >= 0 && qy < H && qx >= 0 && qx < W;
    const float* gp = gy + (((int64_t)b * H + (ok ? qy : 0)) * W + (ok ? qx : 0)) * 64 + 4 * h;
    float4 av[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) av[j] = ok ? *reinterpret_cast<const float4*>(gp + 8 * j) : make_float4(0.f, 0.f, 0.f, 0.f);
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[j].x, bw[j][0], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[j].y, bw[j][1], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[j].z, bw[j][2], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[j].w, bw[j][3], acc, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = (tile0 + mt) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
      Tt[row * C3D_TS + i] = acc[r];
    }
  }
  __syncthreads();
  if (t < C3D_TH * C3D_TW) {
    const int ly = t / C3D_TW, lx = t % C3D_TW;
    const int py = y0 + ly, px = x0 + lx;
    if (py < H && px < W) {
      float o0 = 0.f, o1 = 0.f, o2 = 0.f;
#pragma unroll
      for (int dy = 0; dy < 3; ++dy)
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
          const float* tp = Tt + ((ly + dy) * C3D_PW + lx + dx) * C3D_TS + (dy * 3 + dx) * 3;
          o0 += tp[0]; o1 += tp[1]; o2 += tp[2];
        }
      float* o = gx + (((int64_t)b * H + py) * W + px) * 3;
      o[0] = o0; o[1] = o1; o[2] = o2;
    }
  }
  __syncthreads();                                               // Tt is rewritten by the next tile
  }
}

__global__ void __launch_bounds__(256) conv3x3_c3_dgrad_kernel(const float* __restrict__ gy,
                                                               const float* __restrict__ wp, float* __restrict__ gx,
                                                               int B, int H, int W, int Co) {
  extern __shared__ __attribute__((aligned(16))) float sw[];  // [9][Co][3]
  for (int k = threadIdx.x; k < 27 * Co; k += blockDim.x) sw[k] = wp[k];
  __syncthreads();
  const int64_t pix = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (pix >= (int64_t)B * H * W) return;
  const int xx = (int)(pix % W);
  const int yy = (int)((pix / W) % H);
  float a0 = 0.f, a1 = 0.f, a2 = 0.f;
  for (int dy = -1; dy <= 1; ++dy) {
    if (yy + dy < 0 || yy + dy >= H) continue;
    for (int dx = -1; dx <= 1; ++dx) {
      if (xx + dx < 0 || xx + dx >= W) continue;
      const float4* gp = reinterpret_cast<const float4*>(gy + (pix + (int64_t)dy * W + dx) * Co);
      const float* ws = sw + ((dy + 1) * 3 + dx + 1) * Co * 3;
      for (int c4 = 0; c4 < (Co >> 2); ++c4) {
        const float4 g = gp[c4];
        const float* wq = ws + c4 * 12;
        a0 += g.x * wq[0] + g.y * wq[3] + g.z * wq[6] + g.w * wq[9];
        a1 += g.x * wq[1] + g.y * wq[4] + g.z * wq[7] + g.w * wq[10];
        a2 += g.x * wq[2] + g.y * wq[5] + g.z * wq[8] + g.w * wq[11];
      }
    }
  }
  gx[pix * 3] = a0; gx[pix * 3 + 1] = a1; gx[pix * 3 + 2] = a2;
}

// packing: kind 0: Wp[chunk][tap][n=co][kc] = w[tap][chunk*32+kc][co]
//          kind 1: Wp[chunk][tap][n=ci][kc] = w[8-tap][ci][chunk*32+kc]
__global__ void __launch_bounds__(256) pack_kernel(const float* __restrict__ w, float* __restrict__ wp, int Ci, int Co,
                                                   int kind) {
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= (int64_t)9 * Ci * Co) return;
  if (Ci == 3) {
    if (kind == 0) { wp[gid] = w[gid]; return; }
    // [tap'][co][ci] = w[8-tap'][ci][co]
    const int ci = (int)(gid % 3);
    const int co = (int)((gid / 3) % Co);
    const int tp = (int)(gid / (3 * Co));
    wp[gid] = w[((int64_t)(8 - tp) * 3 + ci) * Co + co];
    return;
  }
  const int Nc = kind == 0 ? Co : Ci;
  const int kc = (int)(gid % KC);
  const int n = (int)((gid / KC) % Nc);
  const int tap = (int)((gid / ((int64_t)KC * Nc)) % 9);
  const int chunk = (int)(gid / ((int64_t)KC * Nc * 9));
  float v;
  if (kind == 0) v = w[((int64_t)tap * Ci + chunk * KC + kc) * Co + n];
  else v = w[((int64_t)(8 - tap) * Ci + n) * Co + chunk * KC + kc];
  wp[gid] = v;
}

// choose the spatial M tile: maximise useful pixel slots, prefer wide tiles
static void pick_tile(int B, int H, int W, int& TH, int& TW, int& tiles_r, int& tiles_c) {
  double best = -1.0;
  const int rows = B * H;
  for (int tw = 4; tw <= 128 && tw <= ((W + 3) / 4) * 4; ++tw) {
    const int twc = tw > W ? W : tw;
    int th = BM / twc;
    if (th > rows) th = rows;
    if (th < 1) continue;
    while (th > 1 && (th + 2) * (twc + 2) > PATCH_MAX) --th;
    if ((th + 2) * (twc + 2) > PATCH_MAX) continue;
    const int tr = (rows + th - 1) / th, tcn = (W + twc - 1) / twc;
    const double eff = (double)rows * W / ((double)tr * tcn * BM);
    if (eff > best + 1e-9) { best = eff; TH = th; TW = twc; tiles_r = tr; tiles_c = tcn; }
  }
}

static int g_cus = 0;
static int device_cus() {
  if (g_cus == 0) {
    int c = nfs_device_cus();
    g_cus = c > 0 ? c : 256;
  }
  return g_cus;
}

// Tile-count model.  A launch is `rounds` waves of blocks over the CU slots; few, equal-sized
// blocks quantise badly (160 blocks on 512 slots = 31 % of the chip), so the K dimension is split
// until the grid is several rounds deep; the partial sums cost one extra pass over M x N.
struct ConvPlan { int bn, ksplit; };
static ConvPlan plan_conv(int mtiles, int Nc, int nchunks, int64_t mn, int64_t ws_floats) {
  const int cus = device_cus();
  ConvPlan best{64, 1};
  double best_t = 1e30;
  for (int bn = 64; bn <= 128; bn += 64) {
    if (Nc % bn) continue;
    const int slots = cus * (bn == 64 ? 3 : 2);        // blocks resident per CU (LDS-limited)
    const double unit = bn == 64 ? 0.55 : 1.0;         // relative time of one (chunk,tap) step
    for (int ks = 1; ks <= nchunks && ks <= 16; ++ks) {
      if (ks > 1 && (int64_t)ks * mn > ws_floats) break;
      const int64_t blocks = (int64_t)mtiles * (Nc / bn) * ks;
      const double rounds = (double)((blocks + slots - 1) / slots);
      const int steps = ((nchunks + ks - 1) / ks) * 9 + 3;   // +3: pipeline fill / epilogue
      // one step of a 128-wide block = 128*128*32*2 flop at ~1/2.2 of a CU's 614 GF/s share
      double tms = rounds * steps * unit * 1.9e-3 * (slots / (double)cus) / 2.0;
      if (ks > 1) tms += (double)(ks + 1) * mn * 4.0 / 4.0e9 + 2.5e-3;   // partial sums: write + read
      if (tms < best_t) { best_t = tms; best = ConvPlan{bn, ks}; }
    }
  }
  return best;
}

static bool takes_winograd(const ConvArgs& a, const float* ws, int64_t ws_floats) {
  static const bool no_wg = getenv("NFS_NO_WINOGRAD") != nullptr;   // timing comparisons only
  return !no_wg && winograd_eligible(a.Kc, a.Nc) && a.H >= 2 && a.W >= 2 && ws &&
         ws_floats >= winograd_workspace_floats(a.B, a.H, a.W, a.Kc, a.Nc);
}

// the pooled forms are only fused on the F(4x4) Winograd path
static bool takes_fused_pool(const ConvArgs& a, const float* ws, int64_t ws_floats) {
  static const bool tile4 = [] { const char* e = getenv("NFS_WINOGRAD_TILE"); return !(e && atoi(e) == 2); }();
  static const bool off = getenv("NFS_NO_POOL_FUSION") != nullptr;   // timing comparisons only
  return tile4 && !off && takes_winograd(a, ws, ws_floats);
}

template <int MODE>
static int launch_conv(const ConvArgs& base, float* ws, int64_t ws_floats, hipStream_t s, float* ypool = nullptr,
                       const float* xmask = nullptr, uint32_t* in_bits = nullptr, uint32_t* out_bits = nullptr,
                       bool pooled_grad = false) {
  ConvArgs a = base;
  if (takes_winograd(a, ws, ws_floats)) {
    const float* U = a.wp + (int64_t)9 * a.Kc * a.Nc;               // Winograd weights follow the direct packing
    const bool cache = takes_fused_pool(a, ws, ws_floats);           // the bit cache lives in the F(4x4) transforms
    return winograd_conv(a.x, U, a.aux0, a.aux1, a.y, ws, a.B, a.H, a.W, a.Kc, a.Nc, MODE, a.relu, device_cus(), s,
                         ypool, xmask, cache ? in_bits : nullptr, cache ? out_bits : nullptr, pooled_grad);
  }
  NFS_REQUIRE(!ypool && !xmask && !pooled_grad, "conv3x3: fused pooling needs the Winograd path");
  int tiles_r = 0;
  a.TH = 0;
  pick_tile(a.B, a.H, a.W, a.TH, a.TW, tiles_r, a.tiles_c);
  NFS_REQUIRE(a.TH > 0, "conv3x3: no valid tile for %dx%dx%d", a.B, a.H, a.W);
  const int mtiles = tiles_r * a.tiles_c;
  const int64_t mn = (int64_t)a.B * a.H * a.W * a.Nc;
  const ConvPlan plan = plan_conv(mtiles, a.Nc, a.Kc / KC, mn, ws ? ws_floats : 0);
  a.ksplit = plan.ksplit;
  a.ws = ws;
#ifdef NFS_ABLATE
  static const int dbg = getenv("NFS_CONV_DBG") ? atoi(getenv("NFS_CONV_DBG")) : 0;
  a.dbg = dbg;
#endif
  const dim3 grid(mtiles, a.Nc / plan.bn, plan.ksplit);
  if (plan.bn == 128) {
    constexpr int BN = 128;
    const size_t oper = PATCH_MAX * LDS_STRIDE + 2 * BN * LDS_STRIDE, tile = BM * (BN + 4);
    const size_t lds = (oper > tile ? oper : tile) * sizeof(float) + BM * sizeof(int);
    static std::once_flag attr_once;   // (one set per kernel instance, safe from several host threads)
    std::call_once(attr_once, [&] {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_mfma_kernel<BN, MODE>),
                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); });
    hipLaunchKernelGGL((conv3x3_mfma_kernel<BN, MODE>), grid, dim3(256), lds, s, a);
  } else {
    constexpr int BN = 64;
    const size_t oper = PATCH_MAX * LDS_STRIDE + 2 * BN * LDS_STRIDE, tile = BM * (BN + 4);
    const size_t lds = (oper > tile ? oper : tile) * sizeof(float) + BM * sizeof(int);
    hipLaunchKernelGGL((conv3x3_mfma_kernel<BN, MODE>), grid, dim3(256), lds, s, a);
  }
  if (plan.ksplit > 1)
    hipLaunchKernelGGL(conv_reduce_kernel<MODE>, dim3(blocks_for(mn / 4, 256)), dim3(256), 0, s, a, mn);
  return check_launch(MODE == 0 ? "nfs_conv3x3_fwd" : "nfs_conv3x3_dgrad");
}

}  // namespace nfs

using namespace nfs;

extern "C" {

int64_t nfs_conv3x3_packed_floats(int Ci, int Co, int kind) {
  if (Ci <= 0 || Co <= 0) return 0;
  const int K = kind == 0 ? Ci : Co, N = kind == 0 ? Co : Ci;
  return (int64_t)9 * Ci * Co + (winograd_eligible(K, N) ? winograd_packed_floats(Ci, Co) : 0);
}

int nfs_conv3x3_pack(const float* w_hwio, float* packed, int Ci, int Co, int kind, nfs_stream_t stream) {
  NFS_REQUIRE(w_hwio && packed, "nfs_conv3x3_pack: null pointer");
  NFS_REQUIRE(kind == 0 || kind == 1, "nfs_conv3x3_pack: kind must be 0 or 1");
  NFS_REQUIRE(Ci == 3 || (Ci % 64 == 0), "nfs_conv3x3_pack: Ci must be 3 or a multiple of 64");
  NFS_REQUIRE(Co % 64 == 0, "nfs_conv3x3_pack: Co must be a multiple of 64");
  const int64_t n = (int64_t)9 * Ci * Co;
  hipLaunchKernelGGL(pack_kernel, dim3(blocks_for(n, 256)), dim3(256), 0, as_stream(stream), w_hwio, packed, Ci, Co,
                     kind);
  const int K = kind == 0 ? Ci : Co, N = kind == 0 ? Co : Ci;
  if (winograd_eligible(K, N))
    if (int e = winograd_pack(w_hwio, packed + n, Ci, Co, kind, as_stream(stream))) return e;
  return check_launch("nfs_conv3x3_pack");
}

int64_t nfs_conv3x3_workspace_floats(int B, int H, int W, int Ci, int Co) {
  if (B <= 0 || H <= 0 || W <= 0 || Ci <= 0 || Co <= 0) return 0;
  if (winograd_eligible(Ci, Co) && winograd_eligible(Co, Ci) && H >= 2 && W >= 2)
    return winograd_workspace_floats(B, H, W, Ci > Co ? Ci : Co, Ci > Co ? Ci : Co);
  // direct path: up to 16 K-splits of the larger M x N output, capped at 128 MB (big layers never split)
  const int64_t n = Ci > Co ? Ci : Co;
  const int64_t want = (int64_t)16 * B * H * W * n;
  return want < ((int64_t)32 << 20) ? want : ((int64_t)32 << 20);
}

// words of a layer's ReLU bit cache: [in: T*Ci/2][out (pooled layers): T*Co/2], T = B*ceil(H/4)*ceil(W/4); 0 when the
// layer does not run the F(4x4) Winograd transforms that keep it (the caller then passes NULL and the float masks)
int64_t nfs_conv3x3_relu_bits_words(int B, int H, int W, int Ci, int Co, int pooled) {
  if (B <= 0 || H < 2 || W < 2 || Ci <= 0 || Co <= 0 || Ci % 64 || Co % 64) return 0;
  static const bool off = getenv("NFS_NO_RELU_BITS") != nullptr;    // timing comparisons only
  ConvArgs a{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, B, H, W, Ci, Co, 0, 0, 0, 1, 1, 0};
  float dummy = 0.f;
  if (off || !takes_fused_pool(a, &dummy, INT64_MAX)) return 0;
  if (!pooled && winograd_path(B, H, W, Ci, Co) == 2) return winograd5_bits_words(B, H, W, Ci);   // F(5x5): its own layout
  const int64_t T = (int64_t)B * ((H + 3) / 4) * ((W + 3) / 4);
  return T * (Ci / 2) + (pooled ? T * (Co / 2) : 0);
}

// MFMA flops a conv call EXECUTES for K input and N output channels of the kernel it runs (forward: K = Ci, N = Co; data
// gradient: K = Co, N = Ci) -- the path is a function of the shapes alone (winograd_path), so this is too:
//   F(4x4,3x3), three kernels or one: 2 * 36 * T4 * K * N, T4 = B ceil(H/4) ceil(W/4)   (direct / 2.25, + tile padding)
//   F(5x5,3x3):                       2 * 49 * T5 * K * N, T5 = B ceil(H/5) ceil(W/5)   (direct / 4.59)
//   direct implicit GEMM (conv1_1, NFS_NO_WINOGRAD): 2 * B H W * 9 * K * N
// `pooled`: the fused-pool forms, which never take F(5x5).  What bench.py divides by the measured time, so that a
// fraction of the MFMA peak is a fraction of the peak.
double nfs_conv3x3_executed_flops(int B, int H, int W, int K, int N, int pooled) {
  if (B <= 0 || H <= 0 || W <= 0 || K <= 0 || N <= 0) return 0.0;
  static const bool no_wg = getenv("NFS_NO_WINOGRAD") != nullptr;
  const double direct = 2.0 * B * H * W * 9.0 * K * N;
  if (no_wg || !winograd_eligible(K, N) || H < 2 || W < 2) return direct;
  const int m = winograd_tile();
  if (m == 4 && !pooled && winograd_path(B, H, W, K, N) == 2)
    return 2.0 * 49.0 * B * ((H + 4) / 5) * ((W + 4) / 5) * (double)K * N;
  return 2.0 * (m + 2) * (m + 2) * B * ((H + m - 1) / m) * ((W + m - 1) / m) * (double)K * N;
}

static inline uint32_t* out_bits_of(uint32_t* relu_bits, int B, int H, int W, int Ci) {
  return relu_bits ? relu_bits + (int64_t)B * ((H + 3) / 4) * ((W + 3) / 4) * (Ci / 2) : nullptr;
}

int nfs_conv3x3_fwd(const float* x, const float* packed_fwd, const float* bias, float* y, int B, int H, int W, int Ci,
                    int Co, int relu, float* workspace, int64_t workspace_floats, uint32_t* relu_bits,
                    nfs_stream_t stream) {
  NFS_REQUIRE(x && packed_fwd && y, "nfs_conv3x3_fwd: null pointer");
  NFS_REQUIRE(B > 0 && H > 0 && W > 0, "nfs_conv3x3_fwd: non-positive dimension");
  NFS_REQUIRE((int64_t)B * H * W < ((int64_t)1 << 31) / 4, "nfs_conv3x3_fwd: too many pixels");
  NFS_REQUIRE(Co > 0 && Co % 64 == 0, "nfs_conv3x3_fwd: Co must be a multiple of 64");
  static const bool old_c3 = getenv("NFS_C3_OLD") != nullptr;        // timing comparisons only
  if (Ci == 3 && Co == 64 && !old_c3) {
    const int tiles_x = (W + C3F_TW - 1) / C3F_TW, tiles_y = (H + C3F_TH - 1) / C3F_TH;
    static const int per_cu = [] { const char* e = getenv("NFS_C3F_BLOCKS"); return e ? atoi(e) : 4; }();   // (27.1 us with 0, 25.2 with 4)
    const int64_t ntiles = (int64_t)B * tiles_x * tiles_y, cap = 256 * (int64_t)per_cu;
    hipLaunchKernelGGL(conv3x3_c3co64_fwd_kernel, dim3((unsigned)(per_cu > 0 && ntiles > cap ? cap : ntiles)), dim3(256), 0,
                       as_stream(stream), x, packed_fwd, bias, y, B, H, W, relu, tiles_x, tiles_y);
    return check_launch("nfs_conv3x3_fwd(c3, Co=64, MFMA)");
  }
  if (Ci == 3) {
    const int64_t n = (int64_t)B * H * W * (Co / 4);
    hipLaunchKernelGGL(conv3x3_c3_fwd_kernel, dim3(blocks_for(n, 256)), dim3(256), 27 * Co * sizeof(float),
                       as_stream(stream), x, packed_fwd, bias, y, B, H, W, Co, relu);
    return check_launch("nfs_conv3x3_fwd(c3)");
  }
  NFS_REQUIRE(Ci > 0 && Ci % 32 == 0, "nfs_conv3x3_fwd: Ci must be 3 or a multiple of 32");
  ConvArgs a{x, packed_fwd, bias, nullptr, y, nullptr, B, H, W, Ci, Co, 0, 0, 0, relu, 1, 0};
  return launch_conv<0>(a, workspace, workspace_floats, as_stream(stream), nullptr, nullptr, relu_bits, nullptr);
}

int nfs_conv3x3_dgrad(const float* gy, const float* packed_dgrad, const float* x_in, const float* addend, float* gx,
                      int B, int H, int W, int Ci, int Co, float* workspace, int64_t workspace_floats,
                      const uint32_t* relu_bits, int addend_unmasked, nfs_stream_t stream) {
  NFS_REQUIRE(gy && packed_dgrad && gx, "nfs_conv3x3_dgrad: null pointer");
  NFS_REQUIRE(B > 0 && H > 0 && W > 0, "nfs_conv3x3_dgrad: non-positive dimension");
  NFS_REQUIRE((int64_t)B * H * W < ((int64_t)1 << 31) / 4, "nfs_conv3x3_dgrad: too many pixels");
  NFS_REQUIRE(Co > 0 && Co % 32 == 0, "nfs_conv3x3_dgrad: Co must be a multiple of 32");
  if (Ci == 3) {
    NFS_REQUIRE(!x_in && !addend, "nfs_conv3x3_dgrad: Ci=3 takes no mask/addend");
    static const bool old_c3 = getenv("NFS_C3_OLD") != nullptr;      // timing comparisons only
    if (Co == 64 && !old_c3) {
      const int tiles_x = (W + C3D_TW - 1) / C3D_TW, tiles_y = (H + C3D_TH - 1) / C3D_TH;
      // blocks per CU of the tile-walking form (0: one block per tile, as before round 4).  8 x 200 x 200, 8 x 16 tiles:
      // 44.3 us with 0, 38.6 / 38.2 / 38.8 / 40.5 / 41.1 with 2 / 3 / 4 / 6 / 8; 14 x 14 tiles: 38.5 with 0, 34.1 / 36.5 /
      // 36.9 with 2 / 3 / 4 (tools/conv11_bench.py)
      static const int per_cu = [] { const char* e = getenv("NFS_C3D_BLOCKS"); return e ? atoi(e) : 2; }();
      const int64_t ntiles = (int64_t)B * tiles_x * tiles_y, cap = 256 * (int64_t)per_cu;
      hipLaunchKernelGGL(conv3x3_c3co64_dgrad_kernel, dim3((unsigned)(per_cu > 0 && ntiles > cap ? cap : ntiles)),
                         dim3(256), 0, as_stream(stream), gy, packed_dgrad, gx, B, H, W, tiles_x, tiles_y);
      return check_launch("nfs_conv3x3_dgrad(c3, Co=64, MFMA)");
    }
    const int64_t n = (int64_t)B * H * W;
    hipLaunchKernelGGL(conv3x3_c3_dgrad_kernel, dim3(blocks_for(n, 256)), dim3(256), 27 * Co * sizeof(float),
                       as_stream(stream), gy, packed_dgrad, gx, B, H, W, Co);
    return check_launch("nfs_conv3x3_dgrad(c3)");
  }
  NFS_REQUIRE(Ci > 0 && Ci % 64 == 0, "nfs_conv3x3_dgrad: Ci must be 3 or a multiple of 64");
  ConvArgs a{gy, packed_dgrad, x_in, addend, gx, nullptr, B, H, W, Co, Ci, 0, 0, 0, addend_unmasked ? 1 : 0, 1, 0};
  NFS_REQUIRE(!addend_unmasked || (x_in && takes_fused_pool(a, workspace, workspace_floats)),
              "nfs_conv3x3_dgrad: addend_unmasked needs x_in and the F(4x4) Winograd path");
  return launch_conv<1>(a, workspace, workspace_floats, as_stream(stream), nullptr, nullptr,
                        const_cast<uint32_t*>(relu_bits), nullptr);
}

// conv + bias + ReLU that also emits the 2x2 VALID average pool of its output (vgg.py: conv*_2/_4 -> pool*).
int nfs_conv3x3_fwd_pool(const float* x, const float* packed_fwd, const float* bias, float* y, float* y_pool, int B, int H,
                         int W, int Ci, int Co, int relu, float* workspace, int64_t workspace_floats,
                         uint32_t* relu_bits, nfs_stream_t stream) {
  NFS_REQUIRE(x && packed_fwd && y_pool, "nfs_conv3x3_fwd_pool: null pointer");
  NFS_REQUIRE(B > 0 && H > 1 && W > 1, "nfs_conv3x3_fwd_pool: need H, W >= 2");
  NFS_REQUIRE((int64_t)B * H * W < ((int64_t)1 << 31) / 4, "nfs_conv3x3_fwd_pool: too many pixels");
  NFS_REQUIRE(Co > 0 && Co % 64 == 0 && Ci > 0 && Ci % 32 == 0, "nfs_conv3x3_fwd_pool: Ci %% 32, Co %% 64 required");
  ConvArgs a{x, packed_fwd, bias, nullptr, y, nullptr, B, H, W, Ci, Co, 0, 0, 0, relu, 1, 0};
  if (takes_fused_pool(a, workspace, workspace_floats)) {
    NFS_REQUIRE(y || relu_bits, "nfs_conv3x3_fwd_pool: without y the ReLU bit cache must be recorded");
    return launch_conv<0>(a, workspace, workspace_floats, as_stream(stream), y_pool, nullptr, relu_bits,
                          out_bits_of(relu_bits, B, H, W, Ci));
  }
  NFS_REQUIRE(!relu_bits, "nfs_conv3x3_fwd_pool: the ReLU bit cache needs the fused Winograd path "
                          "(nfs_conv3x3_relu_bits_words returned 0 for this layer)");
  NFS_REQUIRE(y, "nfs_conv3x3_fwd_pool: y may only be NULL on the fused Winograd path with a ReLU bit cache");
  if (int e = launch_conv<0>(a, workspace, workspace_floats, as_stream(stream))) return e;
  return nfs_avgpool2_fwd(y, y_pool, B, H, W, Co, stream);
}

// data gradient of a conv whose output gradient arrives through the 2x2 average pool that follows its ReLU:
// g_y = 0.25 * gy_pool[h/2, w/2] * (x_out > 0) (0 outside the pooled area) is formed inside the input transform.
int nfs_conv3x3_dgrad_pool(const float* gy_pool, const float* x_out, const float* packed_dgrad, const float* x_in,
                           const float* addend, float* gx, int B, int H, int W, int Ci, int Co, float* workspace,
                           int64_t workspace_floats, const uint32_t* relu_bits, int addend_unmasked,
                           nfs_stream_t stream) {
  NFS_REQUIRE(gy_pool && packed_dgrad && gx, "nfs_conv3x3_dgrad_pool: null pointer");
  NFS_REQUIRE(x_out || relu_bits, "nfs_conv3x3_dgrad_pool: x_out or the layer's ReLU bit cache is required");
  NFS_REQUIRE(B > 0 && H > 1 && W > 1, "nfs_conv3x3_dgrad_pool: need H, W >= 2");
  NFS_REQUIRE((int64_t)B * H * W < ((int64_t)1 << 31) / 4, "nfs_conv3x3_dgrad_pool: too many pixels");
  NFS_REQUIRE(Co > 0 && Co % 32 == 0 && Ci > 0 && Ci % 64 == 0, "nfs_conv3x3_dgrad_pool: Co %% 32, Ci %% 64 required");
  ConvArgs a{gy_pool, packed_dgrad, x_in, addend, gx, nullptr, B, H, W, Co, Ci, 0, 0, 0, addend_unmasked ? 1 : 0, 1, 0};
  NFS_REQUIRE(!addend_unmasked || (x_in && takes_fused_pool(a, workspace, workspace_floats)),
              "nfs_conv3x3_dgrad_pool: addend_unmasked needs x_in and the F(4x4) Winograd path");
  if (takes_fused_pool(a, workspace, workspace_floats)) {
    uint32_t* rb = const_cast<uint32_t*>(relu_bits);
    return launch_conv<1>(a, workspace, workspace_floats, as_stream(stream), nullptr, x_out, rb,
                          out_bits_of(rb, B, H, W, Ci), true);
  }
  NFS_REQUIRE(x_out, "nfs_conv3x3_dgrad_pool: x_out may only be NULL on the fused Winograd path");
  // unfused: full-resolution gradient through the head of the workspace, conv on the rest
  const int64_t n = (int64_t)B * H * W * Co;
  NFS_REQUIRE(workspace && workspace_floats >= n, "nfs_conv3x3_dgrad_pool: workspace of >= B*H*W*Co floats required");
  if (int e = nfs_avgpool2_bwd(gy_pool, x_out, nullptr, workspace, B, H, W, Co, stream)) return e;
  a.x = workspace;
  return launch_conv<1>(a, workspace + n, workspace_floats - n, as_stream(stream));
}

}  // extern "C"
