// A7: Gram matrix G = F^T F, the style loss and its gradient dF = 2 F D
// (styler_base.py:96-102, 152-185) on the f32 MFMA (v_mfma_f32_32x32x2_f32).
//
// gram_fwd: M = N = C channels, K = pixels (up to 40000): a "TN" GEMM whose two operands are the same
//   pixel-major matrix.  G is symmetric, so only the TS x TS tile pairs t1 <= t2 are computed (TS = 64;
//   128 was measured slower) and the pixels are split into slabs of whole 32-pixel chunks so that the grid fills the
//   chip.  A block (4 waves, 2 x 2) stages each chunk [32 px][TS ch] of both strips in LDS with coalesced
//   float4 loads (prefetched one chunk ahead in registers, double-buffered in LDS; a diagonal pair stages one
//   strip only) and reads MFMA fragments as conflict-free ds_read_b32 rows: the pixel-major layout IS the
//   k-major operand layout of v_mfma_f32_32x32x2_f32, no transposition anywhere.  The partial tile goes to the
//   workspace; gram_reduce_kernel sums the slabs in a fixed order, applies the scale and writes the tile and
//   its mirror image (deterministic).  Without a workspace the scaled tile (and its mirror) is added with float atomics.
// gram_bwd: dF_b = 2 s_b F_b D_b is a plain batched GEMM (M = pixels, N = K = C): it runs on the batched f32-MFMA GEMMs
//   of winograd.hip -- the register-B 16-row form, D read in place (symmetric: row n serves as column n, a lane's four
//   k are consecutive floats), ReLU mask and scale in the epilogue.
#include "common.h"

namespace nfs {

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct GramArgs {
  const float* F;
  float* G;
  float* ws;    // partial tiles [b][pair][slab][TS*TS] (two-pass mode) or null (float atomics into G)
  const float* scale_dev;
  float scale;
  int B, HW, C;
  int ts;       // tile size 64 / 128
  int cps;      // 32-pixel chunks per slab
  int nslab;    // slabs per image
  int ntile;    // C / ts
  // grouped form (nfs_gram_style_group_fwd): the style loss is folded into whichever kernel completes a tile
  const float* Gs = nullptr;   // style Gram [Bs][C][C]
  float* Dmat = nullptr;       // 2 w (G - Gs) [B][C][C]
  float* part = nullptr;       // loss partials [slot][B] (one slot per finishing block of an image; plain stores)
  float weight = 0.f;
  int Bs = 1;
  int slot0 = 0;               // first slot of this layer
};

// all style layers of a step in one launch (the tile pairs of every layer, big units first) + one slab-reduce launch
constexpr int GR_MAXL = 8;
struct GramGroupArgs {
  GramArgs L[GR_MAXL];
  int ustart[GR_MAXL + 1];     // first block of layer l (multiples of 8: the XCD deal restarts per layer)
  int n;
};

constexpr int GR_KC = 32;

// One block's work: ``block`` of ``nblocks`` (a multiple of 8) dealt to this layer.
template <int TS>
__device__ __forceinline__ void gram_tn_block(const GramArgs& a, float* smem, int block, int nblocks) {
  constexpr int MT = TS / 64;                 // 32x32 MFMA tiles per wave and dimension (waves 2 x 2)
  constexpr int J = TS / 32;                  // float4 per thread, chunk and strip
  constexpr int R4 = TS / 4;                  // float4 per staged row
  float* As = smem;                           // [2][32][TS]
  float* Bs = smem + 2 * GR_KC * TS;          // [2][32][TS]
  const int t = threadIdx.x, lane = t & 63, wid = t >> 6;
  const int wm = wid >> 1, wn = wid & 1, i = lane & 31, h = lane >> 5;
  // G is symmetric: only the tile pairs t1 <= t2 are computed
  const int npair = a.ntile * (a.ntile + 1) / 2;
  // unit order (image, slab, tile pair) with the pair fastest, and a contiguous range of units per XCD (workgroups
  // are dealt round-robin to the 8 XCDs): the npair blocks that read the same 512-pixel slab of F then run together
  // behind one L2, and each 64-channel strip comes from HBM once instead of once per pair it takes part in
  const int64_t per_xcd = nblocks / 8;
  int64_t unit = (int64_t)(block % 8) * per_xcd + block / 8;
  const int64_t per_img = (int64_t)a.nslab * npair;
  if (unit >= per_img * a.B) return;
  const int b = (int)(unit / per_img);
  unit -= (int64_t)b * per_img;
  const int sl = (int)(unit / npair);
  const int pair = (int)(unit - (int64_t)sl * npair);
  int t1 = 0, t2 = pair;
  while (t2 >= a.ntile - t1) { t2 -= a.ntile - t1; ++t1; }
  t2 += t1;
  const bool diag = t1 == t2;                 // both operands are the same strip: stage it once
  const int total_chunks = (a.HW + GR_KC - 1) / GR_KC;
  const int c_begin = sl * a.cps;
  const int nchunks = min(a.cps, total_chunks - c_begin);
  const float* Fb = a.F + (int64_t)b * a.HW * a.C;

  // staging: thread t moves float4 #(t % R4) of rows t / R4 + (256 / R4) * j of the chunk
  const int col4 = t % R4, row0 = t / R4;
  constexpr int RSTEP = 256 / R4;
  const float* ga = Fb + t1 * TS + 4 * col4;
  const float* gb = Fb + t2 * TS + 4 * col4;
  float4 a0, a1, a2, a3, b0, b1, b2, b3;      // named registers (an indexed array would go to scratch)
#define NFS_GR_LD(dst_, base_, j_, c_)                                                          \
  {                                                                                             \
    const int p_ = (c_begin + (c_)) * GR_KC + row0 + RSTEP * (j_);                              \
    dst_ = make_float4(0.f, 0.f, 0.f, 0.f);                                                     \
    if (p_ < a.HW) dst_ = *reinterpret_cast<const float4*>(base_ + (int64_t)p_ * a.C);          \
  }
#define NFS_GR_LOAD(c_)                                                                         \
  {                                                                                             \
    NFS_GR_LD(a0, ga, 0, c_) NFS_GR_LD(a1, ga, 1, c_)                                           \
    if (J > 2) { NFS_GR_LD(a2, ga, 2, c_) NFS_GR_LD(a3, ga, 3, c_) }                            \
    if (!diag) {                                                                                \
      NFS_GR_LD(b0, gb, 0, c_) NFS_GR_LD(b1, gb, 1, c_)                                         \
      if (J > 2) { NFS_GR_LD(b2, gb, 2, c_) NFS_GR_LD(b3, gb, 3, c_) }                          \
    }                                                                                           \
  }
  b0 = b1 = b2 = b3 = make_float4(0.f, 0.f, 0.f, 0.f);
  a2 = a3 = b0;
  NFS_GR_LOAD(0)

  f32x16 acc[MT][MT];
#pragma unroll
  for (int x = 0; x < MT; ++x)
#pragma unroll
    for (int y = 0; y < MT; ++y)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[x][y][r] = 0.f;

  const int afrag = h * TS + wm * (TS / 2) + i;      // + (2 step) * TS + 32 mt
  const int bfrag = h * TS + wn * (TS / 2) + i;
  for (int c = 0; c < nchunks; ++c) {
    float* Ac = As + (c & 1) * GR_KC * TS;
    float* Bc = diag ? Ac : Bs + (c & 1) * GR_KC * TS;
    {
      float* ad = Ac + row0 * TS + 4 * col4;
      *reinterpret_cast<float4*>(ad) = a0;
      *reinterpret_cast<float4*>(ad + RSTEP * TS) = a1;
      if (J > 2) {
        *reinterpret_cast<float4*>(ad + 2 * RSTEP * TS) = a2;
        *reinterpret_cast<float4*>(ad + 3 * RSTEP * TS) = a3;
      }
      if (!diag) {
        float* bd = Bc + row0 * TS + 4 * col4;
        *reinterpret_cast<float4*>(bd) = b0;
        *reinterpret_cast<float4*>(bd + RSTEP * TS) = b1;
        if (J > 2) {
          *reinterpret_cast<float4*>(bd + 2 * RSTEP * TS) = b2;
          *reinterpret_cast<float4*>(bd + 3 * RSTEP * TS) = b3;
        }
      }
    }
    __syncthreads();                        // buffer (c&1) visible; buffer (c+1)&1 was last read in iteration c-1
    if (c + 1 < nchunks) NFS_GR_LOAD(c + 1)
#pragma unroll
    for (int st = 0; st < GR_KC / 2; ++st) {
      float af[MT], bf[MT];
#pragma unroll
      for (int m = 0; m < MT; ++m) af[m] = Ac[afrag + 2 * st * TS + 32 * m];
#pragma unroll
      for (int n = 0; n < MT; ++n) bf[n] = Bc[bfrag + 2 * st * TS + 32 * n];
#pragma unroll
      for (int n = 0; n < MT; ++n)
#pragma unroll
        for (int m = 0; m < MT; ++m)
          acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[m], bf[n], acc[m][n], 0, 0, 0);
    }
  }
#undef NFS_GR_LOAD
#undef NFS_GR_LD

  // epilogue: transpose the tile through LDS, leave as float4 rows
  constexpr int OS = TS + 4;
  float* otile = smem;
  __syncthreads();
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = wm * (TS / 2) + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
#pragma unroll
      for (int n = 0; n < MT; ++n) otile[row * OS + wn * (TS / 2) + n * 32 + i] = acc[m][n][r];
    }
  __syncthreads();
  if (a.ws) {
    float* wt = a.ws + (((int64_t)b * npair + pair) * a.nslab + sl) * TS * TS;
#pragma unroll
    for (int e = 0; e < (TS * R4) / 256; ++e) {
      const int f = t + 256 * e;
      const int row = f / R4, q = f - row * R4;
      *reinterpret_cast<float4*>(wt + row * TS + 4 * q) = *reinterpret_cast<const float4*>(otile + row * OS + 4 * q);
    }
    return;
  }
  const float sc = a.scale * (a.scale_dev ? a.scale_dev[b] : 1.f);
  float* Gb = a.G + (int64_t)b * a.C * a.C;
  if (a.nslab == 1 && a.Dmat) {
    // grouped form, one slab per tile pair: this block holds the complete sums of its tile, so it also forms the style
    // loss of the tile (styler_base.py:181: sum (G - Gs)^2; an off-diagonal tile stands for its mirror image too) and
    // D = 2 w (G - Gs) (tile and mirror; Gs is exactly symmetric -- its mirrored tiles are copies); G itself is written
    // only when asked for
    const float* Gsb = a.Gs + (int64_t)(b % a.Bs) * a.C * a.C;
    float* Db = a.Dmat + (int64_t)b * a.C * a.C;
    float* Go = a.G ? a.G + (int64_t)b * a.C * a.C : nullptr;
    const float w2 = 2.f * a.weight;
    float lp = 0.f;
    for (int f = t; f < TS * R4; f += 256) {
      const int row = f / R4, q = f - row * R4;
      const int64_t o = (int64_t)(t1 * TS + row) * a.C + t2 * TS + 4 * q;
      float4 v = *reinterpret_cast<const float4*>(otile + row * OS + 4 * q);
      v.x *= sc; v.y *= sc; v.z *= sc; v.w *= sc;
      if (Go) *reinterpret_cast<float4*>(Go + o) = v;
      const float4 g = *reinterpret_cast<const float4*>(Gsb + o);
      const float4 d = make_float4(v.x - g.x, v.y - g.y, v.z - g.z, v.w - g.w);
      lp += (d.x * d.x + d.y * d.y) + (d.z * d.z + d.w * d.w);
      *reinterpret_cast<float4*>(Db + o) = make_float4(w2 * d.x, w2 * d.y, w2 * d.z, w2 * d.w);
    }
    if (!diag)
      for (int f = t; f < TS * R4; f += 256) {
        const int col = f / R4, q = f - col * R4;            // mirrored row t2*TS + col, columns t1*TS + 4q ..
        const float4 v = make_float4(otile[(4 * q) * OS + col] * sc, otile[(4 * q + 1) * OS + col] * sc,
                                     otile[(4 * q + 2) * OS + col] * sc, otile[(4 * q + 3) * OS + col] * sc);
        const int64_t o = (int64_t)(t2 * TS + col) * a.C + t1 * TS + 4 * q;
        if (Go) *reinterpret_cast<float4*>(Go + o) = v;
        const float4 g = *reinterpret_cast<const float4*>(Gsb + o);
        *reinterpret_cast<float4*>(Db + o) = make_float4(w2 * (v.x - g.x), w2 * (v.y - g.y), w2 * (v.z - g.z),
                                                         w2 * (v.w - g.w));
      }
    lp = block_sum(lp, smem + TS * OS);        // (scratch behind the tile: 64 x 68 floats of the 32 KB)
    if (t == 0) a.part[(int64_t)(a.slot0 + pair) * a.B + b] = a.weight * (diag ? lp : 2.f * lp);
    return;
  }
  if (a.nslab == 1) {
    // one slab per tile pair (many pairs, few pixels: the deep layers): this block holds the complete sums -- the scaled
    // tile and its mirror image are written directly, no partials and no second pass
    for (int f = t; f < TS * R4; f += 256) {
      const int row = f / R4, q = f - row * R4;
      float4 v = *reinterpret_cast<const float4*>(otile + row * OS + 4 * q);
      v.x *= sc; v.y *= sc; v.z *= sc; v.w *= sc;
      *reinterpret_cast<float4*>(Gb + (int64_t)(t1 * TS + row) * a.C + t2 * TS + 4 * q) = v;
    }
    if (!diag)
      for (int f = t; f < TS * R4; f += 256) {
        const int col = f / R4, q = f - col * R4;            // mirrored row t2*TS + col, columns t1*TS + 4q ..
        const float4 v = make_float4(otile[(4 * q) * OS + col] * sc, otile[(4 * q + 1) * OS + col] * sc,
                                     otile[(4 * q + 2) * OS + col] * sc, otile[(4 * q + 3) * OS + col] * sc);
        *reinterpret_cast<float4*>(Gb + (int64_t)(t2 * TS + col) * a.C + t1 * TS + 4 * q) = v;
      }
    return;
  }
  for (int f = t; f < TS * TS; f += 256) {
    const int row = f / TS, col = f - row * TS;
    const float v = otile[row * OS + col] * sc;
    const int gr = t1 * TS + row, gc = t2 * TS + col;
    atomicAdd(Gb + (int64_t)gr * a.C + gc, v);
    if (!diag) atomicAdd(Gb + (int64_t)gc * a.C + gr, v);   // mirrored tile
  }
}

template <int TS>
__global__ void __launch_bounds__(256, 2) gram_tn_kernel(GramArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  gram_tn_block<TS>(a, smem, (int)blockIdx.x, (int)gridDim.x);
}

// every style layer's tile pairs in one launch: block -> (layer, its block within the layer)
__global__ void __launch_bounds__(256, 2) gram_tn_group_kernel(GramGroupArgs g) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  int l = 0;
  while (l + 1 < g.n && (int)blockIdx.x >= g.ustart[l + 1]) ++l;
  gram_tn_block<64>(g.L[l], smem, (int)blockIdx.x - g.ustart[l], g.ustart[l + 1] - g.ustart[l]);
}

// Second pass: a block owns GRD_ROWS rows of one (image, tile pair).  Its 4 waves sum interleaved slabs (wave g
// takes slabs g, g+4, ...), the partial sums are combined in a fixed order (deterministic), scaled, written
// into G as float4 rows and -- for an off-diagonal pair -- mirrored into the transposed tile.
constexpr int GRD_ROWS = 4;
__device__ __forceinline__ void gram_reduce_block(const GramArgs& a, int unit) {
  constexpr int TS = 64;
  __shared__ float4 part[4][64];
  __shared__ float tile[GRD_ROWS][TS + 1];
  const int t = threadIdx.x, lane = t & 63, g = t >> 6;
  const int row = lane >> 4, q = lane & 15;
  constexpr int groups = TS / GRD_ROWS;
  const int npair = a.ntile * (a.ntile + 1) / 2;
  const int rg = unit % groups;
  unit /= groups;
  const int pair = unit % npair, b = unit / npair;
  int t1 = 0, t2 = pair;
  while (t2 >= a.ntile - t1) { t2 -= a.ntile - t1; ++t1; }
  t2 += t1;
  const int64_t tsz = (int64_t)TS * TS;
  const float* p = a.ws + (((int64_t)b * npair + pair) * a.nslab) * tsz + (int64_t)(rg * GRD_ROWS + row) * TS + 4 * q;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int k = g; k < a.nslab; k += 4) {
    const float4 x = *reinterpret_cast<const float4*>(p + (int64_t)k * tsz);
    s.x += x.x; s.y += x.y; s.z += x.z; s.w += x.w;
  }
  part[g][lane] = s;
  __syncthreads();
  if (g > 0) return;
  const float sc = a.scale * (a.scale_dev ? a.scale_dev[b] : 1.f);
  const float4 p1 = part[1][lane], p2 = part[2][lane], p3 = part[3][lane];
  s.x = ((s.x + p1.x) + (p2.x + p3.x)) * sc; s.y = ((s.y + p1.y) + (p2.y + p3.y)) * sc;
  s.z = ((s.z + p1.z) + (p2.z + p3.z)) * sc; s.w = ((s.w + p1.w) + (p2.w + p3.w)) * sc;
  float* Gb = a.G ? a.G + (int64_t)b * a.C * a.C : nullptr;
  const int64_t o = (int64_t)(t1 * TS + rg * GRD_ROWS + row) * a.C + t2 * TS + 4 * q;
  if (Gb) *reinterpret_cast<float4*>(Gb + o) = s;
  // grouped form: the style loss of these four rows (an off-diagonal tile stands for its mirror image too) and
  // D = 2 w (G - Gs); one loss partial per block, a plain store into the block's own slot (deterministic, no atomics)
  const float* Gsb = a.Dmat ? a.Gs + (int64_t)(b % a.Bs) * a.C * a.C : nullptr;
  float* Db = a.Dmat ? a.Dmat + (int64_t)b * a.C * a.C : nullptr;
  const float w2 = 2.f * a.weight;
  if (Db) {
    const float4 gs = *reinterpret_cast<const float4*>(Gsb + o);
    const float4 d = make_float4(s.x - gs.x, s.y - gs.y, s.z - gs.z, s.w - gs.w);
    *reinterpret_cast<float4*>(Db + o) = make_float4(w2 * d.x, w2 * d.y, w2 * d.z, w2 * d.w);
    const float lp = wave_sum((d.x * d.x + d.y * d.y) + (d.z * d.z + d.w * d.w));
    if (lane == 0) a.part[(int64_t)(a.slot0 + pair * groups + rg) * a.B + b] = a.weight * (t1 == t2 ? lp : 2.f * lp);
  }
  if (t1 == t2) return;
  tile[row][4 * q] = s.x; tile[row][4 * q + 1] = s.y; tile[row][4 * q + 2] = s.z; tile[row][4 * q + 3] = s.w;
  __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): the wave's own LDS writes are done (single wave from here)
  __builtin_amdgcn_wave_barrier();
  // mirrored: G[t2*TS + c][t1*TS + rg*GRD_ROWS + 0..3] = tile[0..3][c], one float4 per lane (c = lane)
  const float4 v = make_float4(tile[0][lane], tile[1][lane], tile[2][lane], tile[3][lane]);
  const int64_t om = (int64_t)(t2 * TS + lane) * a.C + t1 * TS + rg * GRD_ROWS;
  if (Gb) *reinterpret_cast<float4*>(Gb + om) = v;
  if (Db) {
    const float4 gs = *reinterpret_cast<const float4*>(Gsb + om);
    *reinterpret_cast<float4*>(Db + om) = make_float4(w2 * (v.x - gs.x), w2 * (v.y - gs.y), w2 * (v.z - gs.z),
                                                      w2 * (v.w - gs.w));
  }
}

__global__ void __launch_bounds__(256) gram_reduce_kernel(GramArgs a) { gram_reduce_block(a, (int)blockIdx.x); }

// the slab reductions of every layer that has slabs, one launch (ustart: first block of layer l, any alignment)
__global__ void __launch_bounds__(256) gram_reduce_group_kernel(GramGroupArgs g) {
  int l = 0;
  while (l + 1 < g.n && (int)blockIdx.x >= g.ustart[l + 1]) ++l;
  gram_reduce_block(g.L[l], (int)blockIdx.x - g.ustart[l]);
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

// Content loss on a post-ReLU activation F [B,HW,C] (styler_base.py:135-150); the terms are means over the whole
// batch, image b contributes its own partial sum to loss[b].  g_acc += dL/d(pre-activation) = dL/dF * (F > 0).
//   mode 0 (content_channel c != 0): -mean(F[...,c]) + mean|F[...,:c]| + mean|F[...,c+1:]|
//   mode 1 (no channel):             -mean(F)
//   mode 2 (content image):          mean((F - amp * target)^2), target [Bt,HW,C] with bt = b % Bt
__global__ void __launch_bounds__(256) content_loss_kernel(const float* __restrict__ F, const float* __restrict__ target,
                                                           float* __restrict__ loss, float* __restrict__ g_acc,
                                                           int B, int Bt, int64_t HWC, int C, int channel, int mode,
                                                           float weight, float amp, int signed_f) {
  __shared__ float red[16];
  const int b = blockIdx.y;
  const float n_all = (float)B * (float)HWC;
  const float n_pix = n_all / (float)C;
  const float c_on = -weight / n_pix;                                          // the maximised channel
  const float c_lo = channel > 0 ? weight / (n_pix * (float)channel) : 0.f;     // channels below / above it
  const float c_hi = channel + 1 < C ? weight / (n_pix * (float)(C - channel - 1)) : 0.f;
  float part = 0.f;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < HWC; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t i = (int64_t)b * HWC + e;
    const float f = F[i];
    float l, g;
    if (mode == 0) {
      const int ch = (int)(e % C);
      const float k = ch == channel ? c_on : (ch < channel ? c_lo : c_hi);
      if (ch == channel || !signed_f) {
        l = k * f;                                 // post-ReLU (f >= 0): |f| = f
        g = k;
      } else {                                     // a tensor that is not a ReLU output: |f|, d|f| = sign(f) (0 at 0, as TF)
        l = k * fabsf(f);
        g = f > 0.f ? k : (f < 0.f ? -k : 0.f);
      }
    } else if (mode == 1) {
      l = -weight / n_all * f;
      g = -weight / n_all;
    } else {
      const float diff = f - amp * target[(int64_t)(b % Bt) * HWC + e];
      l = weight / n_all * diff * diff;
      g = 2.f * weight / n_all * diff;
    }
    part += l;
    if (signed_f || f > 0.f) g_acc[i] += g;        // post-ReLU: the gradient wrt the pre-activation
  }
  part = block_sum(part, red);
  if (threadIdx.x == 0) atomicAdd(loss + b, part);
}

// winograd.hip: batched f32-MFMA GEMM (LDS-staged, double-buffered) shared with the Winograd convolution
int gram_bwd_gemm(const float* F, const float* Dm, float* dF, int B, int HW, int C, float alpha, const float* alpha_dev,
                  int relu_mask, int cus, hipStream_t s);
int gram_bwd_gemm_group(const float* const* F, const float* const* Dm, float* const* dF, const int* HW, const int* C,
                        const float* alpha, const int* relu_mask, int n, int B, hipStream_t s);

}  // namespace nfs

using namespace nfs;

extern "C" {

static int gram_cus() {
  static int cus = 0;
  if (cus == 0) { const int c = nfs_device_cus(); cus = c > 0 ? c : 256; }
  return cus;
}

static void gram_plan(GramArgs& a, int cus) {
  static const int env_cps = [] { const char* e = getenv("NFS_GRAM_CPS"); return e ? atoi(e) : 0; }();
  // measured (tools/gram_bench.py, tile-size / NFS_GRAM_CPS sweeps): 64 x 64 tiles beat 128 x 128 at every VGG
  // shape (4x more blocks, 4x smaller partial tiles), and ~16 chunks (512 pixels) per slab is the sweet spot
  // between per-block fill/epilogue cost and the number of partial tiles the reduce pass has to read.
  a.ts = 64;
  a.ntile = a.C / a.ts;
  const int64_t pairs = (int64_t)a.B * a.ntile * (a.ntile + 1) / 2;
  const int total_chunks = (a.HW + GR_KC - 1) / GR_KC;
  int cps = 16;
  // few images / few tile pairs: shorter slabs until the grid covers the chip twice (not below 4 chunks)
  while (cps > 4 && pairs * ((total_chunks + cps - 1) / cps) < 2 * (int64_t)cus) cps >>= 1;
  if (cps > total_chunks) cps = total_chunks;
  if ((total_chunks + cps - 1) / cps > 256) cps = (total_chunks + 255) / 256;   // bound the reduce fan-in
  // many tile pairs, few pixels (relu4_1, relu5_1 at 8 views): one slab -- the tile kernel then writes G itself
  if (pairs >= cus && total_chunks <= 32) cps = total_chunks;
  if (env_cps > 0) cps = env_cps < total_chunks ? env_cps : total_chunks;
  a.cps = cps;
  a.nslab = (total_chunks + cps - 1) / cps;
}

int64_t nfs_gram_workspace_floats(int B, int HW, int C) {
  if (B <= 0 || HW <= 0 || C <= 0 || C % 64) return 0;
  GramArgs a;
  a.B = B; a.HW = HW; a.C = C;
  gram_plan(a, gram_cus());
  return (int64_t)B * (a.ntile * (a.ntile + 1) / 2) * a.nslab * a.ts * a.ts + 4 * 4096;
}

int nfs_gram_fwd(const float* F, float* G, int B, int HW, int C, const float* scale_dev, float scale,
                 float* workspace, int64_t workspace_floats, nfs_stream_t stream) {
  NFS_REQUIRE(F && G, "nfs_gram_fwd: null pointer");
  NFS_REQUIRE(B > 0 && HW > 0, "nfs_gram_fwd: non-positive dimension");
  NFS_REQUIRE(C > 0 && C % 64 == 0, "nfs_gram_fwd: C must be a multiple of 64");
  GramArgs a;
  a.F = F; a.G = G; a.scale_dev = scale_dev; a.scale = scale; a.B = B; a.HW = HW; a.C = C;
  gram_plan(a, gram_cus());
  const int64_t units = (int64_t)B * (a.ntile * (a.ntile + 1) / 2) * a.nslab;
  a.ws = (workspace && workspace_floats >= nfs_gram_workspace_floats(B, HW, C) && a.nslab > 1) ? workspace : nullptr;
  const size_t lds = 4 * GR_KC * 64 * sizeof(float);            // 32 KB (>= the 64 x 68 epilogue tile)
  hipLaunchKernelGGL(gram_tn_kernel<64>, dim3((unsigned)((units + 7) / 8 * 8)), dim3(256), lds, as_stream(stream), a);
  if (a.ws) {
    const unsigned rb = (unsigned)((int64_t)B * (a.ntile * (a.ntile + 1) / 2) * (64 / GRD_ROWS));
    hipLaunchKernelGGL(gram_reduce_kernel, dim3(rb), dim3(256), 0, as_stream(stream), a);
  }
  return check_launch("nfs_gram_fwd");
}

// ---- grouped form: every style layer of a step in two launches (tile pairs + slab reduce), style loss folded in ----
struct GramGroupPlan {
  GramGroupArgs tn, rd;
  int tn_blocks, rd_blocks, parts;
  int64_t ws_floats;
};

static int gram_group_plan(const nfs_gram_layer_t* layers, int n, GramGroupPlan& P) {
  if (!layers || n < 1 || n > GR_MAXL) return -1;
  int order[GR_MAXL];
  GramArgs L[GR_MAXL];
  for (int l = 0; l < n; ++l) {
    const nfs_gram_layer_t& y = layers[l];
    if (!y.F || !y.Gs || !y.Dmat || y.B <= 0 || y.Bs <= 0 || y.HW <= 0 || y.C <= 0 || y.C % 64 || y.B != layers[0].B)
      return -1;
    GramArgs& a = L[l];
    a = GramArgs();
    a.F = y.F; a.G = y.G; a.scale_dev = nullptr; a.scale = y.scale; a.B = y.B; a.HW = y.HW; a.C = y.C;
    a.Gs = y.Gs; a.Dmat = y.Dmat; a.weight = y.weight; a.Bs = y.Bs;
    a.ts = 64;
    a.ntile = y.C / 64;
    // units of ~16 chunks (512 pixels) whatever the layer: the launch is filled by all layers together; a short image
    // (<= 32 chunks) is one slab, its tile kernel completes the tile itself
    const int total_chunks = (y.HW + GR_KC - 1) / GR_KC;
    a.cps = total_chunks <= 32 ? total_chunks : 16;
    a.nslab = (total_chunks + a.cps - 1) / a.cps;
    order[l] = l;
  }
  // big units first (chunks per unit, off-diagonal pairs stage two strips: more pairs first among equals)
  for (int i = 1; i < n; ++i)
    for (int j = i; j > 0 && (L[order[j]].cps > L[order[j - 1]].cps ||
                              (L[order[j]].cps == L[order[j - 1]].cps && L[order[j]].ntile > L[order[j - 1]].ntile)); --j) {
      const int tmp = order[j]; order[j] = order[j - 1]; order[j - 1] = tmp;
    }
  P.tn.n = n;
  P.rd.n = 0;
  int ub = 0, rb = 0, slot = 0;
  int64_t ws = 0;
  for (int k = 0; k < n; ++k) {
    GramArgs a = L[order[k]];
    const int npair = a.ntile * (a.ntile + 1) / 2;
    a.slot0 = slot;
    a.ws = nullptr;
    P.tn.ustart[k] = ub;
    ub += (int)(((int64_t)a.B * npair * a.nslab + 7) / 8 * 8);
    if (a.nslab > 1) {
      a.ws = reinterpret_cast<float*>(ws * sizeof(float));      // offset for now; the base is added at launch
      ws += (int64_t)a.B * npair * a.nslab * 64 * 64;
      slot += npair * (64 / GRD_ROWS);
      P.rd.ustart[P.rd.n] = rb;
      rb += a.B * npair * (64 / GRD_ROWS);
      P.rd.L[P.rd.n++] = a;
    } else {
      slot += npair;
    }
    P.tn.L[k] = a;
  }
  P.tn.ustart[n] = ub;
  P.rd.ustart[P.rd.n] = rb;
  P.tn_blocks = ub; P.rd_blocks = rb; P.parts = slot; P.ws_floats = ws;
  return 0;
}

int64_t nfs_gram_style_group_workspace_floats(const nfs_gram_layer_t* layers, int n) {
  GramGroupPlan P;
  return gram_group_plan(layers, n, P) == 0 ? P.ws_floats : -1;
}

int nfs_gram_style_group_parts(const nfs_gram_layer_t* layers, int n) {
  GramGroupPlan P;
  return gram_group_plan(layers, n, P) == 0 ? P.parts : -1;
}

int nfs_gram_style_group_fwd(const nfs_gram_layer_t* layers, int n, float* loss_parts, float* workspace,
                             int64_t workspace_floats, nfs_stream_t stream) {
  GramGroupPlan P;
  NFS_REQUIRE(gram_group_plan(layers, n, P) == 0,
              "nfs_gram_style_group_fwd: 1..8 layers with F, Gs, Dmat set, one batch size, C a multiple of 64");
  NFS_REQUIRE(loss_parts, "nfs_gram_style_group_fwd: null loss_parts");
  NFS_REQUIRE(P.ws_floats == 0 || (workspace && workspace_floats >= P.ws_floats),
              "nfs_gram_style_group_fwd: workspace smaller than nfs_gram_style_group_workspace_floats");
  for (int k = 0; k < n; ++k) {
    P.tn.L[k].part = loss_parts;
    if (P.tn.L[k].nslab > 1) P.tn.L[k].ws = workspace + reinterpret_cast<int64_t>(P.tn.L[k].ws) / (int64_t)sizeof(float);
  }
  for (int k = 0; k < P.rd.n; ++k) {
    P.rd.L[k].part = loss_parts;
    P.rd.L[k].ws = workspace + reinterpret_cast<int64_t>(P.rd.L[k].ws) / (int64_t)sizeof(float);
  }
  const size_t lds = 4 * GR_KC * 64 * sizeof(float);
  hipLaunchKernelGGL(gram_tn_group_kernel, dim3((unsigned)P.tn_blocks), dim3(256), lds, as_stream(stream), P.tn);
  if (P.rd_blocks > 0)
    hipLaunchKernelGGL(gram_reduce_group_kernel, dim3((unsigned)P.rd_blocks), dim3(256), 0, as_stream(stream), P.rd);
  return check_launch("nfs_gram_style_group_fwd");
}

int nfs_gram_group_bwd(const nfs_gram_layer_t* layers, int n, nfs_stream_t stream) {
  NFS_REQUIRE(layers && n >= 1 && n <= GR_MAXL, "nfs_gram_group_bwd: 1..8 layers");
  const float* F[GR_MAXL]; const float* Dm[GR_MAXL]; float* dF[GR_MAXL];
  int HW[GR_MAXL], C[GR_MAXL], mask[GR_MAXL];
  float alpha[GR_MAXL];
  for (int l = 0; l < n; ++l) {
    const nfs_gram_layer_t& y = layers[l];
    NFS_REQUIRE(y.F && y.Dmat && y.dF, "nfs_gram_group_bwd: null pointer");
    NFS_REQUIRE(y.B > 0 && y.B == layers[0].B && y.HW > 0 && y.C > 0 && y.C % 64 == 0,
                "nfs_gram_group_bwd: one batch size, C a multiple of 64");
    F[l] = y.F; Dm[l] = y.Dmat; dF[l] = y.dF; HW[l] = y.HW; C[l] = y.C; mask[l] = y.relu_mask; alpha[l] = 2.f * y.scale;
  }
  NFS_REQUIRE(gram_bwd_gemm_group(F, Dm, dF, HW, C, alpha, mask, n, layers[0].B, as_stream(stream)) == 0,
              "nfs_gram_group_bwd: shape outside the 16-row GEMM's 32-bit operand offsets");
  return 0;
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

static int content_loss_launch(const char* who, const float* F, const float* target, float* loss_acc, float* g_acc, int B,
                               int Bt, int HW, int C, int channel, int mode, float weight, float amp, int signed_f,
                               nfs_stream_t stream) {
  NFS_REQUIRE(F && loss_acc && g_acc, "%s: null pointer", who);
  NFS_REQUIRE(B > 0 && HW > 0 && C > 0, "%s: non-positive dimension", who);
  NFS_REQUIRE(mode >= 0 && mode <= 2, "%s: mode must be 0 (channel), 1 (all) or 2 (target)", who);
  NFS_REQUIRE(mode != 0 || (channel > 0 && channel < C), "%s: channel out of range", who);
  NFS_REQUIRE(mode != 2 || (target && Bt > 0), "%s: mode 2 needs the target features", who);
  const int64_t HWC = (int64_t)HW * C;
  const unsigned nb = blocks_for(HWC, 256) < 64u ? blocks_for(HWC, 256) : 64u;
  hipLaunchKernelGGL(content_loss_kernel, dim3(nb, B), dim3(256), 0, as_stream(stream), F, target, loss_acc, g_acc, B,
                     Bt > 0 ? Bt : 1, HWC, C, channel, mode, weight, amp, signed_f);
  return check_launch(who);
}

int nfs_content_loss(const float* F, const float* target, float* loss_acc, float* g_acc, int B, int Bt, int HW, int C,
                     int channel, int mode, float weight, float amp, nfs_stream_t stream) {
  return content_loss_launch("nfs_content_loss", F, target, loss_acc, g_acc, B, Bt, HW, C, channel, mode, weight, amp, 0,
                             stream);
}

int nfs_content_loss_signed(const float* F, const float* target, float* loss_acc, float* g_acc, int B, int Bt, int HW,
                            int C, int channel, int mode, float weight, float amp, nfs_stream_t stream) {
  return content_loss_launch("nfs_content_loss_signed", F, target, loss_acc, g_acc, B, Bt, HW, C, channel, mode, weight,
                             amp, 1, stream);
}

int nfs_gram_bwd(const float* F, const float* Dmat, float* dF, int B, int HW, int C, const float* scale_dev,
                 float scale, int relu_mask, nfs_stream_t stream) {
  NFS_REQUIRE(F && Dmat && dF, "nfs_gram_bwd: null pointer");
  NFS_REQUIRE(B > 0 && HW > 0, "nfs_gram_bwd: non-positive dimension");
  NFS_REQUIRE(C > 0 && C % 64 == 0, "nfs_gram_bwd: C must be a multiple of 64");
  // dF[b] = 2 * scale_b * F[b] @ D[b]: M = pixels, N = K = C; D is symmetric, so its rows serve as columns
  return gram_bwd_gemm(F, Dmat, dF, B, HW, C, 2.f * scale, scale_dev, relu_mask, gram_cus(), as_stream(stream));
}

}  // extern "C"
