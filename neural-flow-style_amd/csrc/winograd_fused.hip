// A6, narrow layers (64 or 128 channels on both sides: conv1_2, conv2_1, conv2_2 and their data gradients): the whole
// F(4x4, 3x3) Winograd convolution in ONE kernel.
//
// The three-kernel form (winograd.hip) moves every activation through HBM 2.25 x 4 times (V written and read, M written
// and read) on top of the compulsory read of x and write of y; for these layers the 36 GEMMs have K <= 128 and are
// bound by exactly that traffic.  Here a wave owns a RUN of 16 consecutive 4x4 output tiles x 16 output channels and
// keeps all 36 transform components of that block in accumulators (36 MFMA tiles of 16x16 = 144 AGPRs):
//
//   * the 6x6 input patches of the run are staged through LDS in 16-channel slices (coalesced 64-byte pieces, double
//     buffered, shared by the block's four waves = 64 output channels);
//   * each lane applies B^T d B to ONE (tile, input channel) of the next slice and leaves the 36 components in a second
//     LDS buffer, from which lane (tile t = lane & 15, channel group g = lane >> 4) -- the A-operand owner of
//     v_mfma_f32_16x16x4_f32 -- reads A_z[t][k]; the transform of slice j+1 runs under the MFMAs of slice j;
//   * B_z fragments stream from a pre-swizzled copy of the filters (L2-resident: 36 K N floats), one 8-byte load per
//     (z, half) and lane;
//   * the C layout of the MFMA gives lane (column n = lane & 15, rows 4 g .. 4 g + 3) all 36 components of four tiles
//     of one output channel: A^T M A, bias / ReLU / mask / addend, the ReLU bit cache and the 2x2 average pool are all
//     in-lane (the bit words pair two channels: one __shfl_xor).
//
// HBM traffic: x once (plus halo re-reads that hit L2), y once.  Same V, same transforms as winograd.hip; the k order
// of the sums differs, so results agree to float32 rounding, not bit for bit.
#include "common.h"
#include "winograd_math.h"

namespace nfs {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int WF_PXF = 256;          // floats per staged patch pixel: 16 tiles x 16 channels
constexpr int WF_BUF = 36 * WF_PXF;  // one staging buffer: 36 KB

struct WfArgs {
  const float* x;             // [B,H,W,K]; POOLED: the pooled gradient [B,H/2,W/2,K]
  const float4* Uf;           // filters, fused layout (winograd_pack_fused_kernel)
  const float* aux0;          // MODE 0: bias [N] (nullable); MODE 1: x_in [B,H,W,N], masks the result (nullable)
  const float* aux1;          // MODE 1: addend [B,H,W,N] (nullable)
  float* y;                   // [B,H,W,N] (MODE 0: nullable when only the pool is wanted)
  float* ypool;               // MODE 0: [B,H/2,W/2,N] (nullable)
  const uint32_t* pool_bits;  // POOLED: ReLU bit cache of the operand's layer (words [T][K/2]), or
  const float* xmask;         // POOLED: that layer's output [B,H,W,K] (x > 0 is the mask) when there is no cache
  uint32_t* in_bits;          // MODE 0: mask of x, written (nullable); MODE 1: mask of x_in, read instead of aux0 (nullable)
  uint32_t* out_bits;         // MODE 0: mask of y, written (nullable)
  int B, H, W, TH, TW;
  int64_t T;
  int relu, runs;
  unsigned long long* prof = nullptr;   // -DNFS_ABLATE builds: per-wave phase cycle sums (nfs_fused_prof)
  int dbg = 0;                          // NFS_FUSED_DBG timing ablations
};

// filters: U [36][K/32][N][32] (winograd_pack4_kernel) -> Uf [K/16][2][18][N/16][64 lanes][zl 2][s 2]; lane (g = l>>4,
// c = l&15), element (zl, s) = U_z[k = 16 j + 4 g + 2 hh + s][n = 16 w + c], z = 2 zp + zl: the B operands of one
// wave's four MFMAs for the component pair zp and half hh of k-slice j in one float4
__global__ void __launch_bounds__(256) winograd_pack_fused_kernel(const float* __restrict__ up, float* __restrict__ uf,
                                                                  int K, int N) {
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= (int64_t)36 * K * N) return;
  const int n = (int)(gid % N), k = (int)((gid / N) % K), z = (int)(gid / ((int64_t)N * K));
  const float u = up[(((int64_t)z * (K / 32) + k / 32) * N + n) * 32 + (k & 31)];
  const int j = k >> 4, g = (k >> 2) & 3, hh = (k >> 1) & 1, s = k & 1;
  const int w = n >> 4, c = n & 15;
  uf[((((int64_t)(j * 2 + hh) * 18 + (z >> 1)) * (N / 16) + w) * 64 + g * 16 + c) * 4 + (z & 1) * 2 + s] = u;
}

struct WfTile { int b, ty, tx; };
// (tile counts fit 32 bits: the callers bound B*H*W; 64-bit division is a few hundred instructions per call here)
__device__ __forceinline__ WfTile wf_tile(uint32_t tile, uint32_t TH, uint32_t TW) {
  WfTile t;
  const uint32_t row = tile / TW;
  t.tx = (int)(tile - row * TW);
  t.b = (int)(row / TH);
  t.ty = (int)(row - (uint32_t)t.b * TH);
  return t;
}

// one staged value: 4 channels (ch .. ch+3) of patch pixel p = 6 r + s of a tile; loads are unconditional (clamped
// coordinates) and the out-of-image zero is a select, or the compiler drains vmcnt at the join
template <int K, bool POOLED>
__device__ __forceinline__ float4 wf_fetch(const WfArgs& a, const WfTile& t, int r, int s, int ch) {
  const int yy = 4 * t.ty - 1 + r, xx = 4 * t.tx - 1 + s;
  if (!POOLED) {
    const bool ok = yy >= 0 && yy < a.H && xx >= 0 && xx < a.W;
    const int yc = min(max(yy, 0), a.H - 1), xc = min(max(xx, 0), a.W - 1);
    const float4 v = *reinterpret_cast<const float4*>(a.x + (((int64_t)t.b * a.H + yc) * a.W + xc) * K + ch);
    const float m = ok ? 1.f : 0.f;     // (a multiply, not a select: hipcc turns `ok ? load : 0` into a branch around the load)
    return make_float4(m * v.x, m * v.y, m * v.z, m * v.w);
  } else {
    const int PH = a.H >> 1, PW = a.W >> 1;
    const bool ok = yy >= 0 && xx >= 0 && (yy >> 1) < PH && (xx >> 1) < PW;
    const int yc = min(max(yy, 0), 2 * PH - 1), xc = min(max(xx, 0), 2 * PW - 1);
    const float4 g = *reinterpret_cast<const float4*>(a.x + (((int64_t)t.b * PH + (yc >> 1)) * PW + (xc >> 1)) * K + ch);
    const float q = ok ? 0.25f : 0.f;   // (folded into the scale: no select on a loaded value, see above)
    if (a.pool_bits) {
      const uint2 wd = *reinterpret_cast<const uint2*>(
          a.pool_bits + (((int64_t)t.b * a.TH + (yc >> 2)) * a.TW + (xc >> 2)) * (K >> 1) + (ch >> 1));
      const int sh = ((yc & 3) * 4 + (xc & 3)) * 2;
      const uint32_t m0 = wd.x >> sh, m1 = wd.y >> sh;
      return make_float4((m0 & 1u) ? q * g.x : 0.f, (m0 & 2u) ? q * g.y : 0.f, (m1 & 1u) ? q * g.z : 0.f,
                         (m1 & 2u) ? q * g.w : 0.f);
    }
    const float4 m = *reinterpret_cast<const float4*>(a.xmask + (((int64_t)t.b * a.H + yc) * a.W + xc) * K + ch);
    return make_float4(m.x > 0.f ? q * g.x : 0.f, m.y > 0.f ? q * g.y : 0.f, m.z > 0.f ? q * g.z : 0.f,
                       m.w > 0.f ? q * g.w : 0.f);
  }
}

template <int K, int N, int MODE, bool POOLED>
__global__ void __launch_bounds__(256, 2) winograd_fused_kernel(WfArgs a) {
  constexpr int NIT = K / 16;         // k-slices of 16 input channels
  constexpr int NWT = N / 16;         // column tiles of the layer (a block takes four: blockIdx.y = 64-channel group)
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* const Pb = smem;                  // [36 pixels][16 tiles][16 channels]  staged patches of one slice
  float* const Vb = smem + WF_BUF;         // [36 components][16 tiles][16 channels, float2 slots swizzled]  B^T d B
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, wg = 4 * blockIdx.y + w;
  // XCD-aware order (see winograd_input4_kernel): neighbouring runs share patch rows and the filter stream
  const int per_xcd = gridDim.x / 8;
  const int run = (blockIdx.x % 8) * per_xcd + blockIdx.x / 8;
  if (run >= a.runs) return;
  const int64_t tile0 = (int64_t)run * 16;

  // Three roles per lane.
  // staging: (tile ts, channel quad gs) -- adjacent lanes read adjacent 16 bytes; wave w takes pixels w, w+4, ...
  const int ts = lane >> 2, gs = lane & 3;
  const int64_t stile = tile0 + ts < a.T ? tile0 + ts : a.T - 1;
  const WfTile st = wf_tile((uint32_t)stile, a.TH, a.TW);
  // transform: one (tile tt, channel tc) per lane and slice: wave w owns tiles 4w .. 4w+3, so its reads of a patch pixel
  // and its writes of a component are 64 consecutive dwords
  const int tt = 4 * w + (lane >> 4), tc = lane & 15;
  const int mw = (w & 1) | ((w & 2) << 1);                      // float2-slot swizzle of V rows 4w .. 4w+3
  const int v_wr = tt * 16 + 2 * ((tc >> 1) ^ mw) + (tc & 1);
  // MFMA A operand: lane (row t = tile, k group g): channels 4 g + 2 hh + {0, 1} of the slice = float2 slot 2 g + hh,
  // swizzled by the row's m so that the 32 lanes of a b64 read group cover all 64 banks
  const int t = lane & 15, g = lane >> 4;
  const int mt = ((t >> 2) & 1) | ((t >> 2) & 2) << 1;
  const int v_rd0 = t * 16 + 2 * ((2 * g) ^ mt), v_rd1 = t * 16 + 2 * ((2 * g + 1) ^ mt);

  f32x4 acc[36];
#pragma unroll
  for (int z = 0; z < 36; ++z) acc[z] = f32x4{0.f, 0.f, 0.f, 0.f};
#ifdef NFS_ABLATE
  unsigned long long pt[6] = {0, 0, 0, 0, 0, 0};
  unsigned long long tk = a.prof ? clock64() : 0, t_begin = tk;
#define NFS_TICK(i_) if (a.prof) { const unsigned long long n_ = clock64(); pt[i_] += n_ - tk; tk = n_; }
#else
#define NFS_TICK(i_)
#endif

  // the k-slices are walked from a block-dependent start: all CUs streaming the same filter lines at the same moment
  // queue on the same L2 channels
  const int j0 = run % NIT;
  const int64_t ctile = tile0 + tt < a.T ? tile0 + tt : a.T - 1;      // tile of the transform role (in_bits)

  // patch slice -> registers (9 pixels per wave) / registers -> LDS
  auto fetch = [&](float4* v, int slice) {
#pragma unroll
    for (int i = 0; i < 9; ++i) {
      const int p = w + 4 * i;
      v[i] = wf_fetch<K, POOLED>(a, st, p / 6, p % 6, 16 * slice + 4 * gs);
    }
  };
  auto stash = [&](const float4* v, float* P) {
#pragma unroll
    for (int i = 0; i < 9; ++i) *reinterpret_cast<float4*>(P + (w + 4 * i) * WF_PXF + lane * 4) = v[i];
  };
  // B^T d B of the lane's (tile, channel): P -> V
  auto transform = [&](const float* P, float* V, int slice) {
    float tcol[6][6];
    uint32_t word = 0u;
#pragma unroll
    for (int s = 0; s < 6; ++s) {
      float d[6];
#pragma unroll
      for (int r = 0; r < 6; ++r) {
        d[r] = P[(r * 6 + s) * WF_PXF + w * 64 + lane];
        if (MODE == 0 && !POOLED && r >= 1 && r <= 4 && s >= 1 && s <= 4)
          word |= (d[r] > 0.f ? 1u : 0u) << (((r - 1) * 4 + (s - 1)) * 2);
      }
      wg4_bt1(d, tcol[s]);
    }
    if (MODE == 0 && !POOLED && a.in_bits) {
      // word of the bit cache = two channels (even, odd) of one tile: pair with the neighbouring lane
      const uint32_t pw = __shfl_xor(word, 1, 64);
      if (blockIdx.y == 0 && !(tc & 1) && tile0 + tt < a.T)
        a.in_bits[ctile * (K >> 1) + 8 * slice + (tc >> 1)] = word | (pw << 1);
    }
#pragma unroll
    for (int r = 0; r < 6; ++r) {
      const float row[6] = {tcol[0][r], tcol[1][r], tcol[2][r], tcol[3][r], tcol[4][r], tcol[5][r]};
      float o[6];
      wg4_bt1(row, o);
#pragma unroll
      for (int q = 0; q < 6; ++q) V[(r * 6 + q) * WF_PXF + v_wr] = o[q];
    }
  };

  // Per slice: patch -> LDS | barrier | transform P -> V | barrier | 144 MFMAs from V.  One buffer each: the phases of a
  // block are serial, and it is the OTHER block resident on the CU (two fit: 72 KB of LDS and <= 256 registers each)
  // whose MFMAs run under this block's loads, transform and epilogue.
  const float4* ub0 = a.Uf + (int64_t)wg * 64 + lane;
  constexpr int64_t ZS = (int64_t)NWT * 64;                        // float4 stride between component pairs
  {
    float4 v[9];
    fetch(v, j0);
    stash(v, Pb);
  }
  __syncthreads();
  transform(Pb, Vb, j0);
  __syncthreads();
  NFS_TICK(0)
#pragma unroll 1
  for (int j = 0; j < NIT; ++j) {
    const int jj = (j0 + j) % NIT, j1 = (j0 + j + 1) % NIT;
    // the next slice's patches travel under this slice's MFMAs (after the last slice: a fetch nobody uses -- no branch
    // around the loads)
    float4 v[9];
    fetch(v, j1);
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
      const float4* ub = ub0 + (int64_t)(2 * jj + hh) * 18 * ZS;
      const int rd = hh == 0 ? v_rd0 : v_rd1;
      // six components at a time: B fragments (3 x 16 bytes) and A fragments (6 x 8 bytes) of the next group are in
      // flight under the 12 MFMAs of this one
      float4 bq[2][3];
      float2 A[2][6];
#pragma unroll
      for (int i = 0; i < 3; ++i) bq[0][i] = ub[i * ZS];
#pragma unroll
      for (int i = 0; i < 6; ++i) A[0][i] = *reinterpret_cast<const float2*>(Vb + i * WF_PXF + rd);
#pragma unroll
      for (int zg = 0; zg < 6; ++zg) {
        const int c = zg & 1, n = c ^ 1;
        if (zg + 1 < 6) {
#pragma unroll
          for (int i = 0; i < 3; ++i) bq[n][i] = ub[(3 * (zg + 1) + i) * ZS];
#pragma unroll
          for (int i = 0; i < 6; ++i) A[n][i] = *reinterpret_cast<const float2*>(Vb + (6 * (zg + 1) + i) * WF_PXF + rd);
        }
        // (the two k-steps of a component are issued 6 MFMAs apart: back-to-back MFMAs on one accumulator wait for
        // each other)
#pragma unroll
        for (int i = 0; i < 6; ++i) {
          const float4 b4 = bq[c][i >> 1];
          acc[6 * zg + i] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[c][i].x, (i & 1) ? b4.z : b4.x, acc[6 * zg + i], 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 6; ++i) {
          const float4 b4 = bq[c][i >> 1];
          acc[6 * zg + i] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[c][i].y, (i & 1) ? b4.w : b4.y, acc[6 * zg + i], 0, 0, 0);
        }
      }
    }
    NFS_TICK(1)
    stash(v, Pb);             // (P was consumed by the transform before this slice's MFMAs)
    __syncthreads();          // ... and V by the MFMAs above
    NFS_TICK(2)
    if (j + 1 < NIT) transform(Pb, Vb, j1);
    __syncthreads();
    NFS_TICK(3)
  }

  // ---- output transform + layer epilogue, all in-lane: tiles 4 g + {0..3} of the run (as two pairs), channel n0 ----
  // (H and W are multiples of 4 on this path: every pixel of a live tile is inside the image)
  const int n0 = 16 * wg + t;          // (t = lane & 15 is the C column)
  const int N2 = N >> 1;
  const int rowN = a.W * N;
  const float bias = (MODE == 0 && a.aux0) ? a.aux0[n0] : 0.f;
  const uint32_t* mbits = (MODE == 1 && a.aux0) ? a.in_bits : nullptr;
#pragma unroll
  for (int rp = 0; rp < 2; ++rp) {
    float2 o[4][4];
    {
      float2 tcl[6][4];
#pragma unroll
      for (int s = 0; s < 6; ++s) {
        float2 m[6];
#pragma unroll
        for (int q = 0; q < 6; ++q) m[q] = make_float2(acc[q * 6 + s][2 * rp], acc[q * 6 + s][2 * rp + 1]);
        wg4_at(m, tcl[s]);
      }
#pragma unroll
      for (int aa = 0; aa < 4; ++aa) {
        const float2 row[6] = {tcl[0][aa], tcl[1][aa], tcl[2][aa], tcl[3][aa], tcl[4][aa], tcl[5][aa]};
        wg4_at(row, o[aa]);
      }
    }
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int64_t tile = tile0 + 4 * g + 2 * rp + e;
      const bool live = tile < a.T;
      const WfTile ot = wf_tile((uint32_t)(live ? tile : a.T - 1), a.TH, a.TW);
      const int64_t base = (((int64_t)ot.b * a.H + 4 * ot.ty) * a.W + 4 * ot.tx) * N + n0;
      float vv[16];
#pragma unroll
      for (int px = 0; px < 16; ++px) vv[px] = e == 0 ? o[px >> 2][px & 3].x : o[px >> 2][px & 3].y;
      if (MODE == 0) {
        uint32_t wd = 0u;
#pragma unroll
        for (int px = 0; px < 16; ++px) {
          float v = vv[px] + bias;
          if (a.relu) v = fmaxf(v, 0.f);
          vv[px] = v;
          wd |= (v > 0.f ? 1u : 0u) << (px * 2);
        }
        if (a.y && live) {
          float* yt = a.y + base;
#pragma unroll
          for (int px = 0; px < 16; ++px) yt[(px >> 2) * rowN + (px & 3) * N] = vv[px];
        }
        if (a.out_bits) {
          // word = two channels (even, odd) of one tile: pair with the neighbouring lane
          const uint32_t pw = __shfl_xor(wd, 1, 64);
          if (live && !(n0 & 1)) a.out_bits[tile * N2 + (n0 >> 1)] = wd | (pw << 1);
        }
        if (a.ypool && live) {
          // slim.avg_pool2d [2,2] VALID: a 4x4 output tile holds 2x2 complete pooling windows
          const int PW = a.W >> 1;
          float* yp = a.ypool + (((int64_t)ot.b * (a.H >> 1) + 2 * ot.ty) * PW + 2 * ot.tx) * N + n0;
#pragma unroll
          for (int pa = 0; pa < 2; ++pa)
#pragma unroll
            for (int pc = 0; pc < 2; ++pc)
              yp[(pa * PW + pc) * N] = 0.25f * (vv[(2 * pa) * 4 + 2 * pc] + vv[(2 * pa) * 4 + 2 * pc + 1] +
                                                vv[(2 * pa + 1) * 4 + 2 * pc] + vv[(2 * pa + 1) * 4 + 2 * pc + 1]);
        }
      } else if (live) {
        uint32_t wd = mbits ? (mbits[tile * N2 + (n0 >> 1)] >> (n0 & 1)) : 0x55555555u;
        float ad[16];
        if (a.aux1) {
          const float* at = a.aux1 + base;
#pragma unroll
          for (int px = 0; px < 16; ++px) ad[px] = at[(px >> 2) * rowN + (px & 3) * N];
        } else {
#pragma unroll
          for (int px = 0; px < 16; ++px) ad[px] = 0.f;
        }
        if (!mbits && a.aux0) {
          const float* xt = a.aux0 + base;
          wd = 0u;
#pragma unroll
          for (int px = 0; px < 16; ++px) wd |= (xt[(px >> 2) * rowN + (px & 3) * N] > 0.f ? 1u : 0u) << (px * 2);
        }
        float* yt = a.y + base;
#pragma unroll
        for (int px = 0; px < 16; ++px) {
          float v = vv[px];
          if (a.relu) v += ad[px];            // addend not yet through the mask: add first
          v = ((wd >> (px * 2)) & 1u) ? v : 0.f;
          if (!a.relu) v += ad[px];
          yt[(px >> 2) * rowN + (px & 3) * N] = v;
        }
      }
    }
  }
#ifdef NFS_ABLATE
  NFS_TICK(4)
  if (a.prof && lane == 0) {
    unsigned long long* o = a.prof + ((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 4 + w) * 8;
    for (int i = 0; i < 6; ++i) o[i] = pt[i];
    o[6] = t_begin;
    o[7] = tk;
  }
#endif
#undef NFS_TICK
}

// ---- host side ------------------------------------------------------------------------------------------------
#ifdef NFS_ABLATE
static unsigned long long* g_fused_prof = nullptr;   // measurement builds only (nfs_fused_prof)
#endif
bool winograd_fusable(int K, int N) {
  static const bool off = [] { const char* e = getenv("NFS_WG_FUSED"); return e && atoi(e) == 0; }();
  return !off && (K == 64 || K == 128) && (N == 64 || N == 128);
}
// ... and image sizes: whole 4x4 tiles only (the epilogue has no per-pixel bounds checks)
bool winograd_fused_takes(int H, int W) { return H >= 4 && W >= 4 && H % 4 == 0 && W % 4 == 0; }

int64_t winograd_fused_packed_floats(int K, int N) { return winograd_fusable(K, N) ? (int64_t)36 * K * N : 0; }

int winograd_pack_fused(const float* up, float* uf, int K, int N, hipStream_t s) {
  hipLaunchKernelGGL(winograd_pack_fused_kernel, dim3(blocks_for((int64_t)36 * K * N, 256)), dim3(256), 0, s, up, uf, K, N);
  return check_launch("winograd_pack_fused");
}

template <int K, int N, int MODE, bool POOLED>
static void launch_fused(const WfArgs& a, hipStream_t s) {
  const size_t lds = 2 * WF_BUF * sizeof(float);
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(winograd_fused_kernel<K, N, MODE, POOLED>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_done = true;
  }
  const int grid = (a.runs + 7) / 8 * 8;
  hipLaunchKernelGGL((winograd_fused_kernel<K, N, MODE, POOLED>), dim3(grid, N / 64), dim3(256), lds, s, a);
}

template <int K, int N>
static void launch_fused_kn(const WfArgs& a, int mode, bool pooled, hipStream_t s) {
  if (mode == 0) launch_fused<K, N, 0, false>(a, s);
  else if (pooled) launch_fused<K, N, 1, true>(a, s);
  else launch_fused<K, N, 1, false>(a, s);
}

// same contract as winograd_conv (winograd.hip) for the shapes winograd_fusable() / winograd_fused_takes() accept; Uf from
// winograd_pack_fused.  Whether a ReLU bit cache is given or the float masks does not change the path: the two forms
// stay bit-identical (test_conv_relu_bit_cache_equals_float_masks).
int winograd_fused_conv(const float* x, const float* Uf, const float* aux0, const float* aux1, float* y, int B, int H,
                        int W, int K, int N, int mode, int relu, hipStream_t s, float* ypool, const float* xmask,
                        uint32_t* in_bits, uint32_t* out_bits, bool pooled_grad) {
  WfArgs a;
#ifdef NFS_ABLATE
  a.prof = g_fused_prof;
#endif
  a.x = x;
  a.Uf = reinterpret_cast<const float4*>(Uf);
  a.aux0 = aux0;
  a.aux1 = aux1;
  a.y = y;
  a.ypool = mode == 0 ? ypool : nullptr;
  a.pool_bits = pooled_grad ? out_bits : nullptr;
  a.xmask = xmask;
  a.in_bits = in_bits;
  a.out_bits = mode == 0 ? out_bits : nullptr;
  a.B = B; a.H = H; a.W = W;
  a.TH = (H + 3) / 4; a.TW = (W + 3) / 4;
  a.T = (int64_t)B * a.TH * a.TW;
  a.relu = relu;
  a.runs = (int)((a.T + 15) / 16);
  if (K == 64 && N == 64) launch_fused_kn<64, 64>(a, mode, pooled_grad, s);
  else if (K == 64 && N == 128) launch_fused_kn<64, 128>(a, mode, pooled_grad, s);
  else if (K == 128 && N == 64) launch_fused_kn<128, 64>(a, mode, pooled_grad, s);
  else launch_fused_kn<128, 128>(a, mode, pooled_grad, s);
  return check_launch("winograd_fused_conv");
}

}  // namespace nfs

#ifdef NFS_ABLATE
// measurement builds only: per-wave phase cycle sums of winograd_fused_kernel, 8 x uint64 per (block, wave):
// [prologue, patch read + transform, MFMAs, staging store + barrier, epilogue, -, t_begin, t_end]
extern "C" int nfs_fused_prof(void* buf) { nfs::g_fused_prof = reinterpret_cast<unsigned long long*>(buf); return 0; }
#endif
