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

#include <mutex>
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
  uint32_t x_bytes, m_bytes;  // sizes of x and of the POOLED mask operand (buffer-load range checks)
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

// Staged patches come through buffer loads: one 32-bit byte offset per (lane, pixel), fixed for the whole kernel, plus a
// scalar offset for the k-slice -- no 64-bit address arithmetic in the loop -- and an offset beyond num_records for the
// pixels outside the image, which the hardware answers with zeros (no select, no branch around the load).
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
constexpr uint32_t WF_OOB = 0x80000000u;       // the launcher keeps every buffer below 2 GB
__device__ __forceinline__ float4 wf_ld4(__amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t soff) {
  return __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}
__device__ __forceinline__ uint2 wf_ld2u(__amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t soff) {
  return __builtin_bit_cast(uint2, __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0));
}

// POOLED: 0 plain operand; 1 pooled gradient masked from the ReLU bit cache; 2 ... from the float output (a template
// parameter: a run-time choice is a branch around loads inside the MFMA stream)
// RAG: H or W is not a multiple of 4 -- the last tile row / column hangs over the image.  Staging already answers the
// pixels outside with zeros (the halo's mechanism); the epilogue then tests every pixel of a tile against a 16-bit
// validity mask (stores, addend / mask loads, pooling windows).  A template parameter: the aligned form keeps its
// epilogue without per-pixel branches (they cost 13 % of the kernel when they were unconditional).
template <int K, int N, int MODE, int POOLED, bool RAG>
__global__ void __launch_bounds__(256, 2) winograd_fused_kernel(WfArgs a) {
  constexpr int NIT = K / 16;         // k-slices of 16 input channels
  constexpr int NWT = N / 16;         // column tiles of the layer (a block takes four: blockIdx.y = 64-channel group)
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* const Pb = smem;                  // [36 pixels][16 tiles][16 channels]  staged patches of one slice
  float* const Vb = smem + WF_BUF;         // [36 components][16 tiles][16 channels, float2 slots swizzled]  B^T d B
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6), wg = 4 * blockIdx.y + w;      // wave-uniform: scalar registers
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
  unsigned long long tk = a.prof ? clock64() : 0;
  const unsigned long long t_begin = a.prof ? wall_clock64() : 0;   // (100 MHz, one counter for the whole chip)
#define NFS_TICK(i_) if (a.prof) { const unsigned long long n_ = clock64(); pt[i_] += n_ - tk; tk = n_; }
#else
#define NFS_TICK(i_)
#endif

  // (walking the k-slices from a block-dependent start, so that the CUs do not all stream the same filter lines at the
  // same moment, was measured: no gain -- and the k order of a tile's sums would then depend on the batch it sits in)
  constexpr int j0 = 0;
  const int64_t ctile = tile0 + tt < a.T ? tile0 + tt : a.T - 1;      // tile of the transform role (in_bits)

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
#ifdef NFS_K7F_EMUL_BF16
      // ... and the limb split of the transform's 36 outputs (three cvt_pk levels per pair), folded back so that it stays
#pragma unroll
      for (int q = 0; q < 6; q += 2) {
        typedef float emu_f2 __attribute__((ext_vector_type(2)));
        typedef __bf16 emu_b2 __attribute__((ext_vector_type(2)));
        const emu_f2 x = {o[q], o[q + 1]};
        const emu_b2 bh = __builtin_convertvector(x, emu_b2);
        const emu_f2 r1 = x - __builtin_convertvector(bh, emu_f2);
        const emu_b2 bm = __builtin_convertvector(r1, emu_b2);
        const emu_f2 r2 = r1 - __builtin_convertvector(bm, emu_f2);
        const emu_b2 bl = __builtin_convertvector(r2, emu_b2);
        const emu_f2 back = __builtin_convertvector(bh, emu_f2) + __builtin_convertvector(bm, emu_f2) + __builtin_convertvector(bl, emu_f2);
        o[q] = back.x; o[q + 1] = back.y;
      }
#endif
#pragma unroll
      for (int q = 0; q < 6; ++q) V[(r * 6 + q) * WF_PXF + v_wr] = o[q];
    }
  };

  // Per slice: [barrier] transform P -> V [barrier] 144 MFMAs from V, in 12 groups of 6 components x 2 k-steps.  Under
  // the MFMAs: the B fragments of the group three ahead (a ring of four groups: ~1150 cycles of cover for the L2
  // latency; three / two ahead in the pooled forms; the ring runs on across slices), the A fragments of the next group, and the next slice's patches in
  // three batches that go to P as they arrive (P is free once the slice is transformed).  One buffer each for P and V:
  // two blocks fit a CU (72 KB of LDS, <= 256 registers) and fill each other's barrier and epilogue gaps.
  const __amdgpu_buffer_rsrc_t u_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float4*>(a.Uf), 0, 36u * K * N * 4u, 0x00020000);
  const uint32_t uo = (uint32_t)(wg * 64 + lane) * 16u;
  constexpr uint32_t ZS = NWT * 64 * 16;                           // bytes between component pairs
  // staging offsets of this lane's nine pixels (p = w + 4 i; wave-uniform, so r, s and the mask shift are scalars)
  const __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), 0, a.x_bytes, 0x00020000);
  [[maybe_unused]] const __amdgpu_buffer_rsrc_t m_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      POOLED == 1 ? (void*)const_cast<uint32_t*>(a.pool_bits) : POOLED == 2 ? (void*)const_cast<float*>(a.xmask) : (void*)const_cast<float*>(a.x),
      0, a.m_bytes, 0x00020000);
  // per lane: the offset of its tile's origin in each operand and one validity bit per pixel; the pixel's displacement
  // is a scalar (p, r, s are wave-uniform), so a load address costs an add and a select and no registers are held
  const int PHh = a.H >> 1, PWh = a.W >> 1;
  const uint32_t xb = !POOLED ? (uint32_t)(((st.b * a.H + 4 * st.ty) * a.W + 4 * st.tx) * K + 4 * gs) * 4u
                              : (uint32_t)(((st.b * PHh + 2 * st.ty) * PWh + 2 * st.tx) * K + 4 * gs) * 4u;
  const uint32_t mb =
      POOLED == 1 ? (uint32_t)(((st.b * a.TH + st.ty) * a.TW + st.tx) * (K >> 1) + 2 * gs) * 4u
                  : (uint32_t)(((st.b * a.H + 4 * st.ty) * a.W + 4 * st.tx) * K + 4 * gs) * 4u;
  uint32_t okm = 0u;
#pragma unroll
  for (int i = 0; i < 9; ++i) {
    const int p = w + 4 * i, r = p / 6, sx = p - 6 * r;
    const int yy = 4 * st.ty - 1 + r, xx = 4 * st.tx - 1 + sx;
    const bool ok = !POOLED ? (yy >= 0 && yy < a.H && xx >= 0 && xx < a.W)
                            : (yy >= 0 && xx >= 0 && (yy >> 1) < PHh && (xx >> 1) < PWh);
    okm |= (ok ? 1u : 0u) << i;
  }
  // displacement of pixel (r, sx) from the tile origin, in bytes of the operand (floor division for the -1 row / column)
  auto x_disp = [&](int r, int sx) {
    return !POOLED ? ((r - 1) * a.W + (sx - 1)) * K * 4 : (((r + 1) >> 1) - 1) * PWh * K * 4 + (((sx + 1) >> 1) - 1) * K * 4;
  };
  auto m_disp = [&](int r, int sx) {
    return POOLED == 1 ? ((((r + 3) >> 2) - 1) * a.TW + (((sx + 3) >> 2) - 1)) * (K >> 1) * 4
                       : ((r - 1) * a.W + (sx - 1)) * K * 4;
  };
  // a batch of three pixels: raw loads now, mask arithmetic (POOLED) and the LDS store when they have arrived
  struct Raw { float4 g; float4 m; };       // m: POOLED 1 -> .x/.y carry the two mask words; POOLED 2 -> the float output
  auto fetch3 = [&](Raw* v, int batch, int slice) {
    // (opaque copies: otherwise hipcc hoists the nine / eighteen per-pixel offsets out of the slice loop and spills)
    uint32_t xb_ = xb, mb_ = mb, okm_ = okm;
    asm volatile("" : "+v"(xb_), "+v"(mb_), "+v"(okm_));
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int ii = 3 * batch + i;
      const int p = w + 4 * ii, r = p / 6, sx = p - 6 * r;
      const bool ok = (okm_ >> ii) & 1u;
      const uint32_t xo = ok ? xb_ + (uint32_t)x_disp(r, sx) : WF_OOB;
      [[maybe_unused]] const uint32_t mo = ok ? mb_ + (uint32_t)m_disp(r, sx) : WF_OOB;
      v[i].g = wf_ld4(x_rsrc, xo, (uint32_t)slice * 64u);
      if (POOLED == 1) {
        const uint2 wd = wf_ld2u(m_rsrc, mo, (uint32_t)slice * 32u);
        v[i].m.x = __uint_as_float(wd.x);
        v[i].m.y = __uint_as_float(wd.y);
      } else if (POOLED == 2) {
        v[i].m = wf_ld4(m_rsrc, mo, (uint32_t)slice * 64u);
      }
    }
  };
  auto stash3 = [&](const Raw* v, int batch) {
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int ii = 3 * batch + i;
      float4 d = v[i].g;
      if (POOLED == 1) {
        // d = 0.25 * gpool[y/2, x/2] * (x_out > 0): mask bits of the pixel's own tile, two channels per word
        const int p = w + 4 * ii, r = p / 6, sx = p - 6 * r;
        const int sh = (((r + 3) & 3) * 4 + ((sx + 3) & 3)) * 2;
        const uint32_t m0 = __float_as_uint(v[i].m.x) >> sh, m1 = __float_as_uint(v[i].m.y) >> sh;
        d = make_float4((m0 & 1u) ? 0.25f * d.x : 0.f, (m0 & 2u) ? 0.25f * d.y : 0.f, (m1 & 1u) ? 0.25f * d.z : 0.f,
                        (m1 & 2u) ? 0.25f * d.w : 0.f);
      } else if (POOLED == 2) {
        const float4 m = v[i].m;
        d = make_float4(m.x > 0.f ? 0.25f * d.x : 0.f, m.y > 0.f ? 0.25f * d.y : 0.f, m.z > 0.f ? 0.25f * d.z : 0.f,
                        m.w > 0.f ? 0.25f * d.w : 0.f);
      }
      *reinterpret_cast<float4*>(Pb + (w + 4 * ii) * WF_PXF + lane * 4) = d;
    }
  };
  {
    Raw v[3][3];
#pragma unroll
    for (int bt = 0; bt < 3; ++bt) fetch3(v[bt], bt, j0);
#pragma unroll
    for (int bt = 0; bt < 3; ++bt) stash3(v[bt], bt);
  }
  constexpr int RING = POOLED ? 3 : 4;   // groups of B fragments in registers (the pooled forms hold two raw operands per staged pixel)
  float4 bq[RING][3];
#pragma unroll
  for (int gi = 0; gi < RING - 1; ++gi)
#pragma unroll
    for (int i = 0; i < 3; ++i) bq[gi][i] = wf_ld4(u_rsrc, uo, (uint32_t)((2 * j0) * 18 + 3 * gi + i) * ZS);
  __syncthreads();
  transform(Pb, Vb, j0);
  __syncthreads();
  NFS_TICK(0)
#pragma unroll 1
  for (int j = 0; j < NIT; ++j) {
    const int jj = j, j1 = (j + 1) % NIT;                          // this slice, the next (after the last: fetched, unused)
    float2 A[2][6];
#pragma unroll
    for (int i = 0; i < 6; ++i) A[0][i] = *reinterpret_cast<const float2*>(Vb + i * WF_PXF + v_rd0);
    Raw v[3];
#pragma unroll
    for (int gi = 0; gi < 12; ++gi) {
      {  // B fragments of group gi + RING - 1
        const int gn = (gi + RING - 1) % 12, sl = gi + RING - 1 < 12 ? jj : j1;
#pragma unroll
        for (int i = 0; i < 3; ++i)
          bq[(gi + RING - 1) % RING][i] = wf_ld4(u_rsrc, uo, (uint32_t)((2 * sl + gn / 6) * 18 + 3 * (gn % 6) + i) * ZS);
      }
      if (gi + 1 < 12) {  // A fragments of group gi + 1
        const int gn = gi + 1, rd = gn < 6 ? v_rd0 : v_rd1;
#pragma unroll
        for (int i = 0; i < 6; ++i)
          A[gn & 1][i] = *reinterpret_cast<const float2*>(Vb + (6 * (gn % 6) + i) * WF_PXF + rd);
      }
      if (gi == 4 || gi == 7 || gi == 10) stash3(v, gi / 3 - 1);       // (issued three groups ago)
      if (gi == 1 || gi == 4 || gi == 7) fetch3(v, gi / 3, j1);
      __builtin_amdgcn_sched_barrier(0);     // keep the loads above the MFMAs they travel under (hipcc sinks them to their uses)
      // (the two k-steps of a component are issued 6 MFMAs apart: back-to-back MFMAs on one accumulator wait for
      // each other)
      const int zb = 6 * (gi % 6);
#ifdef NFS_K7F_EMUL_BF16
      // TIMING-ONLY emulation (wrong results by construction; tools/k7f_bf16_emul.sh, never in the product build): what the
      // slice loop would cost on the bf16 pipe -- a 16-deep slice of 36 components is 36 x 3 = 108 v_mfma_f32_16x16x32_bf16
      // of 16 cycles (nine per group here) instead of 144 v_mfma_f32_16x16x4_f32 of 32 -- with everything else as it is
      // (the filter stream of the f32 form: 2/3 of what limb planes at 16 tiles would need, 4/3 of the component-split
      // form's; two blocks per CU)
      {
        typedef __bf16 emu_b8 __attribute__((ext_vector_type(8)));
        const emu_b8 ea = __builtin_bit_cast(emu_b8, bq[gi % RING][0]), eb = __builtin_bit_cast(emu_b8, bq[gi % RING][1]);
#pragma unroll
        for (int i = 0; i < 9; ++i)
          asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[zb + (i % 6)]) : "v"(ea), "v"(eb));
      }
      __builtin_amdgcn_sched_barrier(0);
      continue;
#endif
      // (inline asm: accumulate in place -- the builtin lets hipcc rotate the accumulators through two dozen spare
      // registers this kernel does not have)
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        const float4 b4 = bq[gi % RING][i >> 1];
        asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[zb + i]) : "v"(A[gi & 1][i].x), "v"((i & 1) ? b4.z : b4.x));
      }
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        const float4 b4 = bq[gi % RING][i >> 1];
        asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[zb + i]) : "v"(A[gi & 1][i].y), "v"((i & 1) ? b4.w : b4.y));
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    NFS_TICK(1)
    __syncthreads();          // P complete, V consumed
    NFS_TICK(2)
    if (j + 1 < NIT) transform(Pb, Vb, j1);
    __syncthreads();
    NFS_TICK(3)
  }

  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");   // (inline-asm MFMAs: the last one retires before an accumulator is read)
  // ---- output transform + layer epilogue, all in-lane: tiles 4 g + {0..3} of the run (as two pairs), channel n0 ----
  // (!RAG: H and W are multiples of 4, every pixel of a live tile is inside the image.  Staging the tile through
  // LDS to store 256-byte rows instead of these 64-byte pieces was measured: no gain, one more barrier.)
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
      // pixels of this tile inside the image: bit px = (px >> 2) < rows left && (px & 3) < columns left
      uint32_t vm = 0xffffu;
      if (RAG) {
        const int hy = min(4, a.H - 4 * ot.ty), wx = min(4, a.W - 4 * ot.tx);
        const uint32_t rowm = (1u << wx) - 1u;
        vm = 0u;
#pragma unroll
        for (int r = 0; r < 4; ++r) vm |= (r < hy ? rowm : 0u) << (4 * r);
      }
#define NFS_WF_IN(px_) (!RAG || ((vm >> (px_)) & 1u))
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
          for (int px = 0; px < 16; ++px)
            if (NFS_WF_IN(px)) yt[(px >> 2) * rowN + (px & 3) * N] = vv[px];
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
              // (VALID pooling: a window exists when its lower right pixel does)
              if (NFS_WF_IN((2 * pa + 1) * 4 + 2 * pc + 1))
                yp[(pa * PW + pc) * N] = 0.25f * (vv[(2 * pa) * 4 + 2 * pc] + vv[(2 * pa) * 4 + 2 * pc + 1] +
                                                  vv[(2 * pa + 1) * 4 + 2 * pc] + vv[(2 * pa + 1) * 4 + 2 * pc + 1]);
        }
      } else if (live) {
        uint32_t wd = mbits ? (mbits[tile * N2 + (n0 >> 1)] >> (n0 & 1)) : 0x55555555u;
        float ad[16];
        if (a.aux1) {
          const float* at = a.aux1 + base;
#pragma unroll
          for (int px = 0; px < 16; ++px) ad[px] = NFS_WF_IN(px) ? at[(px >> 2) * rowN + (px & 3) * N] : 0.f;
        } else {
#pragma unroll
          for (int px = 0; px < 16; ++px) ad[px] = 0.f;
        }
        if (!mbits && a.aux0) {
          const float* xt = a.aux0 + base;
          wd = 0u;
#pragma unroll
          for (int px = 0; px < 16; ++px)
            if (NFS_WF_IN(px)) wd |= (xt[(px >> 2) * rowN + (px & 3) * N] > 0.f ? 1u : 0u) << (px * 2);
        }
        float* yt = a.y + base;
#pragma unroll
        for (int px = 0; px < 16; ++px) {
          float v = vv[px];
          if (a.relu) v += ad[px];            // addend not yet through the mask: add first
          v = ((wd >> (px * 2)) & 1u) ? v : 0.f;
          if (!a.relu) v += ad[px];
          if (NFS_WF_IN(px)) yt[(px >> 2) * rowN + (px & 3) * N] = v;
        }
      }
#undef NFS_WF_IN
    }
  }
#ifdef NFS_ABLATE
  NFS_TICK(4)
  if (a.prof && lane == 0) {
    unsigned long long* o = a.prof + ((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 4 + w) * 8;
    for (int i = 0; i < 5; ++i) o[i] = pt[i];
    o[5] = pt[0] + pt[1] + pt[2] + pt[3] + pt[4];   // wave life in shader cycles
    o[6] = t_begin;
    o[7] = wall_clock64();
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
// ... and sizes: activations below 2 GB (32-bit buffer offsets); H or W off a multiple of 4 takes the RAG instance
// (per-pixel validity tests in the epilogue)
// ... and enough of them: a block walks its K / 16 slices one after the other (~4 us each), so with K = 128 and fewer
// than ~128 blocks (runs of 16 tiles x 64-channel groups) the three-kernel form, which spreads the same work over 36
// components, finishes sooner (tools/small_conv_bench.py).  A static rule, not a measurement: the two forms differ in
// rounding, and which one runs must not depend on timing noise.
bool winograd_fused_takes(int B, int H, int W, int K, int N) {
  static const bool no_rag = [] { const char* e = getenv("NFS_WG_FUSED_RAG"); return e && atoi(e) == 0; }();
  if (!(H >= 4 && W >= 4 && (int64_t)B * H * W * (K > N ? K : N) * 4 < ((int64_t)1 << 31))) return false;
  if (no_rag && (H % 4 || W % 4)) return false;          // (ablation: ragged sizes on the three-kernel form)
  const int64_t blocks = (((int64_t)B * ((H + 3) / 4) * ((W + 3) / 4) + 15) / 16) * (N / 64);
  // (the ragged instance's guarded epilogue moves the break-even up: 150 x 225, 272 blocks, 0.129 ms against 0.118 for
  // the three-kernel form; 8 x 75 x 75, 362 blocks, 0.133 against 0.140 -- tools/ragged_conv_bench.py)
  const bool ragged = (H % 4) || (W % 4);
  return K == 64 || blocks >= (ragged ? 300 : 128);
}

int64_t winograd_fused_packed_floats(int K, int N) { return winograd_fusable(K, N) ? (int64_t)36 * K * N : 0; }

int winograd_pack_fused(const float* up, float* uf, int K, int N, hipStream_t s) {
  hipLaunchKernelGGL(winograd_pack_fused_kernel, dim3(blocks_for((int64_t)36 * K * N, 256)), dim3(256), 0, s, up, uf, K, N);
  return check_launch("winograd_pack_fused");
}

template <int K, int N, int MODE, int POOLED, bool RAG>
static void launch_fused(const WfArgs& a, hipStream_t s) {
  const size_t lds = 2 * WF_BUF * sizeof(float);
  static std::once_flag attr_once;   // (one set per kernel instance, safe from several host threads)
  std::call_once(attr_once, [&] {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(winograd_fused_kernel<K, N, MODE, POOLED, RAG>),
                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); });
  const int grid = (a.runs + 7) / 8 * 8;
  hipLaunchKernelGGL((winograd_fused_kernel<K, N, MODE, POOLED, RAG>), dim3(grid, N / 64), dim3(256), lds, s, a);
}

template <int K, int N, bool RAG>
static void launch_fused_knr(const WfArgs& a, int mode, bool pooled, hipStream_t s) {
  if (mode == 0) launch_fused<K, N, 0, 0, RAG>(a, s);
  else if (!pooled) launch_fused<K, N, 1, 0, RAG>(a, s);
  else if (a.pool_bits) launch_fused<K, N, 1, 1, RAG>(a, s);
  else launch_fused<K, N, 1, 2, RAG>(a, s);
}

template <int K, int N>
static void launch_fused_kn(const WfArgs& a, int mode, bool pooled, hipStream_t s) {
  if ((a.H & 3) || (a.W & 3)) launch_fused_knr<K, N, true>(a, mode, pooled, s);
  else launch_fused_knr<K, N, false>(a, mode, pooled, s);
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

  a.x_bytes = (uint32_t)((int64_t)B * (pooled_grad ? (H / 2) * (W / 2) : H * W) * K * 4);
  a.m_bytes = !pooled_grad ? a.x_bytes : out_bits ? (uint32_t)(a.T * (K / 2) * 4) : (uint32_t)((int64_t)B * H * W * K * 4);
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
