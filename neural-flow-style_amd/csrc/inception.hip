// the operators of the reference's second loss network -- the Inception-v1 graph
// ``tensorflow_inception_graph.pb`` that styler_base.py:17-23,51-57 imports with tf.import_graph_def and addresses by
// tensor name (styler_base.py:91-94: 'conv2d2', 'mixed3b', 'mixed4b_pool_reduce_pre_relu', ...).  Node types of that
// graph on the path to its feature tensors: Conv2D (1x1, 3x3, 5x5 stride 1; 7x7 stride 2) + BiasAdd + Relu, MaxPool
// (3x3, stride 1 / 2), LRN, ConcatV2 -- all with TF's SAME padding.  Weights are frozen: backward = data gradient only.
//
//   conv2d_mfma_kernel     generic NHWC SAME convolution as an implicit GEMM on v_mfma_f32_32x32x2_f32 (exact f32):
//                          M = output pixels, N = output channels (64 per block), K = taps x input channels in
//                          16-channel chunks (a partial last chunk is zero-filled: the module widths 24, 204, 508 ...
//                          are not multiples of 16).  Input and output are SLICES of wider rows (pixel strides ldx /
//                          ldy): every branch of an inception module writes straight into its channel range of the
//                          module's concatenated output -- ConcatV2 never runs -- and the data gradient of a branch
//                          reads its range of the concatenated gradient.  The A loader can multiply by (mask > 0):
//                          the ReLU adjoint is applied where the gradient is CONSUMED, so branch gradients are summed
//                          unmasked (`accumulate`) and no masked copy is ever written.  The data gradient of a
//                          stride-1 SAME convolution is the same kernel on filters packed flipped and transposed.
//                          CMODE 1: <= 4 input channels (the 7x7 first layer on the 3-channel image): a K chunk is
//                          4 taps x 4 channels instead of 16 channels of one tap.
//   conv2d_small_dgrad     data gradient down to the <= 4-channel image (stride 1 or 2): VALU, filters in LDS.
//   maxpool3_fwd/bwd       3x3 max pool with the window position of the FIRST maximum kept as one byte per output;
//                          the adjoint is a gather over the <= 9 windows that contain a pixel (no atomics).
//   lrn_fwd/bwd            tf.nn.lrn across channels.
#include "common.h"

namespace nfs {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int IC_BN = 64;   // output channels per block
constexpr int IC_LS = 20;   // LDS row stride in floats (16 + 4: b128 fragment reads hit 16 distinct 16-byte slots)

struct Conv2dArgs {
  const float* x;       // input slice base
  const float* xmask;   // nullable: x is multiplied by (xmask > 0) while it is loaded
  const float* wp;      // packed [K chunks][Npad][16]
  const float* bias;    // nullable [Cout]
  float* y;             // output slice base
  float* ypre;          // nullable: the value before ReLU (the graph's *_pre_relu tensors)
  int ldx, ldm, ldy, ldp;
  int B, H, W, Ho, Wo, Cin, Cout, kh, kw, stride, pad_t, pad_l;
  int relu, accumulate;
  int M, nchunks, cchunks, Npad;
  float* ws;            // partial sums [ztotal][M][Npad] (partial != 0)
  int ksplit, cps;      // K chunks per split
  uint32_t x_bytes, m_bytes;   // extents of x / x_mask from their slice bases (buffer-load range checks)
  int partial;          // the tile leaves as a raw partial sum in ws; conv2d_reduce finishes (split K, or several
  int zbase, ztotal;    // convolutions summed into one output: first partial of this problem / all partials of the sum)
};

constexpr int IC_GMAX = 6;   // problems per grouped launch
struct Conv2dGroupArgs {
  Conv2dArgs p[IC_GMAX];
  int bstart[IC_GMAX + 1];   // first block of each problem
  int n;
};

template <int BM, int CMODE>
__device__ __forceinline__ void conv2d_tile(const Conv2dArgs& a, int mtile, int ntile, int zsplit, float* smem) {
  constexpr int MT = BM / 64;   // 32x32 MFMA tiles per wave along M (= A float4 per thread and chunk)
  float* As = smem;                        // [2][BM][20]
  float* Bs = smem + 2 * BM * IC_LS;       // [2][64][20]
  const int t = threadIdx.x, lane = t & 63, wid = t >> 6;
  const int wm = wid >> 1, wn = wid & 1, i = lane & 31, h = lane >> 5;
  const int m0 = mtile * BM, n0 = ntile * IC_BN;
  const int srow = t >> 2, kq = t & 3;

  int pixbase[MT], iy0[MT], ix0[MT];
  bool rvalid[MT];
#pragma unroll
  for (int r = 0; r < MT; ++r) {
    const int m = m0 + srow + 64 * r;
    rvalid[r] = m < a.M;
    const int mm = rvalid[r] ? m : 0;
    const int hw = a.Ho * a.Wo;
    const int b = mm / hw, rem = mm - b * hw;
    const int oy = rem / a.Wo, ox = rem - oy * a.Wo;
    pixbase[r] = b * a.H * a.W;
    iy0[r] = oy * a.stride - a.pad_t;
    ix0[r] = ox * a.stride - a.pad_l;
  }
  const float4* wp4 = reinterpret_cast<const float4*>(a.wp) + ((int64_t)n0 + srow) * 4 + kq;
  const int64_t slab4 = (int64_t)a.Npad * 4;

  // Operand loads (CMODE 0) go through buffer descriptors: one 32-bit byte offset per (lane, row) that changes only
  // when the tap does, plus the chunk's channel offset; a pixel outside the image (SAME padding), a row beyond M or a
  // channel beyond Cin gets an offset beyond num_records, which the hardware answers with zeros -- no select, no branch
  // around a load (a branch makes hipcc drain vmcnt at the join, i.e. serialises the prefetch).  The ReLU mask of the
  // data-gradient form is applied when the chunk is staged, not when it is requested: the select would otherwise wait
  // for the load it belongs to.  The chunk counter advances through (tap, channel chunk) without divisions.
  constexpr uint32_t OOB = 0x80000000u;           // the launcher keeps x and x_mask below 2 GB
  const __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(a.x), 0, CMODE == 0 ? a.x_bytes : 0u, 0x00020000);
  const __amdgpu_buffer_rsrc_t m_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(a.xmask ? a.xmask : a.x), 0, (CMODE == 0 && a.xmask) ? a.m_bytes : 0u, 0x00020000);
  struct Chunk { float4 x[MT]; float4 m[MT]; float4 b; };
  int ld_tap = 0, ld_cc = 0, ld_dy = 0, ld_dx = 0;          // state of the next chunk to request (wave-uniform)
  uint32_t xo[MT], mo[MT];                                  // byte offsets of the rows' pixels at the current tap
  auto set_tap = [&]() {
#pragma unroll
    for (int r = 0; r < MT; ++r) {
      const int iy = iy0[r] + ld_dy, ix = ix0[r] + ld_dx;
      const bool ok = rvalid[r] && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
      const uint32_t pix = (uint32_t)(pixbase[r] + iy * a.W + ix);
      xo[r] = ok ? pix * (uint32_t)a.ldx * 4u + 16u * kq : OOB;
      mo[r] = ok ? pix * (uint32_t)a.ldm * 4u + 16u * kq : OOB;
    }
  };
  auto seek = [&](int it) {                                  // (once per tile: the first chunk of this K split)
    ld_tap = it / a.cchunks;
    ld_cc = it - ld_tap * a.cchunks;
    ld_dy = ld_tap / a.kw;
    ld_dx = ld_tap - ld_dy * a.kw;
    set_tap();
  };
  auto load = [&](int it, Chunk& ch) {
    if (CMODE == 0) {
      const uint32_t cb = (uint32_t)ld_cc * 64u;             // 16 channels per chunk
      const bool kvalid = ld_cc * 16 + 4 * kq < a.Cin;
#pragma unroll
      for (int r = 0; r < MT; ++r) {
        ch.x[r] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(x_rsrc, kvalid ? xo[r] : OOB, cb, 0));
        if (a.xmask)
          ch.m[r] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(m_rsrc, kvalid ? mo[r] : OOB, cb, 0));
      }
      if (++ld_cc == a.cchunks) {
        ld_cc = 0;
        ++ld_tap;
        if (++ld_dx == a.kw) { ld_dx = 0; ++ld_dy; }
        set_tap();
      }
    } else {
      const int tap0 = it * 4 + kq;
      const bool kvalid = tap0 < a.kh * a.kw;
      const int tap = kvalid ? tap0 : 0;
      const int dy = tap / a.kw, dx = tap - dy * a.kw;
#pragma unroll
      for (int r = 0; r < MT; ++r) {
        const int iy = iy0[r] + dy, ix = ix0[r] + dx;
        const bool ok = rvalid[r] && kvalid && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
        const int64_t off = ok ? (int64_t)(pixbase[r] + iy * a.W + ix) : 0;
        const float* p = a.x + off * a.ldx;
        float4 v;
        v.x = p[0];
        v.y = a.Cin > 1 ? p[a.Cin > 1 ? 1 : 0] : 0.f;
        v.z = a.Cin > 2 ? p[a.Cin > 2 ? 2 : 0] : 0.f;
        v.w = a.Cin > 3 ? p[a.Cin > 3 ? 3 : 0] : 0.f;
        if (!ok) v = make_float4(0.f, 0.f, 0.f, 0.f);
        ch.x[r] = v;
      }
    }
    ch.b = wp4[(int64_t)it * slab4];
  };
  auto stage = [&](int buf, const Chunk& ch) {
#pragma unroll
    for (int r = 0; r < MT; ++r) {
      float4 v = ch.x[r];
      if (CMODE == 0 && a.xmask) {
        const float4 mk = ch.m[r];
        v.x = mk.x > 0.f ? v.x : 0.f; v.y = mk.y > 0.f ? v.y : 0.f;
        v.z = mk.z > 0.f ? v.z : 0.f; v.w = mk.w > 0.f ? v.w : 0.f;
      }
      *reinterpret_cast<float4*>(As + (buf * BM + srow + 64 * r) * IC_LS + 4 * kq) = v;
    }
    *reinterpret_cast<float4*>(Bs + (buf * IC_BN + srow) * IC_LS + 4 * kq) = ch.b;
  };

  f32x16 acc[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[mt][r] = 0.f;

  // split-K (the late layers have a few hundred pixels and K in the thousands): blockIdx.z owns a range of chunks
  const int it0 = zsplit * a.cps;
  const int it1 = it0 + a.cps < a.nchunks ? it0 + a.cps : a.nchunks;
  // three stages, two chunks of look-ahead: chunk it is multiplied from LDS while chunk it + 1 waits in registers for the
  // other LDS buffer and chunk it + 2 is in flight from memory (one chunk of look-ahead -- 512 MFMA cycles of a wave --
  // did not cover a load's latency with the 1-2 blocks per CU these layers give: conv2d2 42 -> see DESIGN.md)
  Chunk cA, cB;
  if (CMODE == 0) seek(it0);
  load(it0, cA);
  stage(0, cA);
  if (it0 + 1 < it1) load(it0 + 1, cA);
  __syncthreads();
  auto mma = [&](int buf) {
    const float* Ab = As + buf * BM * IC_LS + (wm * (BM / 2) + i) * IC_LS + 4 * h;
    const float* Bb = Bs + buf * IC_BN * IC_LS + (wn * 32 + i) * IC_LS + 4 * h;
    // a lane fetches 4 consecutive k with one b128 and feeds 4 MFMA steps with them: the k order inside the chunk is
    // permuted identically for A and B, which a sum permits
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      const float4 bq = *reinterpret_cast<const float4*>(Bb + 8 * g);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const float4 aq = *reinterpret_cast<const float4*>(Ab + mt * 32 * IC_LS + 8 * g);
        acc[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(aq.x, bq.x, acc[mt], 0, 0, 0);
        acc[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(aq.y, bq.y, acc[mt], 0, 0, 0);
        acc[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(aq.z, bq.z, acc[mt], 0, 0, 0);
        acc[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(aq.w, bq.w, acc[mt], 0, 0, 0);
      }
    }
  };
  // (unrolled by two so that the two register sets keep their names)
  for (int it = it0; it < it1; it += 2) {
    // chunk it: LDS buffer 0; registers A hold it + 1
    if (it + 2 < it1) load(it + 2, cB);
    mma(0);
    if (it + 1 < it1) stage(1, cA);                // (buffer 1 was read during chunk it - 1: every wave is past that barrier)
    __syncthreads();
    if (it + 1 >= it1) break;
    // chunk it + 1: LDS buffer 1; registers B hold it + 2
    if (it + 3 < it1) load(it + 3, cA);
    mma(1);
    if (it + 2 < it1) stage(0, cB);
    __syncthreads();
  }

  // epilogue: C/D layout col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5) -> transposed through LDS so that
  // the tile leaves as float4 rows
  constexpr int OS = IC_BN + 4;
  float* otile = smem;                                 // [BM][68], aliases the operand buffers (all reads are done)
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = wm * (BM / 2) + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
      otile[row * OS + wn * 32 + i] = acc[mt][r];
    }
  __syncthreads();
#pragma unroll
  for (int e = 0; e < (BM * 16) / 256; ++e) {
    const int f = t + 256 * e;
    const int row = f >> 4, q = f & 15;
    const int m = m0 + row, n = n0 + 4 * q;
    if (m >= a.M || n >= a.Cout) continue;
    float4 v = *reinterpret_cast<const float4*>(otile + row * OS + 4 * q);
    if (a.partial) {                                 // raw partial sum; conv2d_reduce finishes
      *reinterpret_cast<float4*>(a.ws + ((int64_t)(a.zbase + zsplit) * a.M + m) * a.Npad + n) = v;
      continue;
    }
    if (a.bias) {
      const float4 bb = *reinterpret_cast<const float4*>(a.bias + n);
      v.x += bb.x; v.y += bb.y; v.z += bb.z; v.w += bb.w;
    }
    if (a.ypre) *reinterpret_cast<float4*>(a.ypre + (int64_t)m * a.ldp + n) = v;
    if (a.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
    float4* dst = reinterpret_cast<float4*>(a.y + (int64_t)m * a.ldy + n);
    if (a.accumulate) {
      const float4 o = *dst;
      v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
    }
    *dst = v;
  }
}

template <int BM, int CMODE>
__global__ void __launch_bounds__(256) conv2d_mfma_kernel(Conv2dArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  conv2d_tile<BM, CMODE>(a, blockIdx.x, blockIdx.y, blockIdx.z, smem);
}

// Several convolutions in ONE launch (the branches of an inception module are a few hundred pixels each: alone a branch
// fills a tenth of the chip): block -> (problem, m tile, n tile, K split) through a table in the kernel argument.
// Problems are ordered longest block first by the host.
__global__ void __launch_bounds__(256) conv2d_group_kernel(Conv2dGroupArgs g) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int bid = blockIdx.x;
  int j = 0;
  while (j + 1 < g.n && bid >= g.bstart[j + 1]) ++j;
  const Conv2dArgs& a = g.p[j];
  const int local = bid - g.bstart[j];
  const int mtiles = (a.M + 63) / 64, ntiles = a.Npad / IC_BN;
  const int mt = local % mtiles, r = local / mtiles;
  conv2d_tile<64, 0>(a, mt, r % ntiles, r / ntiles, smem);
}

// second pass of the partial-sum path: y = epilogue(sum_z ws[z]) in a fixed order (deterministic)
__device__ __forceinline__ void conv2d_reduce(const Conv2dArgs& a, int64_t idx) {
  const int C4 = a.Cout >> 2;
  if (idx >= (int64_t)a.M * C4) return;
  const int m = (int)(idx / C4), n = (int)(idx - (int64_t)m * C4) * 4;
  const float* p = a.ws + (int64_t)m * a.Npad + n;
  const int64_t zs = (int64_t)a.M * a.Npad;
  float4 v = *reinterpret_cast<const float4*>(p);
  for (int z = 1; z < a.ztotal; ++z) {
    const float4 o = *reinterpret_cast<const float4*>(p + z * zs);
    v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
  }
  if (a.bias) {
    const float4 bb = *reinterpret_cast<const float4*>(a.bias + n);
    v.x += bb.x; v.y += bb.y; v.z += bb.z; v.w += bb.w;
  }
  if (a.ypre) *reinterpret_cast<float4*>(a.ypre + (int64_t)m * a.ldp + n) = v;
  if (a.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
  float4* dst = reinterpret_cast<float4*>(a.y + (int64_t)m * a.ldy + n);
  if (a.accumulate) {
    const float4 o = *dst;
    v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
  }
  *dst = v;
}

__global__ void __launch_bounds__(256) conv2d_reduce_kernel(Conv2dArgs a) {
  conv2d_reduce(a, (int64_t)blockIdx.x * 256 + threadIdx.x);
}

__global__ void __launch_bounds__(256) conv2d_group_reduce_kernel(Conv2dGroupArgs g) {
  const int bid = blockIdx.x;
  int j = 0;
  while (j + 1 < g.n && bid >= g.bstart[j + 1]) ++j;
  conv2d_reduce(g.p[j], (int64_t)(bid - g.bstart[j]) * 256 + threadIdx.x);
}

// packed[(chunk * Npad + n) * 16 + kk]; transpose = data-gradient filters (taps flipped, channels swapped)
__global__ void __launch_bounds__(256) conv2d_pack_kernel(const float* __restrict__ w, float* __restrict__ wp, int kh,
                                                          int kw, int Ci, int Co, int transpose, int Cin, int Cout,
                                                          int cchunks, int Npad, int64_t total) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int kk = (int)(idx & 15);
  const int64_t cn = idx >> 4;
  const int n = (int)(cn % Npad);
  const int chunk = (int)(cn / Npad);
  int tap, c;
  if (Cin <= 4) { tap = chunk * 4 + (kk >> 2); c = kk & 3; }
  else { tap = chunk / cchunks; c = (chunk - tap * cchunks) * 16 + kk; }
  float v = 0.f;
  if (tap < kh * kw && c < Cin && n < Cout) {
    const int dy = tap / kw, dx = tap - dy * kw;
    // source is HWIO [kh][kw][Ci][Co]
    v = transpose ? w[(((int64_t)(kh - 1 - dy) * kw + (kw - 1 - dx)) * Ci + n) * Co + c]
                  : w[(((int64_t)dy * kw + dx) * Ci + c) * Co + n];
  }
  wp[idx] = v;
}

// data gradient of a convolution down to <= 4 input channels: gx[p][c] = sum over the taps (ky, kx) whose output pixel
// exists, sum_co gy[o][co] (y_act[o][co] > 0) w[ky][kx][c][co].
// A wave holds pixels of ONE parity class (iy mod stride, ix mod stride): the taps that reach a pixel depend only on
// its class, so the tap loop and every filter address are uniform across the wave -- the filters arrive through the
// scalar cache as SGPR operands of the FMAs (no LDS, no per-lane filter traffic); per (tap, 4 channels) a lane issues
// one 16-byte load of gy and 4 CI FMAs.
template <int CI, bool MASK>
__global__ void __launch_bounds__(256) conv2d_small_dgrad_kernel(const float* __restrict__ gy, int ldg,
                                                                 const float* __restrict__ yact, int lda,
                                                                 const float* __restrict__ w, float* __restrict__ gx,
                                                                 int B, int H, int W, int Ho, int Wo, int Co,
                                                                 int kh, int kw, int stride, int pad_t, int pad_l) {
  const int cls = blockIdx.y;
  const int py = cls / stride, px = cls - py * stride;
  const int nqy = (H - py + stride - 1) / stride, nqx = (W - px + stride - 1) / stride;
  const int64_t nq = (int64_t)B * nqy * nqx;
  const int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const bool live = q < nq;
  const int64_t qq = live ? q : 0;
  const int per = nqy * nqx > 0 ? nqy * nqx : 1;
  const int b = (int)(qq / per);
  const int rem = (int)(qq - (int64_t)b * per);
  const int qy = rem / (nqx > 0 ? nqx : 1), qx = rem - qy * (nqx > 0 ? nqx : 1);
  const int ky0 = (py + pad_t) % stride, kx0 = (px + pad_l) % stride;
  const int cy = (py + pad_t - ky0) / stride, cx = (px + pad_l - kx0) / stride;
  float acc[CI];
#pragma unroll
  for (int c = 0; c < CI; ++c) acc[c] = 0.f;
  for (int ky = ky0, jy = 0; ky < kh; ky += stride, ++jy) {
    const int oy = qy + cy - jy;
    for (int kx = kx0, jx = 0; kx < kw; kx += stride, ++jx) {
      const int ox = qx + cx - jx;
      const bool ok = live && (unsigned)oy < (unsigned)Ho && (unsigned)ox < (unsigned)Wo;
      const int64_t o = ok ? ((int64_t)b * Ho + oy) * Wo + ox : 0;
      const float* gp = gy + o * ldg;
      const float* ap = MASK ? yact + o * lda : nullptr;
      const float* wt = w + (int64_t)(ky * kw + kx) * CI * Co;       // [CI][Co], uniform
#pragma unroll 4
      for (int co = 0; co < Co; co += 4) {
        float4 g = *reinterpret_cast<const float4*>(gp + co);
        if (MASK) {
          const float4 m = *reinterpret_cast<const float4*>(ap + co);
          g.x = m.x > 0.f ? g.x : 0.f; g.y = m.y > 0.f ? g.y : 0.f;
          g.z = m.z > 0.f ? g.z : 0.f; g.w = m.w > 0.f ? g.w : 0.f;
        }
        if (!ok) g = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int c = 0; c < CI; ++c) {
          const float* wc = wt + c * Co + co;
          acc[c] = fmaf(g.x, wc[0], fmaf(g.y, wc[1], fmaf(g.z, wc[2], fmaf(g.w, wc[3], acc[c]))));
        }
      }
    }
  }
  if (live) {
    const int iy = py + stride * qy, ix = px + stride * qx;
    float* o = gx + (((int64_t)b * H + iy) * W + ix) * CI;
#pragma unroll
    for (int c = 0; c < CI; ++c) o[c] = acc[c];
  }
}

// The stride-2 case (the graph's own first layer) with the output-gradient tile in LDS: a block owns 16 x 16 image
// pixels, i.e. 8 x 8 pixels of each parity class = one wave per class (filters stay wave-uniform SGPR operands), and the
// 12 x 12 outputs x Co channels those pixels draw from are staged once, masked, with zeros where no output exists -- the
// first form read every output 12 times through the vector memory path (256 16-byte loads per pixel), which is what
// bounded it.  Pixel stride Co + 4 floats and a row stride = 32 mod 64 keep the 16-lane groups of a b128 read (two
// rows of eight neighbours) on distinct banks.
constexpr int SD_T = 16, SD_TO = 12;   // (15 + 7 - 1) / 2 + 2 output rows / columns reach a 16-pixel tile
template <int CI, bool MASK>
__global__ void __launch_bounds__(256) conv2d_small_dgrad_s2_kernel(const float* __restrict__ gy, int ldg,
                                                                    const float* __restrict__ yact, int lda,
                                                                    const float* __restrict__ w, float* __restrict__ gx,
                                                                    int H, int W, int Ho, int Wo, int Co, int kh, int kw,
                                                                    int pad_t, int pad_l, int tiles_x, int tiles_y,
                                                                    int cs, int rs) {
  extern __shared__ __attribute__((aligned(16))) float gt[];    // [12][rs], pixel stride cs
  const int t = threadIdx.x;
  const int tx = blockIdx.x % tiles_x, ty = (blockIdx.x / tiles_x) % tiles_y, b = blockIdx.x / (tiles_x * tiles_y);
  const int iy0 = ty * SD_T, ix0 = tx * SD_T;
  const int oy_lo = (iy0 + pad_t - kh + 1) >> 1, ox_lo = (ix0 + pad_l - kw + 1) >> 1;      // (arithmetic shift = floor)
  const int q4 = Co >> 2;
  for (int idx = t; idx < SD_TO * SD_TO * q4; idx += 256) {
    const int pix = idx / q4, q = idx - pix * q4;
    const int ry = pix / SD_TO, rx = pix - ry * SD_TO;
    const int oy = oy_lo + ry, ox = ox_lo + rx;
    const bool ok = (unsigned)oy < (unsigned)Ho && (unsigned)ox < (unsigned)Wo;
    const int64_t o = ok ? ((int64_t)b * Ho + oy) * Wo + ox : 0;
    float4 g = *reinterpret_cast<const float4*>(gy + o * ldg + 4 * q);
    if (MASK) {
      const float4 m = *reinterpret_cast<const float4*>(yact + o * lda + 4 * q);
      g.x = m.x > 0.f ? g.x : 0.f; g.y = m.y > 0.f ? g.y : 0.f;
      g.z = m.z > 0.f ? g.z : 0.f; g.w = m.w > 0.f ? g.w : 0.f;
    }
    if (!ok) g = make_float4(0.f, 0.f, 0.f, 0.f);
    *reinterpret_cast<float4*>(gt + ry * rs + rx * cs + 4 * q) = g;
  }
  __syncthreads();
  const int cls = __builtin_amdgcn_readfirstlane(t >> 6), lane = t & 63;
  const int py = cls >> 1, px = cls & 1;
  const int iy = iy0 + py + 2 * (lane >> 3), ix = ix0 + px + 2 * (lane & 7);
  const int ky0 = (py + pad_t) & 1, kx0 = (px + pad_l) & 1;
  float acc[CI];
#pragma unroll
  for (int c = 0; c < CI; ++c) acc[c] = 0.f;
  for (int ky = ky0; ky < kh; ky += 2) {
    const int ry = ((iy + pad_t - ky) >> 1) - oy_lo;                      // in [0, 12) by construction
    for (int kx = kx0; kx < kw; kx += 2) {
      const int rx = ((ix + pad_l - kx) >> 1) - ox_lo;
      const float* gp = gt + ry * rs + rx * cs;
      const float* wt = w + (int64_t)(ky * kw + kx) * CI * Co;            // [CI][Co], wave-uniform
#pragma unroll 4
      for (int co = 0; co < Co; co += 4) {
        const float4 g = *reinterpret_cast<const float4*>(gp + co);
#pragma unroll
        for (int c = 0; c < CI; ++c) {
          const float* wc = wt + c * Co + co;
          acc[c] = fmaf(g.x, wc[0], fmaf(g.y, wc[1], fmaf(g.z, wc[2], fmaf(g.w, wc[3], acc[c]))));
        }
      }
    }
  }
  if (iy < H && ix < W) {
    float* o = gx + (((int64_t)b * H + iy) * W + ix) * CI;
#pragma unroll
    for (int c = 0; c < CI; ++c) o[c] = acc[c];
  }
}

// ---- 3x3 max pool, SAME (TF: out-of-range taps do not take part) ------------------------------------------------------
// thread = (output pixel, 4 channels); arg = window position (0..8, row-major) of the first maximum
__global__ void __launch_bounds__(256) maxpool3_fwd_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                           uint32_t* __restrict__ arg, int B, int H, int W, int Ho,
                                                           int Wo, int C4, int stride, int pad_t, int pad_l) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (int64_t)B * Ho * Wo * C4) return;
  const int c4 = (int)(idx % C4);
  const int64_t o = idx / C4;
  const int ox = (int)(o % Wo);
  const int oy = (int)((o / Wo) % Ho);
  const int b = (int)(o / ((int64_t)Wo * Ho));
  const float ninf = -__builtin_inff();
  float4 best = make_float4(ninf, ninf, ninf, ninf);
  uint32_t a0 = 0, a1 = 0, a2 = 0, a3 = 0;
#pragma unroll
  for (int tap = 0; tap < 9; ++tap) {
    const int iy = oy * stride - pad_t + tap / 3, ix = ox * stride - pad_l + tap % 3;
    if ((unsigned)iy >= (unsigned)H || (unsigned)ix >= (unsigned)W) continue;
    const float4 v = reinterpret_cast<const float4*>(x)[(((int64_t)b * H + iy) * W + ix) * C4 + c4];
    if (v.x > best.x) { best.x = v.x; a0 = tap; }
    if (v.y > best.y) { best.y = v.y; a1 = tap; }
    if (v.z > best.z) { best.z = v.z; a2 = tap; }
    if (v.w > best.w) { best.w = v.w; a3 = tap; }
  }
  reinterpret_cast<float4*>(y)[idx] = best;
  arg[idx] = a0 | (a1 << 8) | (a2 << 16) | (a3 << 24);
}

// thread = (input pixel, 4 channels): gather over the windows that contain the pixel
__global__ void __launch_bounds__(256) maxpool3_bwd_kernel(const float* __restrict__ gy, const uint32_t* __restrict__ arg,
                                                           float* __restrict__ gx, int B, int H, int W, int Ho, int Wo,
                                                           int C4, int stride, int pad_t, int pad_l, int accumulate,
                                                           const float* __restrict__ relu_of) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (int64_t)B * H * W * C4) return;
  const int c4 = (int)(idx % C4);
  const int64_t p = idx / C4;
  const int ix = (int)(p % W);
  const int iy = (int)((p / W) % H);
  const int b = (int)(p / ((int64_t)W * H));
  float4 acc = accumulate ? reinterpret_cast<const float4*>(gx)[idx] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int ty = 0; ty < 3; ++ty) {
    const int ny = iy + pad_t - ty;           // = oy * stride
    if (ny < 0 || ny % stride) continue;
    const int oy = ny / stride;
    if (oy >= Ho) continue;
#pragma unroll
    for (int tx = 0; tx < 3; ++tx) {
      const int nx = ix + pad_l - tx;
      if (nx < 0 || nx % stride) continue;
      const int ox = nx / stride;
      if (ox >= Wo) continue;
      const int64_t o = (((int64_t)b * Ho + oy) * Wo + ox) * C4 + c4;
      const uint32_t a = arg[o];
      const uint32_t tap = ty * 3 + tx;
      const float4 g = reinterpret_cast<const float4*>(gy)[o];
      if ((a & 255u) == tap) acc.x += g.x;
      if (((a >> 8) & 255u) == tap) acc.y += g.y;
      if (((a >> 16) & 255u) == tap) acc.z += g.z;
      if ((a >> 24) == tap) acc.w += g.w;
    }
  }
  if (relu_of) {                                   // the pooled tensor is a ReLU output and this is the last term of
    const float4 m = reinterpret_cast<const float4*>(relu_of)[idx];   // its gradient: hand it on with the ReLU adjoint
    acc.x = m.x > 0.f ? acc.x : 0.f; acc.y = m.y > 0.f ? acc.y : 0.f;
    acc.z = m.z > 0.f ? acc.z : 0.f; acc.w = m.w > 0.f ? acc.w : 0.f;
  }
  reinterpret_cast<float4*>(gx)[idx] = acc;
}

// ---- tf.nn.lrn: y = x / (bias + alpha * sum_{|j - c| <= r} x_j^2)^beta ----------------------------------------------------
__global__ void __launch_bounds__(256) lrn_fwd_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                      float* __restrict__ scale, int64_t npix, int C, int ld, int radius,
                                                      float bias, float alpha, float beta) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= npix * C) return;
  const int c = (int)(idx % C);
  const int64_t p = idx / C;
  const float* xp = x + p * ld;
  float s = 0.f;
  const int lo = c - radius < 0 ? 0 : c - radius, hi = c + radius >= C ? C - 1 : c + radius;
  for (int j = lo; j <= hi; ++j) s += xp[j] * xp[j];
  s = bias + alpha * s;
  scale[p * ld + c] = s;
  y[p * ld + c] = xp[c] * powf(s, -beta);
}

// gx_c = gy_c s_c^-beta - 2 alpha beta x_c sum_{|j - c| <= r} gy_j y_j / s_j
__global__ void __launch_bounds__(256) lrn_bwd_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                      const float* __restrict__ scale, const float* __restrict__ gy,
                                                      float* __restrict__ gx, int64_t npix, int C, int ld, int radius,
                                                      float alpha, float beta, int accumulate) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= npix * C) return;
  const int c = (int)(idx % C);
  const int64_t p = idx / C;
  const int64_t o = p * ld;
  float s = 0.f;
  const int lo = c - radius < 0 ? 0 : c - radius, hi = c + radius >= C ? C - 1 : c + radius;
  for (int j = lo; j <= hi; ++j) s += gy[o + j] * y[o + j] / scale[o + j];
  float v = gy[o + c] * powf(scale[o + c], -beta) - 2.f * alpha * beta * x[o + c] * s;
  if (accumulate) v += gx[o + c];
  gx[o + c] = v;
}

// ---- AvgPool k x k, stride 1, VALID (the graph's avgpool0 in front of the classifier) -----------------------------------
__global__ void __launch_bounds__(256) avgpool_valid_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int B,
                                                                int H, int W, int C4, int k) {
  const int Ho = H - k + 1, Wo = W - k + 1;
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (int64_t)B * Ho * Wo * C4) return;
  const int c4 = (int)(idx % C4);
  const int64_t o = idx / C4;
  const int ox = (int)(o % Wo), oy = (int)((o / Wo) % Ho), b = (int)(o / ((int64_t)Wo * Ho));
  float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int dy = 0; dy < k; ++dy)
    for (int dx = 0; dx < k; ++dx) {
      const float4 v = reinterpret_cast<const float4*>(x)[(((int64_t)b * H + oy + dy) * W + ox + dx) * C4 + c4];
      a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
    }
  const float inv = 1.f / (float)(k * k);
  reinterpret_cast<float4*>(y)[idx] = make_float4(a.x * inv, a.y * inv, a.z * inv, a.w * inv);
}

__global__ void __launch_bounds__(256) avgpool_valid_bwd_kernel(const float* __restrict__ gy, float* __restrict__ gx,
                                                                int B, int H, int W, int C4, int k, int accumulate) {
  const int Ho = H - k + 1, Wo = W - k + 1;
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (int64_t)B * H * W * C4) return;
  const int c4 = (int)(idx % C4);
  const int64_t p = idx / C4;
  const int ix = (int)(p % W), iy = (int)((p / W) % H), b = (int)(p / ((int64_t)W * H));
  float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
  const int oy0 = iy - k + 1 < 0 ? 0 : iy - k + 1, oy1 = iy < Ho - 1 ? iy : Ho - 1;
  const int ox0 = ix - k + 1 < 0 ? 0 : ix - k + 1, ox1 = ix < Wo - 1 ? ix : Wo - 1;
  for (int oy = oy0; oy <= oy1; ++oy)
    for (int ox = ox0; ox <= ox1; ++ox) {
      const float4 v = reinterpret_cast<const float4*>(gy)[(((int64_t)b * Ho + oy) * Wo + ox) * C4 + c4];
      a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
    }
  const float inv = 1.f / (float)(k * k);
  float4 r = make_float4(a.x * inv, a.y * inv, a.z * inv, a.w * inv);
  if (accumulate) {
    const float4 o = reinterpret_cast<const float4*>(gx)[idx];
    r.x += o.x; r.y += o.y; r.z += o.z; r.w += o.w;
  }
  reinterpret_cast<float4*>(gx)[idx] = r;
}

// out = g (act > 0) + addend   (gradient injected at a *_pre_relu tensor: added AFTER the ReLU adjoint)
__global__ void __launch_bounds__(256) relu_mask_add_kernel(const float* __restrict__ g, int ldg,
                                                            const float* __restrict__ act, int lda,
                                                            const float* __restrict__ addend, int ldadd,
                                                            float* __restrict__ out, int ldo, int64_t npix, int C4) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= npix * C4) return;
  const int c = (int)(idx % C4) * 4;
  const int64_t p = idx / C4;
  float4 v = g ? *reinterpret_cast<const float4*>(g + p * ldg + c) : make_float4(0.f, 0.f, 0.f, 0.f);
  if (act) {
    const float4 m = *reinterpret_cast<const float4*>(act + p * lda + c);
    v.x = m.x > 0.f ? v.x : 0.f; v.y = m.y > 0.f ? v.y : 0.f; v.z = m.z > 0.f ? v.z : 0.f; v.w = m.w > 0.f ? v.w : 0.f;
  }
  if (addend) {
    const float4 ad = *reinterpret_cast<const float4*>(addend + p * ldadd + c);
    v.x += ad.x; v.y += ad.y; v.z += ad.z; v.w += ad.w;
  }
  *reinterpret_cast<float4*>(out + p * ldo + c) = v;
}

static inline void same_pad(int in, int k, int stride, int& out, int& pad_before) {
  out = (in + stride - 1) / stride;
  int total = (out - 1) * stride + k - in;
  if (total < 0) total = 0;
  pad_before = total / 2;
}

static inline int conv2d_cchunks(int Cin) { return (Cin + 15) / 16; }
static inline int conv2d_nchunks(int kh, int kw, int Cin) {
  return Cin <= 4 ? (kh * kw + 3) / 4 : kh * kw * conv2d_cchunks(Cin);
}
static inline int conv2d_npad(int Cout) { return (Cout + IC_BN - 1) / IC_BN * IC_BN; }

// tile height and K split of one call: 128-row tiles once they still give every CU two blocks, 64-row tiles below;
// when even those leave CUs idle and K is long, split K until ~512 blocks exist (>= 8 chunks per split, <= 16 splits)
struct Conv2dPlan { int bm, ksplit, cps; };
static Conv2dPlan conv2d_plan(int64_t M, int Npad, int nchunks) {
  Conv2dPlan p;
  const int ntiles = Npad / IC_BN;
  p.bm = (M / 128) * ntiles >= 512 ? 128 : 64;
  const int64_t blocks = ((M + p.bm - 1) / p.bm) * ntiles;
  int ks = 1;
  if (blocks < 384 && nchunks >= 16) {
    ks = (int)((512 + blocks - 1) / blocks);
    if (ks > nchunks / 8) ks = nchunks / 8;
    if (ks > 16) ks = 16;
    if (ks < 1) ks = 1;
  }
  p.cps = (nchunks + ks - 1) / ks;
  p.ksplit = (nchunks + p.cps - 1) / p.cps;
  return p;
}

}  // namespace nfs

using namespace nfs;

extern "C" {

int64_t nfs_conv2d_packed_floats(int kh, int kw, int Ci, int Co, int transpose) {
  if (kh <= 0 || kw <= 0 || Ci <= 0 || Co <= 0) return 0;
  const int Cin = transpose ? Co : Ci, Cout = transpose ? Ci : Co;
  return (int64_t)conv2d_nchunks(kh, kw, Cin) * conv2d_npad(Cout) * 16;
}

int nfs_conv2d_pack(const float* w_hwio, float* packed, int kh, int kw, int Ci, int Co, int transpose,
                    nfs_stream_t stream) {
  NFS_REQUIRE(w_hwio && packed, "nfs_conv2d_pack: null pointer");
  NFS_REQUIRE(kh > 0 && kw > 0 && Ci > 0 && Co > 0, "nfs_conv2d_pack: non-positive dimension");
  const int Cin = transpose ? Co : Ci, Cout = transpose ? Ci : Co;
  const int64_t total = nfs_conv2d_packed_floats(kh, kw, Ci, Co, transpose);
  conv2d_pack_kernel<<<blocks_for(total, 256), 256, 0, as_stream(stream)>>>(w_hwio, packed, kh, kw, Ci, Co, transpose, Cin,
                                                                          Cout, conv2d_cchunks(Cin), conv2d_npad(Cout),
                                                                          total);
  return check_launch("nfs_conv2d_pack");
}

int64_t nfs_conv2d_workspace_floats(int B, int H, int W, int Cin, int Cout, int kh, int kw, int stride) {
  if (B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || kh <= 0 || kw <= 0 || (stride != 1 && stride != 2)) return 0;
  int Ho, Wo, pt, pl;
  same_pad(H, kh, stride, Ho, pt);
  same_pad(W, kw, stride, Wo, pl);
  const int64_t M = (int64_t)B * Ho * Wo;
  const Conv2dPlan p = conv2d_plan(M, conv2d_npad(Cout), conv2d_nchunks(kh, kw, Cin));
  return p.ksplit > 1 ? (int64_t)p.ksplit * M * conv2d_npad(Cout) : 0;
}

int nfs_conv2d_fwd(const float* x, int ldx, const float* x_mask, int ldm, const float* packed, const float* bias,
                   float* y, int ldy, float* y_pre, int ldp, int B, int H, int W, int Cin, int Cout, int kh, int kw,
                   int stride, int relu, int accumulate, float* workspace, int64_t workspace_floats,
                   nfs_stream_t stream) {
  NFS_REQUIRE(x && packed && y, "nfs_conv2d_fwd: null pointer");
  NFS_REQUIRE(B > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0, "nfs_conv2d_fwd: non-positive dimension");
  NFS_REQUIRE(kh > 0 && kw > 0 && kh <= 7 && kw <= 7 && (stride == 1 || stride == 2),
              "nfs_conv2d_fwd: filter up to 7x7, stride 1 or 2");
  NFS_REQUIRE(Cout % 4 == 0 && ldy % 4 == 0 && ldy >= Cout, "nfs_conv2d_fwd: Cout and ldy must be multiples of 4, ldy >= Cout");
  NFS_REQUIRE(Cin <= 4 || (Cin % 4 == 0 && ldx % 4 == 0), "nfs_conv2d_fwd: Cin <= 4, or Cin and ldx multiples of 4");
  NFS_REQUIRE(ldx >= Cin, "nfs_conv2d_fwd: ldx < Cin");
  NFS_REQUIRE(!(x_mask && Cin <= 4), "nfs_conv2d_fwd: x_mask needs Cin > 4");
  NFS_REQUIRE(!x_mask || (ldm % 4 == 0 && ldm >= Cin), "nfs_conv2d_fwd: bad ldm");
  NFS_REQUIRE(!y_pre || (ldp % 4 == 0 && ldp >= Cout), "nfs_conv2d_fwd: bad ldp");
  NFS_REQUIRE((((uintptr_t)y | (uintptr_t)y_pre | (uintptr_t)packed | (uintptr_t)bias) & 15) == 0 &&
                  (Cin <= 4 || (((uintptr_t)x | (uintptr_t)x_mask) & 15) == 0),
              "nfs_conv2d_fwd: pointers must be 16-byte aligned");
  Conv2dArgs a;
  a.x = x; a.xmask = x_mask; a.wp = packed; a.bias = bias; a.y = y; a.ypre = y_pre;
  a.ldx = ldx; a.ldm = ldm; a.ldy = ldy; a.ldp = ldp;
  a.B = B; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout; a.kh = kh; a.kw = kw; a.stride = stride;
  same_pad(H, kh, stride, a.Ho, a.pad_t);
  same_pad(W, kw, stride, a.Wo, a.pad_l);
  a.relu = relu; a.accumulate = accumulate;
  const int64_t M = (int64_t)B * a.Ho * a.Wo;
  NFS_REQUIRE((int64_t)B * H * W * (int64_t)(ldx > ldm ? ldx : ldm) < ((int64_t)1 << 40) && M < ((int64_t)1 << 31),
              "nfs_conv2d_fwd: too many pixels");
  NFS_REQUIRE((int64_t)B * H * W < ((int64_t)1 << 31), "nfs_conv2d_fwd: too many pixels");
  a.M = (int)M;
  {
    const int64_t xb = (((int64_t)B * H * W - 1) * ldx + Cin) * 4, mb = x_mask ? (((int64_t)B * H * W - 1) * ldm + Cin) * 4 : 0;
    NFS_REQUIRE(xb < ((int64_t)1 << 31) && mb < ((int64_t)1 << 31), "nfs_conv2d_fwd: operands must stay below 2 GB");
    a.x_bytes = (uint32_t)xb; a.m_bytes = (uint32_t)mb;
  }
  a.cchunks = conv2d_cchunks(Cin);
  a.nchunks = conv2d_nchunks(kh, kw, Cin);
  a.Npad = conv2d_npad(Cout);
  const int ntiles = a.Npad / IC_BN;
  Conv2dPlan p = conv2d_plan(M, a.Npad, a.nchunks);
  if (p.ksplit > 1 && (!workspace || workspace_floats < (int64_t)p.ksplit * M * a.Npad)) {
    p.ksplit = 1;                                   // no (or too small a) scratch buffer: one block per tile
    p.cps = a.nchunks;
  }
  NFS_REQUIRE(p.ksplit == 1 || ((uintptr_t)workspace & 15) == 0, "nfs_conv2d_fwd: workspace must be 16-byte aligned");
  a.ws = workspace; a.ksplit = p.ksplit; a.cps = p.cps;
  a.partial = p.ksplit > 1; a.zbase = 0; a.ztotal = p.ksplit;
  hipStream_t s = as_stream(stream);
  if (p.bm == 128) {
    const size_t lds = sizeof(float) * (size_t)(128 * (IC_BN + 4));          // >= 2 * (128 + 64) * 20
    dim3 grid((unsigned)((M + 127) / 128), ntiles, p.ksplit);
    if (Cin <= 4) conv2d_mfma_kernel<128, 1><<<grid, 256, lds, s>>>(a);
    else conv2d_mfma_kernel<128, 0><<<grid, 256, lds, s>>>(a);
  } else {
    const size_t lds = sizeof(float) * (size_t)(2 * (64 + IC_BN) * IC_LS);   // >= 64 * 68
    dim3 grid((unsigned)((M + 63) / 64), ntiles, p.ksplit);
    if (Cin <= 4) conv2d_mfma_kernel<64, 1><<<grid, 256, lds, s>>>(a);
    else conv2d_mfma_kernel<64, 0><<<grid, 256, lds, s>>>(a);
  }
  if (p.ksplit > 1)
    conv2d_reduce_kernel<<<blocks_for(M * (Cout / 4), 256), 256, 0, s>>>(a);
  return check_launch("nfs_conv2d_fwd");
}

// ---- grouped launch -------------------------------------------------------------------------------------------------
namespace {
struct GroupPlan {
  Conv2dGroupArgs conv, red;
  int64_t ws_floats;
};

// fills plan (pointers into `workspace` when given); returns 0, or a message
const char* conv2d_group_plan(const nfs_conv2d_desc_t* d, int n, int B, float* workspace, GroupPlan& P) {
  if (!d || n < 1 || n > IC_GMAX) return "1..6 problems";
  if (B <= 0) return "non-positive batch";
  Conv2dArgs a[IC_GMAX];
  int mtiles[IC_GMAX], ntiles[IC_GMAX], ks[IC_GMAX];
  for (int j = 0; j < n; ++j) {
    const nfs_conv2d_desc_t& q = d[j];
    if (!q.x || !q.packed || !q.y) return "null pointer";
    if (q.H <= 0 || q.W <= 0 || q.Cin <= 4 || q.Cout <= 0 || q.kh <= 0 || q.kw <= 0 || q.kh > 7 || q.kw > 7)
      return "bad dimension (Cin > 4, filters up to 7x7)";
    if (q.Cin % 4 || q.Cout % 4 || q.ldx % 4 || q.ldy % 4 || q.ldx < q.Cin || q.ldy < q.Cout) return "channel counts and row strides must be multiples of 4";
    if (q.x_mask && (q.ldm % 4 || q.ldm < q.Cin)) return "bad ldm";
    if (q.y_pre && (q.ldp % 4 || q.ldp < q.Cout)) return "bad ldp";
    if ((((uintptr_t)q.x | (uintptr_t)q.x_mask | (uintptr_t)q.packed | (uintptr_t)q.bias | (uintptr_t)q.y | (uintptr_t)q.y_pre) & 15) != 0)
      return "pointers must be 16-byte aligned";
    if ((int64_t)B * q.H * q.W >= ((int64_t)1 << 31)) return "too many pixels";
    Conv2dArgs& c = a[j];
    c.x = q.x; c.xmask = q.x_mask; c.wp = q.packed; c.bias = q.bias; c.y = q.y; c.ypre = q.y_pre;
    c.ldx = q.ldx; c.ldm = q.ldm; c.ldy = q.ldy; c.ldp = q.ldp;
    c.B = B; c.H = q.H; c.W = q.W; c.Cin = q.Cin; c.Cout = q.Cout; c.kh = q.kh; c.kw = q.kw; c.stride = 1;
    same_pad(q.H, q.kh, 1, c.Ho, c.pad_t);
    same_pad(q.W, q.kw, 1, c.Wo, c.pad_l);
    c.relu = q.relu; c.accumulate = q.accumulate;
    c.M = B * c.Ho * c.Wo;
    {
      const int64_t xb = (((int64_t)B * q.H * q.W - 1) * q.ldx + q.Cin) * 4;
      const int64_t mb = q.x_mask ? (((int64_t)B * q.H * q.W - 1) * q.ldm + q.Cin) * 4 : 0;
      if (xb >= ((int64_t)1 << 31) || mb >= ((int64_t)1 << 31)) return "operands must stay below 2 GB";
      c.x_bytes = (uint32_t)xb; c.m_bytes = (uint32_t)mb;
    }
    c.cchunks = conv2d_cchunks(q.Cin);
    c.nchunks = conv2d_nchunks(q.kh, q.kw, q.Cin);
    c.Npad = conv2d_npad(q.Cout);
    mtiles[j] = (c.M + 63) / 64;
    ntiles[j] = c.Npad / IC_BN;
    if (q.sum_with_prev) {
      if (j == 0) return "the first problem cannot be summed with a previous one";
      const Conv2dArgs& f = a[j - 1];
      if (f.M != c.M || f.Cout != c.Cout || f.y != c.y || f.ldy != c.ldy) return "summed problems must share pixels, Cout and y";
    }
  }
  // K chunks per block: the largest T of the ladder that yields >= 512 blocks (>= 8 chunks per block, <= 16 splits)
  static const int ladder[] = {1 << 30, 256, 128, 64, 32, 16, 8};
  for (int li = 0; li < 7; ++li) {
    int64_t blocks = 0;
    for (int j = 0; j < n; ++j) {
      int k = (a[j].nchunks + ladder[li] - 1) / ladder[li];
      if (k > 16) k = 16;
      if (k < 1) k = 1;
      ks[j] = k;
      blocks += (int64_t)mtiles[j] * ntiles[j] * k;
    }
    if (blocks >= 512) break;
  }
  // partial-sum regions: a run of summed problems, or a single split problem
  int64_t off = 0;
  int nred = 0;
  for (int j = 0; j < n;) {
    int e = j + 1;
    while (e < n && d[e].sum_with_prev) ++e;
    const bool joint = e - j > 1;
    int ztot = 0;
    for (int i = j; i < e; ++i) {
      a[i].cps = (a[i].nchunks + ks[i] - 1) / ks[i];
      a[i].ksplit = (a[i].nchunks + a[i].cps - 1) / a[i].cps;
      a[i].partial = joint || a[i].ksplit > 1;
      a[i].zbase = ztot;
      ztot += a[i].ksplit;
    }
    for (int i = j; i < e; ++i) {
      a[i].ztotal = ztot;
      a[i].ws = a[i].partial && workspace ? workspace + off : nullptr;
    }
    if (a[j].partial) {
      P.red.p[nred++] = a[j];
      off += (int64_t)ztot * a[j].M * a[j].Npad;
    }
    j = e;
  }
  P.ws_floats = off;
  // reduce table
  P.red.n = nred;
  int b0 = 0;
  for (int i = 0; i < nred; ++i) {
    P.red.bstart[i] = b0;
    b0 += (int)blocks_for((int64_t)P.red.p[i].M * (P.red.p[i].Cout / 4), 256);
  }
  P.red.bstart[nred] = b0;
  // conv table: longest blocks first
  int order[IC_GMAX];
  for (int j = 0; j < n; ++j) order[j] = j;
  for (int i = 1; i < n; ++i)
    for (int k = i; k > 0 && a[order[k]].cps > a[order[k - 1]].cps; --k) { const int t = order[k]; order[k] = order[k - 1]; order[k - 1] = t; }
  b0 = 0;
  for (int i = 0; i < n; ++i) {
    const int j = order[i];
    P.conv.p[i] = a[j];
    P.conv.bstart[i] = b0;
    b0 += mtiles[j] * ntiles[j] * a[j].ksplit;
  }
  P.conv.bstart[n] = b0;
  P.conv.n = n;
  return nullptr;
}
}  // namespace

int64_t nfs_conv2d_group_workspace_floats(const nfs_conv2d_desc_t* descs, int n, int B) {
  GroupPlan P;
  return conv2d_group_plan(descs, n, B, nullptr, P) ? -1 : P.ws_floats;
}

int nfs_conv2d_group(const nfs_conv2d_desc_t* descs, int n, int B, float* workspace, int64_t workspace_floats,
                     nfs_stream_t stream) {
  GroupPlan P;
  const char* err = conv2d_group_plan(descs, n, B, workspace, P);
  NFS_REQUIRE(!err, "nfs_conv2d_group: %s", err);
  NFS_REQUIRE(P.ws_floats == 0 || (workspace && workspace_floats >= P.ws_floats && ((uintptr_t)workspace & 15) == 0),
              "nfs_conv2d_group: workspace of %lld floats needed (16-byte aligned)", (long long)P.ws_floats);
  hipStream_t s = as_stream(stream);
  const size_t lds = sizeof(float) * (size_t)(2 * (64 + IC_BN) * IC_LS);
  conv2d_group_kernel<<<P.conv.bstart[P.conv.n], 256, lds, s>>>(P.conv);
  if (P.red.n > 0) conv2d_group_reduce_kernel<<<P.red.bstart[P.red.n], 256, 0, s>>>(P.red);
  return check_launch("nfs_conv2d_group");
}

int nfs_conv2d_dgrad_small(const float* gy, int ldg, const float* y_act, int lda, const float* w_hwio, float* gx, int B,
                           int H, int W, int Ci, int Co, int kh, int kw, int stride, nfs_stream_t stream) {
  NFS_REQUIRE(gy && w_hwio && gx, "nfs_conv2d_dgrad_small: null pointer");
  NFS_REQUIRE(B > 0 && H > 0 && W > 0 && Ci > 0 && Ci <= 4 && Co > 0 && Co % 4 == 0,
              "nfs_conv2d_dgrad_small: 1..4 input channels, Co a multiple of 4");
  NFS_REQUIRE(kh > 0 && kw > 0 && kh <= 7 && kw <= 7 && (stride == 1 || stride == 2),
              "nfs_conv2d_dgrad_small: filter up to 7x7, stride 1 or 2");
  NFS_REQUIRE(ldg % 4 == 0 && ldg >= Co && (!y_act || (lda % 4 == 0 && lda >= Co)), "nfs_conv2d_dgrad_small: bad row stride");
  NFS_REQUIRE((((uintptr_t)gy | (uintptr_t)y_act | (uintptr_t)w_hwio) & 15) == 0,
              "nfs_conv2d_dgrad_small: pointers must be 16-byte aligned");
  int Ho, Wo, pt, pl;
  same_pad(H, kh, stride, Ho, pt);
  same_pad(W, kw, stride, Wo, pl);
  hipStream_t s = as_stream(stream);
  if (stride == 2 && Co <= 64 && (int64_t)B * ((H + 15) / 16) * ((W + 15) / 16) < ((int64_t)1 << 31)) {
    // 16 x 16-pixel tiles with the 12 x 12 outputs they draw from in LDS
    const int tiles_x = (W + SD_T - 1) / SD_T, tiles_y = (H + SD_T - 1) / SD_T;
    const int cs = Co + 4;
    int rs = SD_TO * cs;
    rs += ((32 - rs % 64) + 64) % 64;                                             // row stride = 32 mod 64 floats
    const size_t lds = sizeof(float) * (size_t)SD_TO * rs;
#define NFS_LAUNCH_S2(CI_)                                                                                              \
  do {                                                                                                                    \
    if (y_act)                                                                                                            \
      conv2d_small_dgrad_s2_kernel<CI_, true><<<B * tiles_x * tiles_y, 256, lds, s>>>(                                    \
          gy, ldg, y_act, lda, w_hwio, gx, H, W, Ho, Wo, Co, kh, kw, pt, pl, tiles_x, tiles_y, cs, rs);                   \
    else                                                                                                                  \
      conv2d_small_dgrad_s2_kernel<CI_, false><<<B * tiles_x * tiles_y, 256, lds, s>>>(                                   \
          gy, ldg, y_act, lda, w_hwio, gx, H, W, Ho, Wo, Co, kh, kw, pt, pl, tiles_x, tiles_y, cs, rs);                   \
  } while (0)
    if (Ci == 1) NFS_LAUNCH_S2(1);
    else if (Ci == 2) NFS_LAUNCH_S2(2);
    else if (Ci == 3) NFS_LAUNCH_S2(3);
    else NFS_LAUNCH_S2(4);
#undef NFS_LAUNCH_S2
    return check_launch("nfs_conv2d_dgrad_small");
  }
  const int nqy = (H + stride - 1) / stride, nqx = (W + stride - 1) / stride;      // class (0, 0) is the largest
  dim3 grid(blocks_for((int64_t)B * nqy * nqx, 256), stride * stride);
#define NFS_LAUNCH(CI_)                                                                                               \
  do {                                                                                                                \
    if (y_act)                                                                                                        \
      conv2d_small_dgrad_kernel<CI_, true><<<grid, 256, 0, s>>>(gy, ldg, y_act, lda, w_hwio, gx, B, H, W, Ho, Wo, Co, \
                                                                kh, kw, stride, pt, pl);                              \
    else                                                                                                              \
      conv2d_small_dgrad_kernel<CI_, false><<<grid, 256, 0, s>>>(gy, ldg, y_act, lda, w_hwio, gx, B, H, W, Ho, Wo,    \
                                                                 Co, kh, kw, stride, pt, pl);                         \
  } while (0)
  if (Ci == 1) NFS_LAUNCH(1);
  else if (Ci == 2) NFS_LAUNCH(2);
  else if (Ci == 3) NFS_LAUNCH(3);
  else NFS_LAUNCH(4);
#undef NFS_LAUNCH
  return check_launch("nfs_conv2d_dgrad_small");
}

int nfs_maxpool3_fwd(const float* x, float* y, uint8_t* arg, int B, int H, int W, int C, int stride, nfs_stream_t stream) {
  NFS_REQUIRE(x && y && arg, "nfs_maxpool3_fwd: null pointer");
  NFS_REQUIRE(B > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0, "nfs_maxpool3_fwd: C must be a positive multiple of 4");
  NFS_REQUIRE(stride == 1 || stride == 2, "nfs_maxpool3_fwd: stride 1 or 2");
  NFS_REQUIRE((((uintptr_t)x | (uintptr_t)y) & 15) == 0 && ((uintptr_t)arg & 3) == 0, "nfs_maxpool3_fwd: misaligned pointer");
  int Ho, Wo, pt, pl;
  same_pad(H, 3, stride, Ho, pt);
  same_pad(W, 3, stride, Wo, pl);
  maxpool3_fwd_kernel<<<blocks_for((int64_t)B * Ho * Wo * (C / 4), 256), 256, 0, as_stream(stream)>>>(
      x, y, reinterpret_cast<uint32_t*>(arg), B, H, W, Ho, Wo, C / 4, stride, pt, pl);
  return check_launch("nfs_maxpool3_fwd");
}

int nfs_maxpool3_bwd(const float* gy, const uint8_t* arg, float* gx, int B, int H, int W, int C, int stride,
                     int accumulate, const float* relu_of, nfs_stream_t stream) {
  NFS_REQUIRE(gy && gx && arg, "nfs_maxpool3_bwd: null pointer");
  NFS_REQUIRE(B > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0, "nfs_maxpool3_bwd: C must be a positive multiple of 4");
  NFS_REQUIRE(stride == 1 || stride == 2, "nfs_maxpool3_bwd: stride 1 or 2");
  NFS_REQUIRE((((uintptr_t)gx | (uintptr_t)gy | (uintptr_t)relu_of) & 15) == 0 && ((uintptr_t)arg & 3) == 0,
              "nfs_maxpool3_bwd: misaligned pointer");
  int Ho, Wo, pt, pl;
  same_pad(H, 3, stride, Ho, pt);
  same_pad(W, 3, stride, Wo, pl);
  maxpool3_bwd_kernel<<<blocks_for((int64_t)B * H * W * (C / 4), 256), 256, 0, as_stream(stream)>>>(
      gy, reinterpret_cast<const uint32_t*>(arg), gx, B, H, W, Ho, Wo, C / 4, stride, pt, pl, accumulate, relu_of);
  return check_launch("nfs_maxpool3_bwd");
}

int nfs_lrn_fwd(const float* x, float* y, float* scale, int64_t npix, int C, int ld, int radius, float bias, float alpha,
                float beta, nfs_stream_t stream) {
  NFS_REQUIRE(x && y && scale, "nfs_lrn_fwd: null pointer");
  NFS_REQUIRE(npix > 0 && C > 0 && ld >= C && radius >= 0, "nfs_lrn_fwd: bad dimension");
  lrn_fwd_kernel<<<blocks_for(npix * C, 256), 256, 0, as_stream(stream)>>>(x, y, scale, npix, C, ld, radius, bias, alpha, beta);
  return check_launch("nfs_lrn_fwd");
}

int nfs_lrn_bwd(const float* x, const float* y, const float* scale, const float* gy, float* gx, int64_t npix, int C, int ld,
                int radius, float alpha, float beta, int accumulate, nfs_stream_t stream) {
  NFS_REQUIRE(x && y && scale && gy && gx, "nfs_lrn_bwd: null pointer");
  NFS_REQUIRE(npix > 0 && C > 0 && ld >= C && radius >= 0, "nfs_lrn_bwd: bad dimension");
  lrn_bwd_kernel<<<blocks_for(npix * C, 256), 256, 0, as_stream(stream)>>>(x, y, scale, gy, gx, npix, C, ld, radius, alpha,
                                                                         beta, accumulate);
  return check_launch("nfs_lrn_bwd");
}

int nfs_avgpool_valid_fwd(const float* x, float* y, int B, int H, int W, int C, int k, nfs_stream_t stream) {
  NFS_REQUIRE(x && y, "nfs_avgpool_valid_fwd: null pointer");
  NFS_REQUIRE(B > 0 && k > 0 && H >= k && W >= k && C > 0 && C % 4 == 0,
              "nfs_avgpool_valid_fwd: the image must hold one k x k window; C a multiple of 4");
  NFS_REQUIRE((((uintptr_t)x | (uintptr_t)y) & 15) == 0, "nfs_avgpool_valid_fwd: misaligned pointer");
  avgpool_valid_fwd_kernel<<<blocks_for((int64_t)B * (H - k + 1) * (W - k + 1) * (C / 4), 256), 256, 0,
                             as_stream(stream)>>>(x, y, B, H, W, C / 4, k);
  return check_launch("nfs_avgpool_valid_fwd");
}

int nfs_avgpool_valid_bwd(const float* gy, float* gx, int B, int H, int W, int C, int k, int accumulate,
                          nfs_stream_t stream) {
  NFS_REQUIRE(gy && gx, "nfs_avgpool_valid_bwd: null pointer");
  NFS_REQUIRE(B > 0 && k > 0 && H >= k && W >= k && C > 0 && C % 4 == 0,
              "nfs_avgpool_valid_bwd: the image must hold one k x k window; C a multiple of 4");
  NFS_REQUIRE((((uintptr_t)gx | (uintptr_t)gy) & 15) == 0, "nfs_avgpool_valid_bwd: misaligned pointer");
  avgpool_valid_bwd_kernel<<<blocks_for((int64_t)B * H * W * (C / 4), 256), 256, 0, as_stream(stream)>>>(
      gy, gx, B, H, W, C / 4, k, accumulate);
  return check_launch("nfs_avgpool_valid_bwd");
}

int nfs_relu_mask_add(const float* g, int ldg, const float* act, int lda, const float* addend, int ldadd, float* out,
                      int ldo, int64_t npix, int C, nfs_stream_t stream) {
  NFS_REQUIRE(out && (g || addend), "nfs_relu_mask_add: null pointer");
  NFS_REQUIRE(npix > 0 && C > 0 && C % 4 == 0 && ldo % 4 == 0 && ldo >= C, "nfs_relu_mask_add: C and row strides multiples of 4");
  NFS_REQUIRE((!g || (ldg % 4 == 0 && ldg >= C)) && (!act || (lda % 4 == 0 && lda >= C)) &&
                  (!addend || (ldadd % 4 == 0 && ldadd >= C)), "nfs_relu_mask_add: bad row stride");
  NFS_REQUIRE((((uintptr_t)g | (uintptr_t)act | (uintptr_t)addend | (uintptr_t)out) & 15) == 0,
              "nfs_relu_mask_add: pointers must be 16-byte aligned");
  relu_mask_add_kernel<<<blocks_for(npix * (C / 4), 256), 256, 0, as_stream(stream)>>>(g, ldg, act, lda, addend, ldadd, out,
                                                                                     ldo, npix, C / 4);
  return check_launch("nfs_relu_mask_add");
}

}  // extern "C"
