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
};

template <int BM, int CMODE>
__global__ void __launch_bounds__(256) conv2d_mfma_kernel(Conv2dArgs a) {
  constexpr int MT = BM / 64;   // 32x32 MFMA tiles per wave along M (= A float4 per thread and chunk)
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As = smem;                        // [2][BM][20]
  float* Bs = smem + 2 * BM * IC_LS;       // [2][64][20]
  const int t = threadIdx.x, lane = t & 63, wid = t >> 6;
  const int wm = wid >> 1, wn = wid & 1, i = lane & 31, h = lane >> 5;
  const int m0 = blockIdx.x * BM, n0 = blockIdx.y * IC_BN;
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

  float4 av[MT], bv;
  // every load goes out unconditionally from a clamped address and is zeroed by a select: a branch around a load
  // makes hipcc drain vmcnt at the join, i.e. serialises the prefetch
  auto load = [&](int it) {
    int tap, c;
    bool kvalid;
    if (CMODE == 0) {
      tap = it / a.cchunks;
      c = (it - tap * a.cchunks) * 16 + 4 * kq;
      kvalid = c < a.Cin;
    } else {
      tap = it * 4 + kq;
      c = 0;
      kvalid = tap < a.kh * a.kw;
    }
    if (!kvalid) { tap = 0; c = 0; }
    const int dy = tap / a.kw, dx = tap - dy * a.kw;
#pragma unroll
    for (int r = 0; r < MT; ++r) {
      const int iy = iy0[r] + dy, ix = ix0[r] + dx;
      const bool ok = rvalid[r] && kvalid && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
      const int64_t off = ok ? (int64_t)(pixbase[r] + iy * a.W + ix) : 0;
      float4 v;
      if (CMODE == 0) {
        v = *reinterpret_cast<const float4*>(a.x + off * a.ldx + c);
        if (a.xmask) {
          const float4 mk = *reinterpret_cast<const float4*>(a.xmask + off * a.ldm + c);
          v.x = mk.x > 0.f ? v.x : 0.f; v.y = mk.y > 0.f ? v.y : 0.f;
          v.z = mk.z > 0.f ? v.z : 0.f; v.w = mk.w > 0.f ? v.w : 0.f;
        }
      } else {
        const float* p = a.x + off * a.ldx;
        v.x = p[0];
        v.y = a.Cin > 1 ? p[a.Cin > 1 ? 1 : 0] : 0.f;
        v.z = a.Cin > 2 ? p[a.Cin > 2 ? 2 : 0] : 0.f;
        v.w = a.Cin > 3 ? p[a.Cin > 3 ? 3 : 0] : 0.f;
      }
      if (!ok) v = make_float4(0.f, 0.f, 0.f, 0.f);
      av[r] = v;
    }
    bv = wp4[(int64_t)it * slab4];
  };
  auto stage = [&](int buf) {
#pragma unroll
    for (int r = 0; r < MT; ++r)
      *reinterpret_cast<float4*>(As + (buf * BM + srow + 64 * r) * IC_LS + 4 * kq) = av[r];
    *reinterpret_cast<float4*>(Bs + (buf * IC_BN + srow) * IC_LS + 4 * kq) = bv;
  };

  f32x16 acc[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[mt][r] = 0.f;

  load(0);
  stage(0);
  __syncthreads();
  for (int it = 0; it < a.nchunks; ++it) {
    const int buf = it & 1;
    if (it + 1 < a.nchunks) load(it + 1);          // in flight under this chunk's MFMAs
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
    if (it + 1 < a.nchunks) stage(buf ^ 1);        // (read during chunk it-1: every wave is past that barrier)
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
// exists, sum_co gy[o][co] (y_act[o][co] > 0) w[ky][kx][c][co]
__global__ void __launch_bounds__(256) conv2d_small_dgrad_kernel(const float* __restrict__ gy, int ldg,
                                                                 const float* __restrict__ yact, int lda,
                                                                 const float* __restrict__ w, float* __restrict__ gx,
                                                                 int B, int H, int W, int Ho, int Wo, int Ci, int Co,
                                                                 int kh, int kw, int stride, int pad_t, int pad_l) {
  extern __shared__ __attribute__((aligned(16))) float sw[];   // [taps][Co][4]
  const int t = threadIdx.x;
  for (int idx = t; idx < kh * kw * Co; idx += 256) {
    const int tap = idx / Co, co = idx - tap * Co;
#pragma unroll
    for (int c = 0; c < 4; ++c) sw[idx * 4 + c] = c < Ci ? w[((int64_t)tap * Ci + c) * Co + co] : 0.f;
  }
  __syncthreads();
  const int64_t p = (int64_t)blockIdx.x * 256 + t;
  if (p >= (int64_t)B * H * W) return;
  const int b = (int)(p / ((int64_t)H * W));
  const int rem = (int)(p - (int64_t)b * H * W);
  const int iy = rem / W, ix = rem - iy * W;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int ky = 0; ky < kh; ++ky) {
    const int ny = iy + pad_t - ky;
    if (ny < 0 || ny % stride) continue;
    const int oy = ny / stride;
    if (oy >= Ho) continue;
    for (int kx = 0; kx < kw; ++kx) {
      const int nx = ix + pad_l - kx;
      if (nx < 0 || nx % stride) continue;
      const int ox = nx / stride;
      if (ox >= Wo) continue;
      const int64_t o = ((int64_t)b * Ho + oy) * Wo + ox;
      const float* gp = gy + o * ldg;
      const float* ap = yact ? yact + o * lda : nullptr;
      const float4* wt = reinterpret_cast<const float4*>(sw) + (ky * kw + kx) * Co;
      for (int co = 0; co < Co; co += 4) {
        float4 g = *reinterpret_cast<const float4*>(gp + co);
        if (ap) {
          const float4 m = *reinterpret_cast<const float4*>(ap + co);
          g.x = m.x > 0.f ? g.x : 0.f; g.y = m.y > 0.f ? g.y : 0.f;
          g.z = m.z > 0.f ? g.z : 0.f; g.w = m.w > 0.f ? g.w : 0.f;
        }
        const float4 w0 = wt[co], w1 = wt[co + 1], w2 = wt[co + 2], w3 = wt[co + 3];
        acc.x += g.x * w0.x + g.y * w1.x + g.z * w2.x + g.w * w3.x;
        acc.y += g.x * w0.y + g.y * w1.y + g.z * w2.y + g.w * w3.y;
        acc.z += g.x * w0.z + g.y * w1.z + g.z * w2.z + g.w * w3.z;
        acc.w += g.x * w0.w + g.y * w1.w + g.z * w2.w + g.w * w3.w;
      }
    }
  }
  float* o = gx + p * Ci;
  o[0] = acc.x;
  if (Ci > 1) o[1] = acc.y;
  if (Ci > 2) o[2] = acc.z;
  if (Ci > 3) o[3] = acc.w;
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
                                                           int C4, int stride, int pad_t, int pad_l, int accumulate) {
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

int nfs_conv2d_fwd(const float* x, int ldx, const float* x_mask, int ldm, const float* packed, const float* bias,
                   float* y, int ldy, float* y_pre, int ldp, int B, int H, int W, int Cin, int Cout, int kh, int kw,
                   int stride, int relu, int accumulate, nfs_stream_t stream) {
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
  a.cchunks = conv2d_cchunks(Cin);
  a.nchunks = conv2d_nchunks(kh, kw, Cin);
  a.Npad = conv2d_npad(Cout);
  const int ntiles = a.Npad / IC_BN;
  // 128-row tiles once they still give every CU two blocks; 64-row tiles for the small late layers
  const bool big = (M / 128) * ntiles >= 512;
  hipStream_t s = as_stream(stream);
  if (big) {
    const size_t lds = sizeof(float) * (size_t)(128 * (IC_BN + 4));          // >= 2 * (128 + 64) * 20
    dim3 grid((unsigned)((M + 127) / 128), ntiles);
    if (Cin <= 4) conv2d_mfma_kernel<128, 1><<<grid, 256, lds, s>>>(a);
    else conv2d_mfma_kernel<128, 0><<<grid, 256, lds, s>>>(a);
  } else {
    const size_t lds = sizeof(float) * (size_t)(2 * (64 + IC_BN) * IC_LS);   // >= 64 * 68
    dim3 grid((unsigned)((M + 63) / 64), ntiles);
    if (Cin <= 4) conv2d_mfma_kernel<64, 1><<<grid, 256, lds, s>>>(a);
    else conv2d_mfma_kernel<64, 0><<<grid, 256, lds, s>>>(a);
  }
  return check_launch("nfs_conv2d_fwd");
}

int nfs_conv2d_dgrad_small(const float* gy, int ldg, const float* y_act, int lda, const float* w_hwio, float* gx, int B,
                           int H, int W, int Ci, int Co, int kh, int kw, int stride, nfs_stream_t stream) {
  NFS_REQUIRE(gy && w_hwio && gx, "nfs_conv2d_dgrad_small: null pointer");
  NFS_REQUIRE(B > 0 && H > 0 && W > 0 && Ci > 0 && Ci <= 4 && Co > 0 && Co % 4 == 0,
              "nfs_conv2d_dgrad_small: 1..4 input channels, Co a multiple of 4");
  NFS_REQUIRE(kh > 0 && kw > 0 && kh <= 7 && kw <= 7 && (stride == 1 || stride == 2),
              "nfs_conv2d_dgrad_small: filter up to 7x7, stride 1 or 2");
  NFS_REQUIRE(ldg % 4 == 0 && ldg >= Co && (!y_act || (lda % 4 == 0 && lda >= Co)), "nfs_conv2d_dgrad_small: bad row stride");
  NFS_REQUIRE((((uintptr_t)gy | (uintptr_t)y_act) & 15) == 0, "nfs_conv2d_dgrad_small: pointers must be 16-byte aligned");
  const size_t lds = sizeof(float) * (size_t)kh * kw * Co * 4;
  NFS_REQUIRE(lds <= 64 * 1024, "nfs_conv2d_dgrad_small: filters do not fit 64 KB of LDS");
  int Ho, Wo, pt, pl;
  same_pad(H, kh, stride, Ho, pt);
  same_pad(W, kw, stride, Wo, pl);
  conv2d_small_dgrad_kernel<<<blocks_for((int64_t)B * H * W, 256), 256, lds, as_stream(stream)>>>(
      gy, ldg, y_act, lda, w_hwio, gx, B, H, W, Ho, Wo, Ci, Co, kh, kw, stride, pt, pl);
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
                     int accumulate, nfs_stream_t stream) {
  NFS_REQUIRE(gy && gx && arg, "nfs_maxpool3_bwd: null pointer");
  NFS_REQUIRE(B > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0, "nfs_maxpool3_bwd: C must be a positive multiple of 4");
  NFS_REQUIRE(stride == 1 || stride == 2, "nfs_maxpool3_bwd: stride 1 or 2");
  NFS_REQUIRE((((uintptr_t)gx | (uintptr_t)gy) & 15) == 0 && ((uintptr_t)arg & 3) == 0, "nfs_maxpool3_bwd: misaligned pointer");
  int Ho, Wo, pt, pl;
  same_pad(H, 3, stride, Ho, pt);
  same_pad(W, 3, stride, Wo, pl);
  maxpool3_bwd_kernel<<<blocks_for((int64_t)B * H * W * (C / 4), 256), 256, 0, as_stream(stream)>>>(
      gy, reinterpret_cast<const uint32_t*>(arg), gx, B, H, W, Ho, Wo, C / 4, stride, pt, pl, accumulate);
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
