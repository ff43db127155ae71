// Histogram loss of _loss (styler_base.py:187-209) with util.histogram_match_tf (util.py:317-399):
//   per image i and channel j:  matched = histogram_match(feature[i,...,j], template[i,...,j])  (255 fixed-width bins
//   over the joint value range, template CDF inverted by linear interpolation of the bin index, bin-centre values),
//   loss += sum((feature - matched)^2);  `matched` carries no gradient (py_func / integer casts / tf.range in the
//   reference graph), so d loss / d feature = 2 (feature - matched).
// Deliberate divergence from the mounted reference (also DESIGN.md section 5): the caller passes the template
// features OF THE HIST LAYER and the weight w_hist_layer.  The reference's unmasked branch (styler_base.py:203, 207) reads
// `style_feature` / `w_style_layer` -- stale loop variables of the style loop above it, i.e. the LAST style layer's
// placeholder and weight -- so its results coincide with this only for hist_layer == [style_layer[-1]] and
// w_hist_layer == [w_style_layer[-1]].  The masked branch (196-201) uses the hist placeholder and is followed as written.
// One block per (image, channel): min/max -> two 255-bin histograms in LDS (integer atomics) -> quantiles (double, as
// NumPy's cumsum / total) -> the 255-entry lookup table -> one pass applying it.  The activations of a channel are
// read three times by the same block (L2-resident); everything else lives in LDS.
#include "common.h"

namespace nfs {

constexpr int HB = 255;   // hist_bins (util.py:317)

__device__ __forceinline__ float block_min(float v, float* red) {
  return -block_max(-v, red);
}

// ``mask`` [B,HW] (styler_base.py:104-125, 196-201: the bicubic-resized density mask): pixels where it is 0 are removed from
// the source (tf.boolean_mask) -- they count neither for the value range nor for the source histogram nor for the loss;
// the template stays whole.  A channel with nothing to match -- flat (max == min over source and template: the
// reference's tf.range(min, max, 0) has no defined result) or with every source pixel masked out -- contributes loss 0
// and gradient 0.
__global__ void __launch_bounds__(256) hist_loss_kernel(const float* __restrict__ feat, const float* __restrict__ templ,
                                                        const float* __restrict__ mask, float* __restrict__ loss_acc,
                                                        float* __restrict__ g_acc, int B, int Bt, int HW, int HWt, int C,
                                                        float weight, int relu_mask) {
  __shared__ float red[16];
  __shared__ unsigned hs[256], ht[256];
  __shared__ double sq[256], tq[256];
  __shared__ float lut[256];
  const int b = blockIdx.x / C, c = blockIdx.x - b * C;
  const int bt = b < Bt ? b : Bt - 1;
  const float* s = feat + (int64_t)b * HW * C + c;
  const float* tp = templ + (int64_t)bt * HWt * C + c;
  const float* mk = mask ? mask + (int64_t)b * HW : nullptr;
  const int t = threadIdx.x;
  // 1. joint value range (util.py:326-327)
  float lo = 3.0e38f, hi = -3.0e38f;
  int live = 0;
  for (int p = t; p < HW; p += 256) {
    if (mk && mk[p] == 0.f) continue;
    const float v = s[(int64_t)p * C]; lo = fminf(lo, v); hi = fmaxf(hi, v); live = 1;
  }
  for (int p = t; p < HWt; p += 256) { const float v = tp[(int64_t)p * C]; lo = fminf(lo, v); hi = fmaxf(hi, v); }
  const float vmax = block_max(hi, red);
  const float vmin = block_min(lo, red);
  const float any_live = block_max((float)live, red);
  if (!(vmax > vmin) || any_live == 0.f) return;           // nothing to match (uniform across the block): loss 0, gradient 0
  const float range = vmax - vmin;
  const float delta = range / (float)HB;
  // 2. tf.histogram_fixed_width: index = floor(nbins * (v - min) / (max - min)) clipped to [0, nbins-1]
  hs[t] = 0u; ht[t] = 0u;
  __syncthreads();
  for (int p = t; p < HW; p += 256) {
    if (mk && mk[p] == 0.f) continue;
    const float sc = (s[(int64_t)p * C] - vmin) / range;
    int k = (int)floorf((float)HB * sc);
    k = k < 0 ? 0 : (k > HB - 1 ? HB - 1 : k);
    atomicAdd(&hs[k], 1u);
  }
  for (int p = t; p < HWt; p += 256) {
    const float sc = (tp[(int64_t)p * C] - vmin) / range;
    int k = (int)floorf((float)HB * sc);
    k = k < 0 ? 0 : (k > HB - 1 ? HB - 1 : k);
    atomicAdd(&ht[k], 1u);
  }
  __syncthreads();
  // 3. quantiles = cumsum / total (util.py:352-356), in double like NumPy's true division of int64 counts
  if (t == 0) {
    unsigned long long a = 0, bq = 0;
    for (int k = 0; k < HB; ++k) { a += hs[k]; bq += ht[k]; sq[k] = (double)a; tq[k] = (double)bq; }
    const double ta = (double)a, tb = (double)bq;
    for (int k = 0; k < HB; ++k) { sq[k] /= ta; tq[k] /= tb; }
  }
  __syncthreads();
  // 4. nearest_indices = round(interp1d(t_quantiles, arange(nbins), fill (0, nbins-1))(s_quantiles)) (util.py:358-362).
  //    SciPy's linear interp1d on 1-D float data delegates to numpy.interp: j = LAST index with xp[j] <= x (rightmost
  //    among equal quantiles, i.e. past the template's empty bins), fp[j] + (x - xp[j]) / (xp[j+1] - xp[j]);
  //    then round-half-even
  if (t < HB) {
    const double x = sq[t];
    int n;
    if (x < tq[0]) n = 0;
    else if (x >= tq[HB - 1]) n = HB - 1;
    else {
      int l = 0, r = HB;                       // first index with tq[idx] > x
      while (l < r) { const int mid = (l + r) >> 1; if (tq[mid] <= x) l = mid + 1; else r = mid; }
      const int j = l - 1;                     // 0 <= j <= HB-2, tq[j] <= x < tq[j+1]
      const double y = (double)j + (x - tq[j]) / (tq[j + 1] - tq[j]);
      n = (int)rint(y);
      n = n < 0 ? 0 : (n > HB - 1 ? HB - 1 : n);
    }
    // hist_range[n] = (min + delta n) + delta / 2 (util.py:331-334), float32 like the graph
    lut[t] = (vmin + delta * (float)n) + delta * 0.5f;
  }
  __syncthreads();
  // 5. matched = lut[clip(int((source - min) / delta))];  loss and gradient
  float part = 0.f;
  for (int p = t; p < HW; p += 256) {
    if (mk && mk[p] == 0.f) continue;
    const int64_t o = (int64_t)p * C;
    const float v = s[o];
    int k = (int)((v - vmin) / delta);
    k = k < 0 ? 0 : (k > HB - 1 ? HB - 1 : k);
    const float d = v - lut[k];
    part += d * d;
    if (g_acc && (!relu_mask || v > 0.f)) g_acc[(int64_t)b * HW * C + c + o] += 2.f * weight * d;
  }
  part = block_sum(part, red);
  if (t == 0) atomicAdd(loss_acc + b, weight * part);
}

// ---- the same loss for images of FEW channels (the default hist layer is the loss-net INPUT: 3 channels) -------------
// One block per (image, channel) is 3 blocks for a 300 x 450 image -- 0.7 ms, 3.9 ms at 512 x 1024 -- so for C <= 4 the
// work is spread over the pixels instead: a thread owns whole pixels (C contiguous floats), the per-(image, channel)
// state -- value range as order-preserving integers, the two 255-bin histograms, the lookup table -- lives in a
// caller-provided workspace, and the steps of hist_loss_kernel become launches:
//   init -> range (block min / max, integer atomics) -> histograms (LDS, flushed with integer atomics) -> quantiles +
//   table (one block per (image, channel), the code of steps 3-4 above) -> apply (loss as per-block partial sums) ->
//   fixed-order total.  Integer atomics and fixed-order sums only: deterministic, and the same bins, table and
//   matched values as the per-channel kernel (the same float expressions per element).
constexpr int HW_CMAX = 4;
struct HistState {                         // per (image, channel), in the workspace
  unsigned vmin_enc, vmax_enc, live, pad;
  float vmin, delta;
  unsigned skip, pad2;
  unsigned hs[256], ht[256];
  float lut[256];
};
__device__ __forceinline__ unsigned enc_ordered(float f) {      // monotone float -> unsigned
  const unsigned u = __float_as_uint(f);
  return u ^ ((u >> 31) ? 0xffffffffu : 0x80000000u);
}
__device__ __forceinline__ float dec_ordered(unsigned e) {
  return __uint_as_float(e ^ ((e >> 31) ? 0x80000000u : 0xffffffffu));
}

__global__ void __launch_bounds__(256) hist_wide_init_kernel(HistState* st, int n) {
  const int i = blockIdx.x, t = threadIdx.x;
  if (i >= n) return;
  st[i].hs[t] = 0u; st[i].ht[t] = 0u;
  if (t == 0) { st[i].vmin_enc = 0xffffffffu; st[i].vmax_enc = 0u; st[i].live = 0u; st[i].skip = 0u; }
}

template <int C>
__global__ void __launch_bounds__(256) hist_wide_range_kernel(const float* __restrict__ feat, const float* __restrict__ templ,
                                                              const float* __restrict__ mask, HistState* st, int Bt,
                                                              int HW, int HWt, int ld) {
  __shared__ float red[16];
  const int b = blockIdx.y, bt = b < Bt ? b : Bt - 1, t = threadIdx.x;
  const int c0 = blockIdx.z * C;                     // this block's channel window of the ld-float rows
  const float* s = feat + (int64_t)b * HW * ld + c0;
  const float* tp = templ + (int64_t)bt * HWt * ld + c0;
  const float* mk = mask ? mask + (int64_t)b * HW : nullptr;
  float lo[C], hi[C];
#pragma unroll
  for (int c = 0; c < C; ++c) { lo[c] = 3.0e38f; hi[c] = -3.0e38f; }
  int live = 0;
  const int stride = gridDim.x * 256;
  for (int p = blockIdx.x * 256 + t; p < HW; p += stride) {
    if (mk && mk[p] == 0.f) continue;
    live = 1;
#pragma unroll
    for (int c = 0; c < C; ++c) { const float v = s[(int64_t)p * ld + c]; lo[c] = fminf(lo[c], v); hi[c] = fmaxf(hi[c], v); }
  }
  for (int p = blockIdx.x * 256 + t; p < HWt; p += stride) {
#pragma unroll
    for (int c = 0; c < C; ++c) { const float v = tp[(int64_t)p * ld + c]; lo[c] = fminf(lo[c], v); hi[c] = fmaxf(hi[c], v); }
  }
  const float any_live = block_max((float)live, red);
#pragma unroll
  for (int c = 0; c < C; ++c) {
    const float vmax = block_max(hi[c], red);
    const float vmin = block_min(lo[c], red);
    if (t == 0) {
      HistState& h = st[b * ld + c0 + c];
      if (vmin < 3.0e38f) atomicMin(&h.vmin_enc, enc_ordered(vmin));
      if (vmax > -3.0e38f) atomicMax(&h.vmax_enc, enc_ordered(vmax));
      if (any_live != 0.f) atomicOr(&h.live, 1u);
    }
  }
}

template <int C>
__global__ void __launch_bounds__(256) hist_wide_count_kernel(const float* __restrict__ feat, const float* __restrict__ templ,
                                                              const float* __restrict__ mask, HistState* st, int Bt,
                                                              int HW, int HWt, int ld) {
  __shared__ unsigned hs[C][256], ht[C][256];
  __shared__ float smin[C], srange[C];
  __shared__ unsigned sskip[C];
  const int b = blockIdx.y, bt = b < Bt ? b : Bt - 1, t = threadIdx.x;
  const int c0 = blockIdx.z * C;
  if (t < C) {
    HistState& h = st[b * ld + c0 + t];
    const float vmin = dec_ordered(h.vmin_enc), vmax = dec_ordered(h.vmax_enc);
    smin[t] = vmin; srange[t] = vmax - vmin;
    sskip[t] = (!(vmax > vmin) || h.live == 0u) ? 1u : 0u;       // nothing to match: loss 0, gradient 0
  }
#pragma unroll
  for (int c = 0; c < C; ++c) { hs[c][t] = 0u; ht[c][t] = 0u; }
  __syncthreads();
  const float* s = feat + (int64_t)b * HW * ld + c0;
  const float* tp = templ + (int64_t)bt * HWt * ld + c0;
  const float* mk = mask ? mask + (int64_t)b * HW : nullptr;
  const int stride = gridDim.x * 256;
  for (int p = blockIdx.x * 256 + t; p < HW; p += stride) {
    if (mk && mk[p] == 0.f) continue;
#pragma unroll
    for (int c = 0; c < C; ++c) {
      if (sskip[c]) continue;
      const float sc = (s[(int64_t)p * ld + c] - smin[c]) / srange[c];
      int k = (int)floorf((float)HB * sc);
      k = k < 0 ? 0 : (k > HB - 1 ? HB - 1 : k);
      atomicAdd(&hs[c][k], 1u);
    }
  }
  for (int p = blockIdx.x * 256 + t; p < HWt; p += stride) {
#pragma unroll
    for (int c = 0; c < C; ++c) {
      if (sskip[c]) continue;
      const float sc = (tp[(int64_t)p * ld + c] - smin[c]) / srange[c];
      int k = (int)floorf((float)HB * sc);
      k = k < 0 ? 0 : (k > HB - 1 ? HB - 1 : k);
      atomicAdd(&ht[c][k], 1u);
    }
  }
  __syncthreads();
#pragma unroll
  for (int c = 0; c < C; ++c) {
    if (sskip[c]) continue;
    if (hs[c][t]) atomicAdd(&st[b * ld + c0 + c].hs[t], hs[c][t]);
    if (ht[c][t]) atomicAdd(&st[b * ld + c0 + c].ht[t], ht[c][t]);
  }
}

// steps 3-4 of hist_loss_kernel on the global histograms: one block per (image, channel)
__global__ void __launch_bounds__(256) hist_wide_table_kernel(HistState* st) {
  __shared__ double sq[256], tq[256];
  HistState& h = st[blockIdx.x];
  const int t = threadIdx.x;
  const float vmin = dec_ordered(h.vmin_enc), vmax = dec_ordered(h.vmax_enc);
  if (!(vmax > vmin) || h.live == 0u) { if (t == 0) h.skip = 1u; return; }
  const float range = vmax - vmin, delta = range / (float)HB;
  if (t == 0) {
    unsigned long long a = 0, bq = 0;
    for (int k = 0; k < HB; ++k) { a += h.hs[k]; bq += h.ht[k]; sq[k] = (double)a; tq[k] = (double)bq; }
    const double ta = (double)a, tb = (double)bq;
    for (int k = 0; k < HB; ++k) { sq[k] /= ta; tq[k] /= tb; }
    h.vmin = vmin; h.delta = delta; h.skip = 0u;
  }
  __syncthreads();
  if (t < HB) {
    const double x = sq[t];
    int n;
    if (x < tq[0]) n = 0;
    else if (x >= tq[HB - 1]) n = HB - 1;
    else {
      int l = 0, r = HB;
      while (l < r) { const int mid = (l + r) >> 1; if (tq[mid] <= x) l = mid + 1; else r = mid; }
      const int j = l - 1;
      const double y = (double)j + (x - tq[j]) / (tq[j + 1] - tq[j]);
      n = (int)rint(y);
      n = n < 0 ? 0 : (n > HB - 1 ? HB - 1 : n);
    }
    h.lut[t] = (vmin + delta * (float)n) + delta * 0.5f;
  }
}

template <int C>
__global__ void __launch_bounds__(256) hist_wide_apply_kernel(const float* __restrict__ feat, const float* __restrict__ mask,
                                                              const HistState* __restrict__ st, float* __restrict__ g_acc,
                                                              float* __restrict__ parts, int HW, int ld, float weight,
                                                              int relu_mask) {
  __shared__ float red[16];
  __shared__ float lut[C][256];
  __shared__ float smin[C], sdelta[C];
  __shared__ unsigned sskip[C];
  const int b = blockIdx.y, t = threadIdx.x, c0 = blockIdx.z * C;
#pragma unroll
  for (int c = 0; c < C; ++c) lut[c][t] = st[b * ld + c0 + c].lut[t];
  if (t < C) { smin[t] = st[b * ld + c0 + t].vmin; sdelta[t] = st[b * ld + c0 + t].delta; sskip[t] = st[b * ld + c0 + t].skip; }
  __syncthreads();
  const float* s = feat + (int64_t)b * HW * ld + c0;
  const float* mk = mask ? mask + (int64_t)b * HW : nullptr;
  float part = 0.f;
  const int stride = gridDim.x * 256;
  for (int p = blockIdx.x * 256 + t; p < HW; p += stride) {
    if (mk && mk[p] == 0.f) continue;
#pragma unroll
    for (int c = 0; c < C; ++c) {
      if (sskip[c]) continue;
      const int64_t o = (int64_t)p * ld + c;
      const float v = s[o];
      int k = (int)((v - smin[c]) / sdelta[c]);
      k = k < 0 ? 0 : (k > HB - 1 ? HB - 1 : k);
      const float d = v - lut[c][k];
      part += d * d;
      if (g_acc && (!relu_mask || v > 0.f)) g_acc[(int64_t)b * HW * ld + c0 + o] += 2.f * weight * d;
    }
  }
  part = block_sum(part, red);
  if (t == 0) parts[((int64_t)b * gridDim.z + blockIdx.z) * gridDim.x + blockIdx.x] = part;
}

__global__ void __launch_bounds__(256) hist_wide_total_kernel(const float* __restrict__ parts, float* __restrict__ loss_acc,
                                                              int nblk, float weight) {
  __shared__ double tot[256];
  const int b = blockIdx.x, t = threadIdx.x;
  double a = 0.0;
  for (int i = t; i < nblk; i += 256) a += (double)parts[(int64_t)b * nblk + i];
  tot[t] = a;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if (t < w) tot[t] += tot[t + w];
    __syncthreads();
  }
  if (t == 0) loss_acc[b] += weight * (float)tot[0];
}

// ---- style mask (styler_base.py:165-173): features weighted by the bicubic-resized density mask --------------------
// TF-1 legacy ResizeBicubic (align_corners = False, no half-pixel centres): src = dst * in/out, taps floor-1..floor+2
// clamped to the image, Keys weights (A = -0.75) from the 1024-entry table looked up at round(frac * 1024).
__device__ __forceinline__ void bicubic_taps(int o, int n_in, int n_out, int* idx, float* w) {
  const float a = -0.75f;
  const float src = (float)o * ((float)n_in / (float)n_out);
  const float fl = floorf(src);
  const int loc = (int)fl;
  const float t = rintf((src - fl) * 1024.f) / 1024.f;          // table quantisation of the fraction
  const float t1 = t + 1.f, u = 1.f - t, u1 = u + 1.f;
  w[0] = ((a * t1 - 5.f * a) * t1 + 8.f * a) * t1 - 4.f * a;
  w[1] = ((a + 2.f) * t - (a + 3.f)) * t * t + 1.f;
  w[2] = ((a + 2.f) * u - (a + 3.f)) * u * u + 1.f;
  w[3] = ((a * u1 - 5.f * a) * u1 + 8.f * a) * u1 - 4.f * a;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int i = loc - 1 + k;
    idx[k] = i < 0 ? 0 : (i > n_in - 1 ? n_in - 1 : i);
  }
}

__global__ void __launch_bounds__(256) resize_bicubic_kernel(const float* __restrict__ x, float* __restrict__ out, int B,
                                                             int H, int W, int C, int oh, int ow) {
  const int64_t n = (int64_t)B * oh * ow * C;
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= n) return;
  const int c = (int)(gid % C);
  int64_t v = gid / C;
  const int xo = (int)(v % ow); v /= ow;
  const int yo = (int)(v % oh);
  const int b = (int)(v / oh);
  int iy[4], ix[4];
  float wy[4], wx[4];
  bicubic_taps(yo, H, oh, iy, wy);
  bicubic_taps(xo, W, ow, ix, wx);
  const float* im = x + (int64_t)b * H * W * C + c;
  // rows first, then columns (the order of the two passes of the reference restatement)
  float colv[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    float r = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) r += wy[k] * im[((int64_t)iy[k] * W + ix[q]) * C];
    colv[q] = r;
  }
  out[gid] = ((wx[0] * colv[0] + wx[1] * colv[1]) + wx[2] * colv[2]) + wx[3] * colv[3];
}

// Fm = F * m (mask broadcast over channels) and scale[b] = 1 / (2 * sum_pixels(m) * C)   (styler_base.py:167-169)
__global__ void __launch_bounds__(256) style_mask_apply_kernel(const float* __restrict__ F, const float* __restrict__ m,
                                                               float* __restrict__ Fm, float* __restrict__ scale, int HW,
                                                               int C) {
  __shared__ float red[16];
  const int b = blockIdx.y;
  const int64_t base = (int64_t)b * HW * C;
  const int64_t n = (int64_t)HW * C;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    Fm[base + i] = F[base + i] * m[(int64_t)b * HW + i / C];
  if (blockIdx.x == 0) {                                  // one block per image also sums the mask (fixed order)
    float s = 0.f;
    for (int p = threadIdx.x; p < HW; p += blockDim.x) s += m[(int64_t)b * HW + p];
    s = block_sum(s, red);
    if (threadIdx.x == 0) scale[b] = 1.f / (2.f * s * (float)C);
  }
}

// dF = dFm * m * (F > 0): gradient wrt the pre-activation of the masked feature
__global__ void __launch_bounds__(256) style_mask_bwd_kernel(const float* __restrict__ dFm, const float* __restrict__ m,
                                                             const float* __restrict__ F, float* __restrict__ dF, int64_t n,
                                                             int C) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  dF[i] = F[i] > 0.f ? dFm[i] * m[i / C] : 0.f;
}

}  // namespace nfs

using namespace nfs;

static int hist_wide_blocks(int HW, int HWt) {
  const int n = HW > HWt ? HW : HWt;
  int nb = (n + 256 * 4 - 1) / (256 * 4);                 // ~4 pixels per thread
  return nb < 1 ? 1 : (nb > 1024 ? 1024 : nb);
}

template <int C>
static void hist_wide_launch(const float* feat, const float* templ, const float* mask, float* loss_acc, float* g_acc,
                             HistState* st, float* parts, int B, int Bt, int HW, int HWt, int ld, float weight,
                             int relu_mask, hipStream_t s) {
  const int nb = hist_wide_blocks(HW, HWt), groups = ld / C;
  const dim3 grid(nb, B, groups);
  hipLaunchKernelGGL(hist_wide_init_kernel, dim3(B * ld), dim3(256), 0, s, st, B * ld);
  hipLaunchKernelGGL(hist_wide_range_kernel<C>, grid, dim3(256), 0, s, feat, templ, mask, st, Bt, HW, HWt, ld);
  hipLaunchKernelGGL(hist_wide_count_kernel<C>, grid, dim3(256), 0, s, feat, templ, mask, st, Bt, HW, HWt, ld);
  hipLaunchKernelGGL(hist_wide_table_kernel, dim3(B * ld), dim3(256), 0, s, st);
  hipLaunchKernelGGL(hist_wide_apply_kernel<C>, grid, dim3(256), 0, s, feat, mask, st, g_acc, parts, HW, ld, weight,
                     relu_mask);
  hipLaunchKernelGGL(hist_wide_total_kernel, dim3(B), dim3(256), 0, s, parts, loss_acc, nb * groups, weight);
}

extern "C" {

int nfs_hist_loss_masked(const float* feat, const float* templ, const float* mask, float* loss_acc, float* g_acc, int B,
                         int Bt, int HW, int HWt, int C, float weight, int relu_mask, nfs_stream_t stream) {
  NFS_REQUIRE(feat && templ && loss_acc, "nfs_hist_loss: null pointer");
  NFS_REQUIRE(B > 0 && Bt > 0 && HW > 0 && HWt > 0 && C > 0, "nfs_hist_loss: non-positive dimension");
  NFS_REQUIRE((int64_t)B * C < (int64_t)1 << 30, "nfs_hist_loss: too many (image, channel) pairs");
  hipLaunchKernelGGL(hist_loss_kernel, dim3((unsigned)(B * C)), dim3(256), 0, as_stream(stream), feat, templ, mask,
                     loss_acc, g_acc, B, Bt, HW, HWt, C, weight, relu_mask);
  return check_launch("nfs_hist_loss");
}

int nfs_hist_loss(const float* feat, const float* templ, float* loss_acc, float* g_acc, int B, int Bt, int HW, int HWt,
                  int C, float weight, int relu_mask, nfs_stream_t stream) {
  return nfs_hist_loss_masked(feat, templ, nullptr, loss_acc, g_acc, B, Bt, HW, HWt, C, weight, relu_mask, stream);
}

int64_t nfs_hist_loss_wide_workspace_floats(int B, int C, int HW, int HWt) {
  if (B <= 0 || C <= 0 || (C > HW_CMAX && C % HW_CMAX) || HW <= 0 || HWt <= 0) return -1;
  const int groups = C <= HW_CMAX ? 1 : C / HW_CMAX;
  return (int64_t)B * C * (int64_t)(sizeof(HistState) / sizeof(float)) + (int64_t)B * groups * hist_wide_blocks(HW, HWt);
}

int nfs_hist_loss_wide(const float* feat, const float* templ, const float* mask, float* loss_acc, float* g_acc,
                       float* workspace, int64_t workspace_floats, int B, int Bt, int HW, int HWt, int C, float weight,
                       int relu_mask, nfs_stream_t stream) {
  NFS_REQUIRE(feat && templ && loss_acc && workspace, "nfs_hist_loss_wide: null pointer");
  NFS_REQUIRE(B > 0 && Bt > 0 && HW > 0 && HWt > 0 && C > 0 && (C <= HW_CMAX || C % HW_CMAX == 0),
              "nfs_hist_loss_wide: 1..4 channels or a multiple of 4, positive sizes");
  NFS_REQUIRE(C <= HW_CMAX || (((uintptr_t)feat | (uintptr_t)templ) & 15) == 0, "nfs_hist_loss_wide: misaligned features");
  NFS_REQUIRE(((uintptr_t)workspace & 15) == 0 && workspace_floats >= nfs_hist_loss_wide_workspace_floats(B, C, HW, HWt),
              "nfs_hist_loss_wide: workspace too small or misaligned (nfs_hist_loss_wide_workspace_floats)");
  HistState* st = reinterpret_cast<HistState*>(workspace);
  float* parts = workspace + (int64_t)B * C * (sizeof(HistState) / sizeof(float));
  hipStream_t s = as_stream(stream);
  const int per = C <= HW_CMAX ? C : HW_CMAX;              // channels per thread; rows are C floats long
  switch (per) {
    case 1: hist_wide_launch<1>(feat, templ, mask, loss_acc, g_acc, st, parts, B, Bt, HW, HWt, C, weight, relu_mask, s); break;
    case 2: hist_wide_launch<2>(feat, templ, mask, loss_acc, g_acc, st, parts, B, Bt, HW, HWt, C, weight, relu_mask, s); break;
    case 3: hist_wide_launch<3>(feat, templ, mask, loss_acc, g_acc, st, parts, B, Bt, HW, HWt, C, weight, relu_mask, s); break;
    default: hist_wide_launch<4>(feat, templ, mask, loss_acc, g_acc, st, parts, B, Bt, HW, HWt, C, weight, relu_mask, s); break;
  }
  return check_launch("nfs_hist_loss_wide");
}

int nfs_resize_bicubic_tf1(const float* x, float* out, int B, int H, int W, int C, int oh, int ow, nfs_stream_t stream) {
  NFS_REQUIRE(x && out, "nfs_resize_bicubic_tf1: null pointer");
  NFS_REQUIRE(B > 0 && H > 0 && W > 0 && C > 0 && oh > 0 && ow > 0, "nfs_resize_bicubic_tf1: non-positive dimension");
  const int64_t n = (int64_t)B * oh * ow * C;
  hipLaunchKernelGGL(resize_bicubic_kernel, dim3(blocks_for(n, 256)), dim3(256), 0, as_stream(stream), x, out, B, H, W, C,
                     oh, ow);
  return check_launch("nfs_resize_bicubic_tf1");
}

int nfs_style_mask_apply(const float* F, const float* mask, float* Fm, float* scale, int B, int HW, int C,
                         nfs_stream_t stream) {
  NFS_REQUIRE(F && mask && Fm && scale, "nfs_style_mask_apply: null pointer");
  NFS_REQUIRE(B > 0 && HW > 0 && C > 0, "nfs_style_mask_apply: non-positive dimension");
  unsigned bx = blocks_for((int64_t)HW * C, 256 * 4);
  if (bx > 1024) bx = 1024;
  hipLaunchKernelGGL(style_mask_apply_kernel, dim3(bx, (unsigned)B), dim3(256), 0, as_stream(stream), F, mask, Fm, scale,
                     HW, C);
  return check_launch("nfs_style_mask_apply");
}

int nfs_style_mask_bwd(const float* dFm, const float* mask, const float* F, float* dF, int B, int HW, int C,
                       nfs_stream_t stream) {
  NFS_REQUIRE(dFm && mask && F && dF, "nfs_style_mask_bwd: null pointer");
  NFS_REQUIRE(B > 0 && HW > 0 && C > 0, "nfs_style_mask_bwd: non-positive dimension");
  const int64_t n = (int64_t)B * HW * C;
  hipLaunchKernelGGL(style_mask_bwd_kernel, dim3(blocks_for(n, 256)), dim3(256), 0, as_stream(stream), dFm, mask, F, dF, n,
                     C);
  return check_launch("nfs_style_mask_bwd");
}

}  // extern "C"
