// Histogram loss of _loss (styler_base.py:187-209) with util.histogram_match_tf (util.py:317-399):
//   per image i and channel j:  matched = histogram_match(feature[i,...,j], template[i,...,j])  (255 fixed-width bins
//   over the joint value range, template CDF inverted by linear interpolation of the bin index, bin-centre values),
//   loss += sum((feature - matched)^2);  `matched` carries no gradient (py_func / integer casts / tf.range in the
//   reference graph), so d loss / d feature = 2 (feature - matched).
// One block per (image, channel): min/max -> two 255-bin histograms in LDS (integer atomics) -> quantiles (double, as
// NumPy's cumsum / total) -> the 255-entry lookup table -> one pass applying it.  The activations of a channel are
// read three times by the same block (L2-resident); everything else lives in LDS.
#include "common.h"

namespace nfs {

constexpr int HB = 255;   // hist_bins (util.py:317)

__device__ __forceinline__ float block_min(float v, float* red) {
  return -block_max(-v, red);
}

__global__ void __launch_bounds__(256) hist_loss_kernel(const float* __restrict__ feat, const float* __restrict__ templ,
                                                        float* __restrict__ loss_acc, float* __restrict__ g_acc, int B,
                                                        int Bt, int HW, int HWt, int C, float weight, int relu_mask) {
  __shared__ float red[16];
  __shared__ unsigned hs[256], ht[256];
  __shared__ double sq[256], tq[256];
  __shared__ float lut[256];
  const int b = blockIdx.x / C, c = blockIdx.x - b * C;
  const int bt = b < Bt ? b : Bt - 1;
  const float* s = feat + (int64_t)b * HW * C + c;
  const float* tp = templ + (int64_t)bt * HWt * C + c;
  const int t = threadIdx.x;
  // 1. joint value range (util.py:326-327)
  float lo = 3.0e38f, hi = -3.0e38f;
  for (int p = t; p < HW; p += 256) { const float v = s[(int64_t)p * C]; lo = fminf(lo, v); hi = fmaxf(hi, v); }
  for (int p = t; p < HWt; p += 256) { const float v = tp[(int64_t)p * C]; lo = fminf(lo, v); hi = fmaxf(hi, v); }
  const float vmax = block_max(hi, red);
  const float vmin = block_min(lo, red);
  const float range = vmax - vmin;
  const float delta = range / (float)HB;
  // 2. tf.histogram_fixed_width: index = floor(nbins * (v - min) / (max - min)) clipped to [0, nbins-1]
  hs[t] = 0u; ht[t] = 0u;
  __syncthreads();
  for (int p = t; p < HW; p += 256) {
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

}  // namespace nfs

using namespace nfs;

extern "C" {

int nfs_hist_loss(const float* feat, const float* templ, float* loss_acc, float* g_acc, int B, int Bt, int HW, int HWt,
                  int C, float weight, int relu_mask, nfs_stream_t stream) {
  NFS_REQUIRE(feat && templ && loss_acc, "nfs_hist_loss: null pointer");
  NFS_REQUIRE(B > 0 && Bt > 0 && HW > 0 && HWt > 0 && C > 0, "nfs_hist_loss: non-positive dimension");
  NFS_REQUIRE((int64_t)B * C < (int64_t)1 << 30, "nfs_hist_loss: too many (image, channel) pairs");
  hipLaunchKernelGGL(hist_loss_kernel, dim3((unsigned)(B * C)), dim3(256), 0, as_stream(stream), feat, templ, loss_acc,
                     g_acc, B, Bt, HW, HWt, C, weight, relu_mask);
  return check_launch("nfs_hist_loss");
}

}  // extern "C"
