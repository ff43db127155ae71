// A8: SPH particle -> grid splat (transform.py:1233-1267 W, 1310-1453 p2g, 1577-1704
// p2g_wavg) and its adjoint.  The reference issues (2*nsize+1)^d separate ScatterNd ops
// with N x (d+1) index tensors each; here one thread owns a particle, evaluates the whole
// neighbourhood in registers and scatters with float atomics (forward) / gathers
// (backward, no atomics).  HBM-bound: N*(4*nd+4*C) B of particle data + the touched cells.
#include "common.h"

#include <stdlib.h>

namespace nfs {

struct SplatDev {
  int nd, mode, nsize, clip;
  int res[3];
  float dom[3];
  float cell, h, sigma, mass, rest_density;
};

__device__ __forceinline__ float cubic_w(float q, float sigma) {
  if (q > 1.f) return 0.f;
  if (q <= 0.5f) return sigma * (6.f * (q * q * q - q * q) + 1.f);
  const float t = 1.f - q;
  return sigma * 2.f * t * t * t;
}
__device__ __forceinline__ float cubic_dw(float q, float sigma) {
  if (q > 1.f) return 0.f;
  if (q <= 0.5f) return sigma * 6.f * (3.f * q * q - 2.f * q);
  const float t = 1.f - q;
  return -sigma * 6.f * t * t;
}

struct Particle {
  bool valid;
  int idx[3];
  float r[3];
  bool grad_ok[3];  // clip mode: gradient passes the clamp
};

__device__ __forceinline__ Particle load_particle(const SplatDev& s, const float* __restrict__ p, int64_t a) {
  Particle P;
  P.valid = true;
#pragma unroll
  for (int k = 0; k < 3; ++k) { P.idx[k] = 0; P.r[k] = 0.f; P.grad_ok[k] = true; }
  for (int k = 0; k < s.nd; ++k) {
    float v = p[a * s.nd + k] * s.dom[k];
    if (s.clip) {
      const float hi = s.dom[k] - 1e-6f;
      P.grad_ok[k] = (v >= 0.f) && (v <= hi);
      v = fminf(fmaxf(v, 0.f), hi);
    } else if (!(v >= 0.f && v < s.dom[k])) {
      P.valid = false;
    }
    const float fl = floorf(v / s.cell);
    P.idx[k] = (int)fl;
    P.r[k] = v - (fl + 0.5f) * s.cell;
  }
  return P;
}

// linear cell index with the H flip of the reference (axis 0 in 2-D, axis 1 in 3-D), or -1
__device__ __forceinline__ int64_t cell_index(const SplatDev& s, const int* c) {
  int64_t lin = 0;
  const int hax = s.nd == 2 ? 0 : 1;
  for (int k = 0; k < s.nd; ++k) {
    if (c[k] < 0 || c[k] >= s.res[k]) return -1;
    const int ck = (k == hax) ? s.res[k] - 1 - c[k] : c[k];
    lin = lin * s.res[k] + ck;
  }
  return lin;
}


// The (2 nsize + 1)^nd neighbourhood of a particle with compile-time extents: the per-axis offsets r - n cell and their
// squares are formed once (3 x SP values instead of 3 per cell), the loops unroll, no integer division (the generic
// loops below spend ~120 instructions per cell on o / span, o % span with a run-time span).  f(i0, i1, i2, d2) with
// i_k = n_k + NS in [0, SP); rr[k][i_k] = r_k - n_k cell.
template <int ND, int NS>
struct Hood {
  static constexpr int SP = 2 * NS + 1;
  float rr[3][SP];
  __device__ __forceinline__ void init(const SplatDev& s, const Particle& P) {
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
      for (int i = 0; i < SP; ++i) rr[k][i] = k < ND ? P.r[k] - (float)(i - NS) * s.cell : 0.f;
  }
  template <class F>
  __device__ __forceinline__ void each(F&& f) const {
#pragma unroll
    for (int i0 = 0; i0 < SP; ++i0) {
      const float q0 = rr[0][i0] * rr[0][i0];
#pragma unroll
      for (int i1 = 0; i1 < SP; ++i1) {
        const float q01 = q0 + rr[1][i1] * rr[1][i1];
        if (ND == 2) {
          f(i0, i1, NS, q01);
        } else {
#pragma unroll
          for (int i2 = 0; i2 < SP; ++i2) f(i0, i1, i2, q01 + rr[2][i2] * rr[2][i2]);
        }
      }
    }
  }
};

__global__ void __launch_bounds__(256) p2g_fwd_kernel(SplatDev s, const float* __restrict__ p,
                                                      const float* __restrict__ attr, const float* __restrict__ pd,
                                                      float* __restrict__ grid, float* __restrict__ wsum, int N,
                                                      int C) {
  const int64_t a = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (a >= N) return;
  const Particle P = load_particle(s, p, a);
  if (!P.valid) return;
  const int span = 2 * s.nsize + 1;
  const int total = s.nd == 2 ? span * span : span * span * span;
  float coef = 1.f;
  if (s.mode == 0) coef = s.mass;
  if (s.mode == 1) coef = s.mass / (pd ? pd[a] : s.rest_density);
  for (int o = 0; o < total; ++o) {
    int n[3] = {0, 0, 0};
    if (s.nd == 2) { n[0] = o / span - s.nsize; n[1] = o % span - s.nsize; }
    else { n[0] = o / (span * span) - s.nsize; n[1] = (o / span) % span - s.nsize; n[2] = o % span - s.nsize; }
    float d2 = 0.f;
    int c[3];
    for (int k = 0; k < s.nd; ++k) {
      const float rr = P.r[k] - (float)n[k] * s.cell;
      d2 += rr * rr;
      c[k] = P.idx[k] + n[k];
    }
    const float w = cubic_w(sqrtf(d2) / s.h, s.sigma);
    if (w == 0.f) continue;
    const int64_t ci = cell_index(s, c);
    if (ci < 0) continue;
    if (s.mode == 0) {
      atomicAdd(grid + ci, coef * w);
    } else {
      for (int ch = 0; ch < C; ++ch) atomicAdd(grid + ci * C + ch, coef * w * attr[a * C + ch]);
      if (s.mode == 2) atomicAdd(wsum + ci, w);
    }
  }
}

// The same scatter with the block's cells privatised in LDS.  Particles arrive in the order of the grid (Styler.run
// sorts them once per sequence, by 8-cell bricks), so the own cells of a block's 256 particles sit in a small box
// [lo, hi] per axis -- and stay in one while a Lagrangian run moves them by a few cells.  When the box, widened by
// nsize, fits the LDS accumulators the block accumulates there (neighbouring particles share 18 of their 27 cells) and
// flushes each touched cell ONCE with a global atomic; otherwise (unsorted or very sparse particles: decided per block
// from a min / max reduction of the cell coordinates) it falls back to the per-cell global atomics above.  Same
// arithmetic per contribution.
//
// Round 4: the LDS accumulators are 64-bit FIXED POINT, as in the rotate adjoint (warp.hip; on gfx950 ds_add_f32
// sustains 0.33 lanes/clk/CU where ds_add_u64 runs at 9.4, tools/lds_atomic_bench.hip -- here that alone did not change
// the time, the index arithmetic was the bound; what it buys is sums that do not depend on the order of the adds).
// Every block scales by its own power of two, chosen from the largest
// |contribution| its particles can make (kernel maximum sigma x coefficient x largest |attribute|) times the
// particles per block, so that no cell sum can overflow 2^62; > 40 bits stay below the largest contribution (more than
// float accumulation keeps), and the integer sums do not depend on the order of the adds.
constexpr int SPL_LDS = 8192;      // 64-bit LDS accumulators (64 KB: two blocks per CU)

__device__ __forceinline__ unsigned long long splat_fix64(float c) {     // c = contribution * 2^(k - 32), |c| < 2^31
  unsigned hi, lo;
  asm("v_cvt_flr_i32_f32 %0, %1" : "=v"(hi) : "v"(c));
  const float fr = __builtin_amdgcn_fractf(c) * 4294967296.f;
  asm("v_cvt_u32_f32 %0, %1" : "=v"(lo) : "v"(fr));
  return ((unsigned long long)hi << 32) | lo;
}

// One particle per thread, 256 per block.  A block whose box does not fit (15 % of the blocks of the 5e5-particle set:
// a block that straddles the end of a brick row, or sparse bricks far apart) sends its contributions to the grid with
// global atomics.  Measured in round 4 and removed (tools/splat_phase_bench.py on an ablation build): peeling such a
// block into groups that fit (first remaining particle's brick row, up to four bricks along the last axis; zero,
// accumulate, flush, repeat) -- 171 us against 116: every extra pass costs its own box reduction, LDS clear and flush
// (the flush runs at the global-atomic rate, ~85 G/s), more than the 29 us the scattered atomics of those blocks take.
// Where the 116 us go: block skeleton (reductions, LDS clear, barriers) ~40, accumulation ~45 (neighbouring lanes hold
// particles of the same cell: same-address LDS atomics serialise), flush ~35, fallback blocks ~29.
template <int ND, int NS>          // (ND, NS) = (nd, nsize) known at compile time, or ND = 0: generic loops
__global__ void __launch_bounds__(256) p2g_fwd_lds_kernel(SplatDev s, const float* __restrict__ p,
                                                          const float* __restrict__ attr, const float* __restrict__ pd,
                                                          float* __restrict__ grid, float* __restrict__ wsum, int N,
                                                          int C, int allow_lds) {
  extern __shared__ unsigned long long acc[];
  __shared__ int red[8 * 4];
  const int t = threadIdx.x;
  const int64_t a = (int64_t)blockIdx.x * 256 + t;
  Particle P;
  P.valid = false;
  float coef = 1.f, at[4] = {0.f, 0.f, 0.f, 0.f}, cmax = 0.f;
  if (a < N) {
    P = load_particle(s, p, a);
    if (P.valid) {
      if (s.mode == 0) coef = s.mass;
      if (s.mode == 1) coef = s.mass / (pd ? pd[a] : s.rest_density);
      float am = 1.f;
      if (s.mode != 0) {
        am = s.mode == 2 ? 1.f : 0.f;                   // (mode 2 also accumulates the bare weights)
        for (int ch = 0; ch < C; ++ch) { at[ch] = attr[a * C + ch]; am = fmaxf(am, fabsf(at[ch])); }
      }
      cmax = fabsf(coef) * am;                          // largest |coefficient x attribute| of this particle
    }
  }
  const int nch = s.mode == 0 ? 1 : (s.mode == 2 ? C + 1 : C);
  const float h2 = s.h * s.h, inv_h = 1.f / s.h;
  // one contribution of this thread's particle to cell c / LDS slot dst
  auto weight = [&](float d2) {
    // q = |r| / h as d2 * rsq(d2) * (1 / h): one v_rsq_f32 instead of a correctly rounded square root and a division
    // (~20 instructions per cell; 1-2 ulp in q, far inside the parity tolerance)
    return cubic_w(d2 > 0.f ? d2 * __frsqrt_rn(d2) * inv_h : 0.f, s.sigma);
  };
  auto to_global = [&](const int* c, float w) {
    const int64_t ci = cell_index(s, c);
    if (ci < 0) return;
    if (s.mode == 0) {
      atomicAdd(grid + ci, coef * w);
    } else {
      for (int ch = 0; ch < C; ++ch) atomicAdd(grid + ci * C + ch, coef * w * at[ch]);
      if (s.mode == 2) atomicAdd(wsum + ci, w);
    }
  };
  // the whole neighbourhood of the particle through global atomics
  auto all_global = [&]() {
    if (ND > 0) {
      constexpr int NDc = ND > 0 ? ND : 3, NSc = ND > 0 ? NS : 1;
      Hood<NDc, NSc> hd;
      hd.init(s, P);
      hd.each([&](int i0, int i1, int i2, float d2) {
        if (d2 > h2) return;
        const float w = weight(d2);
        if (w == 0.f) return;
        const int c[3] = {P.idx[0] + i0 - NSc, P.idx[1] + i1 - NSc, NDc > 2 ? P.idx[2] + i2 - NSc : 0};
        to_global(c, w);
      });
    } else {
      const int span = 2 * s.nsize + 1;
      const int total = s.nd == 2 ? span * span : span * span * span;
      for (int o = 0; o < total; ++o) {
        int n[3] = {0, 0, 0};
        if (s.nd == 2) { n[0] = o / span - s.nsize; n[1] = o % span - s.nsize; }
        else { n[0] = o / (span * span) - s.nsize; n[1] = (o / span) % span - s.nsize; n[2] = o % span - s.nsize; }
        float d2 = 0.f;
        int c[3] = {0, 0, 0};
        for (int k = 0; k < s.nd; ++k) {
          const float rr = P.r[k] - (float)n[k] * s.cell;
          d2 += rr * rr;
          c[k] = P.idx[k] + n[k];
        }
        const float w = cubic_w(sqrtf(d2) / s.h, s.sigma);
        if (w != 0.f) to_global(c, w);
      }
    }
  };
  // block-wide maximum of cmax -> fixed-point scale 2^kexp: |any cell sum| <= 256 sigma cmax < 2^ebound stays below 2^62
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) cmax = fmaxf(cmax, __shfl_xor(cmax, o, 64));
  if ((t & 63) == 0) red[7 * 4 + (t >> 6)] = __float_as_int(cmax);
  __syncthreads();
  cmax = fmaxf(fmaxf(__int_as_float(red[28]), __int_as_float(red[29])), fmaxf(__int_as_float(red[30]), __int_as_float(red[31])));
  const float bound = 256.f * s.sigma * cmax;
  const bool scalable = bound > 0.f && bound < 3.0e38f;       // (inf / nan attributes: the float path reproduces them)
  int kexp = 0;
  float fs = 1.f, fs2 = 1.f;
  if (scalable) {
    int ebound;
    frexpf(bound, &ebound);
    kexp = 62 - ebound;
    const int ks = kexp - 32, k1 = min(max(ks, -120), 120);
    fs = ldexpf(1.f, k1);
    fs2 = ldexpf(1.f, ks - k1);                               // (2^ks may exceed the float range: two factors)
  }
  if (!(allow_lds && scalable && ND > 0)) {                   // generic neighbourhoods / unscalable values: global atomics
    if (P.valid) all_global();
    return;
  }
  constexpr int NDc = ND > 0 ? ND : 3, NSc = ND > 0 ? NS : 1;
  int lo[3] = {0x7fffffff, 0x7fffffff, 0x7fffffff}, hi[3] = {-1, -1, -1};
  if (P.valid)
#pragma unroll
    for (int k = 0; k < 3; ++k) { lo[k] = P.idx[k]; hi[k] = P.idx[k]; }
#pragma unroll
  for (int k = 0; k < 3; ++k) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      lo[k] = min(lo[k], __shfl_xor(lo[k], o, 64));
      hi[k] = max(hi[k], __shfl_xor(hi[k], o, 64));
    }
    if ((t & 63) == 0) { red[(2 * k) * 4 + (t >> 6)] = lo[k]; red[(2 * k + 1) * 4 + (t >> 6)] = hi[k]; }
  }
  __syncthreads();
  int ext[3] = {1, 1, 1};
  bool any = true;
  int64_t vol = 1;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const int* r0 = red + (2 * k) * 4;
    const int* r1 = red + (2 * k + 1) * 4;
    lo[k] = min(min(r0[0], r0[1]), min(r0[2], r0[3]));
    hi[k] = max(max(r1[0], r1[1]), max(r1[2], r1[3]));
    if (k < NDc) {
      any = any && hi[k] >= lo[k];
      lo[k] = max(lo[k] - NSc, 0);
      hi[k] = min(hi[k] + NSc, s.res[k] - 1);
      ext[k] = hi[k] - lo[k] + 1;
      vol *= ext[k] > 0 ? ext[k] : 1;
    } else {
      lo[k] = 0; hi[k] = 0;
    }
  }
  if (!any) return;                                           // no live particle in the block
  if (vol * nch > SPL_LDS) {
    if (P.valid) all_global();
    return;
  }
  const int nvol = (int)vol;
  for (int i = t; i < nvol * nch; i += 256) acc[i] = 0ull;
  __syncthreads();
  if (P.valid) {
    Hood<NDc, NSc> hd;
    hd.init(s, P);
    constexpr int SP = 2 * NSc + 1;
    int off[3][SP];                  // address term of the cell along each axis, or -1 outside the grid
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
      for (int i = 0; i < SP; ++i) {
        const int c = P.idx[k] + i - NSc;
        const bool in = k < NDc && c >= lo[k] && c <= hi[k];        // (the box is clipped to the grid)
        const int stride = k == 0 ? ext[1] * ext[2] : (k == 1 ? ext[2] : 1);
        off[k][i] = in ? (c - lo[k]) * stride : (k < NDc ? -1 : 0);
      }
    hd.each([&](int i0, int i1, int i2, float d2) {
      if (d2 > h2) return;
      if ((off[0][i0] | off[1][i1] | off[2][i2]) < 0) return;
      const float w = weight(d2);
      if (w == 0.f) return;
      unsigned long long* dst = acc + (off[0][i0] + off[1][i1] + off[2][i2]) * nch;
      if (s.mode == 0) {
        atomicAdd(dst, splat_fix64(coef * w * fs * fs2));
      } else {
        for (int ch = 0; ch < C; ++ch) atomicAdd(dst + ch, splat_fix64(coef * w * at[ch] * fs * fs2));
        if (s.mode == 2) atomicAdd(dst + C, splat_fix64(w * fs * fs2));
      }
    });
  }
  __syncthreads();
  for (int i = t; i < nvol; i += 256) {
    int c[3];
    c[2] = lo[2] + i % ext[2];
    const int r = i / ext[2];
    c[1] = lo[1] + r % ext[1];
    c[0] = lo[0] + r / ext[1];
    const unsigned long long* src = acc + (int64_t)i * nch;
    bool nz = false;
    for (int ch = 0; ch < nch; ++ch) nz = nz || src[ch] != 0ull;
    if (!nz) continue;
    const int64_t cell = cell_index(s, c);
#define NFS_SPL_F(q_) ((float)ldexp((double)(long long)(q_), -kexp))
    if (s.mode == 0) {
      atomicAdd(grid + cell, NFS_SPL_F(src[0]));
    } else {
      for (int ch = 0; ch < C; ++ch)
        if (src[ch] != 0ull) atomicAdd(grid + cell * C + ch, NFS_SPL_F(src[ch]));
      if (s.mode == 2 && src[C] != 0ull) atomicAdd(wsum + cell, NFS_SPL_F(src[C]));
    }
#undef NFS_SPL_F
  }
}

__global__ void __launch_bounds__(256) p2g_bwd_kernel(SplatDev s, const float* __restrict__ p,
                                                      const float* __restrict__ attr, const float* __restrict__ pd,
                                                      const float* __restrict__ g_grid,
                                                      const float* __restrict__ g_wsum, float* __restrict__ g_p,
                                                      float* __restrict__ g_attr, float* __restrict__ g_pd, int N,
                                                      int C) {
  const int64_t a = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (a >= N) return;
  const Particle P = load_particle(s, p, a);
  float gp[3] = {0.f, 0.f, 0.f};
  float ga[4] = {0.f, 0.f, 0.f, 0.f};  // C <= 4
  float gpd = 0.f;
  if (P.valid) {
    const int span = 2 * s.nsize + 1;
    const int total = s.nd == 2 ? span * span : span * span * span;
    const float pdv = (s.mode == 1) ? (pd ? pd[a] : s.rest_density) : 1.f;
    float coef = 1.f;
    if (s.mode == 0) coef = s.mass;
    if (s.mode == 1) coef = s.mass / pdv;
    for (int o = 0; o < total; ++o) {
      int n[3] = {0, 0, 0};
      if (s.nd == 2) { n[0] = o / span - s.nsize; n[1] = o % span - s.nsize; }
      else { n[0] = o / (span * span) - s.nsize; n[1] = (o / span) % span - s.nsize; n[2] = o % span - s.nsize; }
      float d2 = 0.f, rr[3] = {0.f, 0.f, 0.f};
      int c[3];
      for (int k = 0; k < s.nd; ++k) {
        rr[k] = P.r[k] - (float)n[k] * s.cell;
        d2 += rr[k] * rr[k];
        c[k] = P.idx[k] + n[k];
      }
      const float dist = sqrtf(d2);
      const float q = dist / s.h;
      if (q > 1.f) continue;
      const int64_t ci = cell_index(s, c);
      if (ci < 0) continue;
      const float w = cubic_w(q, s.sigma);
      // dL/dW for this (particle, cell)
      float gw;
      if (s.mode == 0) {
        gw = coef * g_grid[ci];
      } else {
        float dot = 0.f;
        for (int ch = 0; ch < C; ++ch) {
          const float g = g_grid[ci * C + ch];
          dot += attr[a * C + ch] * g;
          ga[ch] += coef * w * g;
        }
        gw = coef * dot;
        if (s.mode == 1) gpd -= coef * w * dot / pdv;
        if (s.mode == 2) gw += g_wsum[ci];
      }
      if (g_p && dist > 0.f) {  // safe sqrt: zero gradient at the cell centre
        const float f = gw * cubic_dw(q, s.sigma) / (dist * s.h);
        for (int k = 0; k < s.nd; ++k) gp[k] += f * rr[k];
      }
    }
  }
  if (g_p)
    for (int k = 0; k < s.nd; ++k) g_p[a * s.nd + k] = P.grad_ok[k] ? gp[k] * s.dom[k] : 0.f;
  if (g_attr)
    for (int ch = 0; ch < C; ++ch) g_attr[a * C + ch] = ga[ch];
  if (g_pd) g_pd[a] = gpd;
}

// The adjoint with compile-time neighbourhoods and the block's box of the grid gradient staged in LDS: the 256 particles
// of a block (grid order) read the same ~1000 cells 27 times between them -- one coalesced pass brings them in, the
// gathers are LDS reads.  Same per-cell arithmetic and the same order of the sums as p2g_bwd_kernel (cells in the
// order of the generic loop: axis 0 outermost), so the two agree bit for bit; a block whose box does not fit falls
// back to global gathers.
constexpr int SPB_LDS = 12288;     // floats of staged gradient (48 KB)

// WAVG: mode 2 with the adjoint of nfs_p2g_wavg_finish folded in -- g_grid is then dL/d(out) of the finished average,
// fin.xsum / fin.wsum the raw accumulators, and the gradients wrt the accumulators (g / w and -sum g x / w^2 where
// w > eps, g and 0 elsewhere: wavg_finish_bwd_kernel's lines) are formed per cell while the box is staged, instead of
// in a 5-array streaming pass over the whole grid (240 MB at 200 x 300 x 200).
struct WavgFinish { const float* xsum; const float* wsum; float eps; };

template <int ND, int NS, bool WAVG>
__global__ void __launch_bounds__(256) p2g_bwd_box_kernel(SplatDev s, const float* __restrict__ p,
                                                          const float* __restrict__ attr, const float* __restrict__ pd,
                                                          const float* __restrict__ g_grid,
                                                          const float* __restrict__ g_wsum, float* __restrict__ g_p,
                                                          float* __restrict__ g_attr, float* __restrict__ g_pd, int N,
                                                          int C, WavgFinish fin) {
  extern __shared__ float gbox[];
  __shared__ int red[6 * 4];
  const int t = threadIdx.x;
  const int64_t a = (int64_t)blockIdx.x * 256 + t;
  Particle P;
  P.valid = false;
  int lo[3] = {0x7fffffff, 0x7fffffff, 0x7fffffff}, hi[3] = {-1, -1, -1};
  if (a < N) {
    P = load_particle(s, p, a);
    if (P.valid)
#pragma unroll
      for (int k = 0; k < 3; ++k) { lo[k] = P.idx[k]; hi[k] = P.idx[k]; }
  }
#pragma unroll
  for (int k = 0; k < 3; ++k) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      lo[k] = min(lo[k], __shfl_xor(lo[k], o, 64));
      hi[k] = max(hi[k], __shfl_xor(hi[k], o, 64));
    }
    if ((t & 63) == 0) { red[(2 * k) * 4 + (t >> 6)] = lo[k]; red[(2 * k + 1) * 4 + (t >> 6)] = hi[k]; }
  }
  __syncthreads();
  int ext[3] = {1, 1, 1};
  bool any = true;
  int64_t vol = 1;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const int* r0 = red + (2 * k) * 4;
    const int* r1 = red + (2 * k + 1) * 4;
    lo[k] = min(min(r0[0], r0[1]), min(r0[2], r0[3]));
    hi[k] = max(max(r1[0], r1[1]), max(r1[2], r1[3]));
    if (k < ND) {
      any = any && hi[k] >= lo[k];
      lo[k] = max(lo[k] - NS, 0);
      hi[k] = min(hi[k] + NS, s.res[k] - 1);
      ext[k] = hi[k] - lo[k] + 1;
      vol *= ext[k] > 0 ? ext[k] : 1;
    } else {
      lo[k] = 0; hi[k] = 0;
    }
  }
  if (!any) {                                  // no live particle in the block: zero gradients
    if (a < N) {
      if (g_p) for (int k = 0; k < ND; ++k) g_p[a * ND + k] = 0.f;
      if (g_attr) for (int ch = 0; ch < C; ++ch) g_attr[a * C + ch] = 0.f;
      if (g_pd) g_pd[a] = 0.f;
    }
    return;
  }
  const int nch = s.mode == 2 ? C + 1 : C;     // (mode 0: C == 1)
  const bool staged = vol * nch <= SPB_LDS;
  const int nvol = (int)vol;
  if (staged) {
    for (int i = t; i < nvol; i += 256) {
      int c[3];
      c[2] = lo[2] + i % ext[2];
      const int r = i / ext[2];
      c[1] = lo[1] + r % ext[1];
      c[0] = lo[0] + r / ext[1];
      const int64_t cell = cell_index(s, c);
      if (WAVG) {
        const float w = fin.wsum[cell];
        float gw = 0.f;
        for (int ch = 0; ch < C; ++ch) {
          const float g = g_grid[cell * C + ch];
          if (w > fin.eps) {
            gbox[i * nch + ch] = g / w;
            gw -= g * fin.xsum[cell * C + ch] / (w * w);
          } else {
            gbox[i * nch + ch] = g;
          }
        }
        gbox[i * nch + C] = gw;
      } else {
        for (int ch = 0; ch < C; ++ch) gbox[i * nch + ch] = g_grid[cell * C + ch];
        if (s.mode == 2) gbox[i * nch + C] = g_wsum[cell];
      }
    }
    __syncthreads();
  }
  if (a >= N) return;
  float gp[3] = {0.f, 0.f, 0.f};
  float ga[4] = {0.f, 0.f, 0.f, 0.f};  // C <= 4
  float gpd = 0.f;
  if (P.valid) {
    const float pdv = (s.mode == 1) ? (pd ? pd[a] : s.rest_density) : 1.f;
    float coef = 1.f;
    if (s.mode == 0) coef = s.mass;
    if (s.mode == 1) coef = s.mass / pdv;
    float at[4] = {0.f, 0.f, 0.f, 0.f};
    if (s.mode != 0)
      for (int ch = 0; ch < C; ++ch) at[ch] = attr[a * C + ch];
    Hood<ND, NS> hd;
    hd.init(s, P);
    constexpr int SP = 2 * NS + 1;
    int off[3][SP];
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
      for (int i = 0; i < SP; ++i) {
        const int c = P.idx[k] + i - NS;
        const bool in = k < ND && c >= lo[k] && c <= hi[k];
        const int stride = k == 0 ? ext[1] * ext[2] : (k == 1 ? ext[2] : 1);
        off[k][i] = in ? (c - lo[k]) * stride : (k < ND ? -1 : 0);
      }
    const float h2 = s.h * s.h, inv_h = 1.f / s.h;
    hd.each([&](int i0, int i1, int i2, float d2) {
      if (d2 > h2) return;
      if ((off[0][i0] | off[1][i1] | off[2][i2]) < 0) return;
      const float inv_d = d2 > 0.f ? __frsqrt_rn(d2) : 0.f;       // (as in the forward kernel: rsq instead of sqrt + two divisions)
      const float dist = d2 * inv_d;
      const float q = dist * inv_h;
      if (q > 1.f) return;
      const int li = off[0][i0] + off[1][i1] + off[2][i2];
      float gv[5];
      if (staged) {
        for (int ch = 0; ch < nch; ++ch) gv[ch] = gbox[li * nch + ch];
      } else {
        const int c[3] = {P.idx[0] + i0 - NS, P.idx[1] + i1 - NS, P.idx[2] + i2 - NS};
        const int64_t ci = cell_index(s, c);
        if (WAVG) {
          const float ws_ = fin.wsum[ci];
          float gws = 0.f;
          for (int ch = 0; ch < C; ++ch) {
            const float g = g_grid[ci * C + ch];
            if (ws_ > fin.eps) {
              gv[ch] = g / ws_;
              gws -= g * fin.xsum[ci * C + ch] / (ws_ * ws_);
            } else {
              gv[ch] = g;
            }
          }
          gv[C] = gws;
        } else {
          for (int ch = 0; ch < C; ++ch) gv[ch] = g_grid[ci * C + ch];
          if (s.mode == 2) gv[C] = g_wsum[ci];
        }
      }
      const float w = cubic_w(q, s.sigma);
      float gw;
      if (s.mode == 0) {
        gw = coef * gv[0];
      } else {
        float dot = 0.f;
        for (int ch = 0; ch < C; ++ch) {
          dot += at[ch] * gv[ch];
          ga[ch] += coef * w * gv[ch];
        }
        gw = coef * dot;
        if (s.mode == 1) gpd -= coef * w * dot / pdv;
        if (s.mode == 2) gw += gv[C];
      }
      if (g_p && dist > 0.f) {  // safe sqrt: zero gradient at the cell centre
        const float f = gw * cubic_dw(q, s.sigma) * (inv_d * inv_h);
        gp[0] += f * hd.rr[0][i0];
        if (ND > 1) gp[1] += f * hd.rr[1][i1];
        if (ND > 2) gp[2] += f * hd.rr[2][i2];
      }
    });
  }
  if (g_p)
    for (int k = 0; k < ND; ++k) g_p[a * ND + k] = P.grad_ok[k] ? gp[k] * s.dom[k] : 0.f;
  if (g_attr)
    for (int ch = 0; ch < C; ++ch) g_attr[a * C + ch] = ga[ch];
  if (g_pd) g_pd[a] = gpd;
}

__global__ void __launch_bounds__(256) wavg_finish_kernel(const float* __restrict__ xsum,
                                                          const float* __restrict__ wsum, float* __restrict__ out,
                                                          int64_t n, int C, float eps) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float w = wsum[i];
  for (int c = 0; c < C; ++c) out[i * C + c] = w > eps ? xsum[i * C + c] / w : xsum[i * C + c];
}

__global__ void __launch_bounds__(256) wavg_finish_bwd_kernel(const float* __restrict__ xsum,
                                                              const float* __restrict__ wsum,
                                                              const float* __restrict__ g_out,
                                                              float* __restrict__ g_xsum, float* __restrict__ g_wsum,
                                                              int64_t n, int C, float eps) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float w = wsum[i];
  float gw = 0.f;
  for (int c = 0; c < C; ++c) {
    const float g = g_out[i * C + c];
    if (w > eps) {
      g_xsum[i * C + c] = g / w;
      gw -= g * xsum[i * C + c] / (w * w);
    } else {
      g_xsum[i * C + c] = g;
    }
  }
  g_wsum[i] = gw;
}

static int make_dev_cfg(const nfs_splat_cfg* c, int C, SplatDev& s) {
  NFS_REQUIRE(c, "p2g: null config");
  NFS_REQUIRE(c->nd == 2 || c->nd == 3, "p2g: nd must be 2 or 3");
  NFS_REQUIRE(c->mode >= 0 && c->mode <= 2, "p2g: mode must be 0..2");
  NFS_REQUIRE(c->nsize >= 0 && c->nsize <= 8, "p2g: nsize out of range");
  NFS_REQUIRE(C >= 1 && C <= 4, "p2g: 1 <= C <= 4");
  s.nd = c->nd; s.mode = c->mode; s.nsize = c->nsize; s.clip = c->clip;
  for (int k = 0; k < 3; ++k) { s.res[k] = c->res[k]; s.dom[k] = c->domain[k]; }
  for (int k = 0; k < c->nd; ++k) NFS_REQUIRE(c->res[k] > 0 && c->domain[k] > 0.f, "p2g: bad res/domain");
  s.cell = c->domain[0] / (float)c->res[0];  // cell_size[0] (transform.py:1328-1330)
  s.h = c->radius * c->support;
  const double pi = 3.14159265358979323846;
  s.sigma = (float)(c->nd == 3 ? 8.0 / pi / ((double)s.h * s.h * s.h) : 40.0 / 7.0 / pi / ((double)s.h * s.h));
  const double d2r = 2.0 * c->radius;
  s.mass = (float)(0.8 * (c->nd == 3 ? d2r * d2r * d2r : d2r * d2r) * c->rest_density);
  s.rest_density = c->rest_density;
  return NFS_OK;
}


// ---- grid -> particle sampling (transform.py:771-1231) ---------------------------------------
// g [X,Y,(Z),C] cell-centred, p [N,nd] in [0,1] (axis order = array order): x = p * n, base = floor(x - 0.5).
// linear: cells (base, base+1) clipped to [0,n-1], weight dx = x - (clipped base + 0.5) (so outside the centre
// lattice both cells coincide and the value is the border cell's); cubic: cells base-1..base+2 clipped,
// t = x - (clipped base + 0.5), Catmull-Rom weights of the reference's _hermite.  One thread per particle.
struct G2PArgs {
  const float* g;
  const float* p;
  float* out;
  int nd, n[3], C, cubic;
  int64_t N;
};

__device__ __forceinline__ void g2p_axis(float x, int n, int cubic, int* idx, float* w) {
  const float fb = floorf(x - 0.5f);
  const int b = (int)fminf(fmaxf(fb, -4.f), (float)n + 4.f);
  if (!cubic) {
    idx[0] = min(max(b, 0), n - 1);
    idx[1] = min(max(b + 1, 0), n - 1);
    const float dx = x - ((float)idx[0] + 0.5f);
    w[0] = 1.f - dx; w[1] = dx;
    return;
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) idx[k] = min(max(b - 1 + k, 0), n - 1);
  const float t = x - ((float)idx[1] + 0.5f);
  const float t2 = t * t, t3 = t2 * t;
  w[0] = -0.5f * t3 + t2 - 0.5f * t;
  w[1] = 1.5f * t3 - 2.5f * t2 + 1.f;
  w[2] = -1.5f * t3 + 2.f * t2 + 0.5f * t;
  w[3] = 0.5f * t3 - 0.5f * t2;
}

__global__ void __launch_bounds__(256) g2p_kernel(G2PArgs a) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= a.N) return;
  const int taps = a.cubic ? 4 : 2;
  int ix[3][4];
  float wx[3][4];
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    if (d < a.nd) {
      g2p_axis(a.p[i * a.nd + d] * (float)a.n[d], a.n[d], a.cubic, ix[d], wx[d]);
    } else {
      ix[d][0] = 0; wx[d][0] = 1.f;
    }
  }
  const int t2 = a.nd >= 3 ? taps : 1;
  for (int c0 = 0; c0 < a.C; c0 += 4) {
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    const int cn = min(4, a.C - c0);
    for (int ka = 0; ka < taps; ++ka)
      for (int kb = 0; kb < taps; ++kb) {
        const float wab = wx[0][ka] * wx[1][kb];
        const int64_t rowb = (int64_t)ix[0][ka] * a.n[1] + ix[1][kb];
        for (int kc = 0; kc < t2; ++kc) {
          const float w = wab * wx[2][kc];
          const float* gp = a.g + ((a.nd >= 3 ? rowb * a.n[2] + ix[2][kc] : rowb) * a.C + c0);
          for (int c = 0; c < cn; ++c) acc[c] += w * gp[c];
        }
      }
    for (int c = 0; c < cn; ++c) a.out[i * a.C + c0 + c] = acc[c];
  }
}


// p2g adjoint launch: compile-time neighbourhoods with the box staged in LDS where they exist, the generic gather otherwise;
// fin != nullptr (mode 2): the adjoint of the weighted-average finish folded in (the generic kernel has no such form:
// the caller falls back to the two-step path)
static bool launch_p2g_bwd(const SplatDev& s, const float* p, const float* attr, const float* pd, const float* g_grid,
                           const float* g_wsum, float* g_p, float* g_attr, float* g_pd, int N, int C,
                           const WavgFinish* fin, hipStream_t stream) {
  int64_t cells = 1;
  for (int k = 0; k < s.nd; ++k) cells *= s.res[k];
  static const int allow_box = [] { const char* e = getenv("NFS_SPLAT_LDS"); return e ? atoi(e) : 1; }();
  const bool box = allow_box && cells < ((int64_t)1 << 31);
  const WavgFinish f = fin ? *fin : WavgFinish{nullptr, nullptr, 0.f};
#define NFS_SPB_LAUNCH(ND_, NS_)                                                                                       \
  do {                                                                                                                 \
    if (fin)                                                                                                           \
      hipLaunchKernelGGL((p2g_bwd_box_kernel<ND_, NS_, true>), dim3(blocks_for(N, 256)), dim3(256),                    \
                         SPB_LDS * sizeof(float), stream, s, p, attr, pd, g_grid, g_wsum, g_p, g_attr, g_pd, N, C, f); \
    else                                                                                                               \
      hipLaunchKernelGGL((p2g_bwd_box_kernel<ND_, NS_, false>), dim3(blocks_for(N, 256)), dim3(256),                   \
                         SPB_LDS * sizeof(float), stream, s, p, attr, pd, g_grid, g_wsum, g_p, g_attr, g_pd, N, C, f); \
  } while (0)
  if (box && s.nd == 3 && s.nsize == 1) NFS_SPB_LAUNCH(3, 1);
  else if (box && s.nd == 3 && s.nsize == 2) NFS_SPB_LAUNCH(3, 2);
  else if (box && s.nd == 2 && s.nsize == 1) NFS_SPB_LAUNCH(2, 1);
  else if (box && s.nd == 2 && s.nsize == 2) NFS_SPB_LAUNCH(2, 2);
  else if (box && s.nd == 2 && s.nsize == 3) NFS_SPB_LAUNCH(2, 3);
  else if (box && s.nd == 2 && s.nsize == 4) NFS_SPB_LAUNCH(2, 4);
  else {
    if (fin) return false;
    hipLaunchKernelGGL(p2g_bwd_kernel, dim3(blocks_for(N, 256)), dim3(256), 0, stream, s, p, attr, pd, g_grid, g_wsum, g_p,
                       g_attr, g_pd, N, C);
  }
#undef NFS_SPB_LAUNCH
  return true;
}

}  // namespace nfs

using namespace nfs;

extern "C" {

int nfs_p2g_fwd(const float* p, const float* attr, const float* pd, float* grid, float* wsum, int N, int C,
                const nfs_splat_cfg* cfg_host, nfs_stream_t stream) {
  NFS_REQUIRE(p && grid && N > 0, "nfs_p2g_fwd: bad argument");
  SplatDev s;
  if (int e = make_dev_cfg(cfg_host, C, s)) return e;
  NFS_REQUIRE(s.mode == 0 || attr, "nfs_p2g_fwd: attr required for mode 1/2");
  NFS_REQUIRE(s.mode != 2 || wsum, "nfs_p2g_fwd: wsum required for mode 2");
  NFS_REQUIRE(s.mode != 0 || C == 1, "nfs_p2g_fwd: density mode has C=1");
  int64_t cells = 1;
  for (int k = 0; k < s.nd; ++k) cells *= s.res[k];
  if (cells >= ((int64_t)1 << 31)) {                 // (the LDS form keeps linear cell indices in 32 bits)
    hipLaunchKernelGGL(p2g_fwd_kernel, dim3(blocks_for(N, 256)), dim3(256), 0, as_stream(stream), s, p, attr, pd, grid,
                       wsum, N, C);
    return check_launch("nfs_p2g_fwd");
  }
  static const int allow_lds = [] { const char* e = getenv("NFS_SPLAT_LDS"); return e ? atoi(e) : 1; }();
  // 256 particles per block: measured 0.156 ms against 0.184 (512) and 0.247 (1024) on the 5e5-particle blob set
#define NFS_SPL_LAUNCH(ND_, NS_)                                                                                       \
  hipLaunchKernelGGL((p2g_fwd_lds_kernel<ND_, NS_>), dim3(blocks_for(N, 256)), dim3(256),                         \
                     SPL_LDS * sizeof(unsigned long long), as_stream(stream), s, p, attr, pd, grid, wsum, N, C, allow_lds)
  // the neighbourhoods the drivers use as compile-time instances (test_smokegun / chocolate: 3-D nsize 1;
  // test_dambreak2d: 2-D nsize 2..4), anything else through the generic loops
  if (s.nd == 3 && s.nsize == 1) NFS_SPL_LAUNCH(3, 1);
  else if (s.nd == 3 && s.nsize == 2) NFS_SPL_LAUNCH(3, 2);
  else if (s.nd == 2 && s.nsize == 1) NFS_SPL_LAUNCH(2, 1);
  else if (s.nd == 2 && s.nsize == 2) NFS_SPL_LAUNCH(2, 2);
  else if (s.nd == 2 && s.nsize == 3) NFS_SPL_LAUNCH(2, 3);
  else if (s.nd == 2 && s.nsize == 4) NFS_SPL_LAUNCH(2, 4);
  else NFS_SPL_LAUNCH(0, 0);
#undef NFS_SPL_LAUNCH
  return check_launch("nfs_p2g_fwd");
}

int nfs_p2g_bwd(const float* p, const float* attr, const float* pd, const float* g_grid, const float* g_wsum,
                float* g_p, float* g_attr, float* g_pd, int N, int C, const nfs_splat_cfg* cfg_host,
                nfs_stream_t stream) {
  NFS_REQUIRE(p && g_grid && N > 0, "nfs_p2g_bwd: bad argument");
  SplatDev s;
  if (int e = make_dev_cfg(cfg_host, C, s)) return e;
  NFS_REQUIRE(s.mode == 0 || attr, "nfs_p2g_bwd: attr required for mode 1/2");
  NFS_REQUIRE(s.mode != 2 || g_wsum, "nfs_p2g_bwd: g_wsum required for mode 2");
  NFS_REQUIRE(s.mode != 0 || (!g_attr && !g_pd), "nfs_p2g_bwd: density mode has no attr/pd gradient");
  (void)launch_p2g_bwd(s, p, attr, pd, g_grid, g_wsum, g_p, g_attr, g_pd, N, C, nullptr, as_stream(stream));
  return check_launch("nfs_p2g_bwd");
}

/* nfs_p2g_wavg_finish_bwd + nfs_p2g_bwd (mode 2) in one launch: g_out [cells,C] = dL/d(finished average); returns
 * NFS_EINVAL for neighbourhoods without a compile-time instance (the caller then takes the two-step path) */
int nfs_p2g_wavg_bwd(const float* p, const float* attr, const float* xsum, const float* wsum, const float* g_out,
                     float* g_p, float* g_attr, int N, int C, float eps, const nfs_splat_cfg* cfg_host,
                     nfs_stream_t stream) {
  NFS_REQUIRE(p && attr && xsum && wsum && g_out && N > 0, "nfs_p2g_wavg_bwd: bad argument");
  SplatDev s;
  if (int e = make_dev_cfg(cfg_host, C, s)) return e;
  NFS_REQUIRE(s.mode == 2, "nfs_p2g_wavg_bwd: mode 2 (weighted average) only");
  const WavgFinish fin{xsum, wsum, eps};
  NFS_REQUIRE(launch_p2g_bwd(s, p, attr, nullptr, g_out, nullptr, g_p, g_attr, nullptr, N, C, &fin, as_stream(stream)),
              "nfs_p2g_wavg_bwd: no compile-time instance for nd %d, nsize %d (use nfs_p2g_wavg_finish_bwd + nfs_p2g_bwd)",
              s.nd, s.nsize);
  return check_launch("nfs_p2g_wavg_bwd");
}

int nfs_p2g_wavg_finish(const float* xsum, const float* wsum, float* out, int64_t n, int C, float eps,
                        nfs_stream_t stream) {
  NFS_REQUIRE(xsum && wsum && out && n > 0 && C > 0, "nfs_p2g_wavg_finish: bad argument");
  hipLaunchKernelGGL(wavg_finish_kernel, dim3(blocks_for(n, 256)), dim3(256), 0, as_stream(stream), xsum, wsum, out, n,
                     C, eps);
  return check_launch("nfs_p2g_wavg_finish");
}

int nfs_p2g_wavg_finish_bwd(const float* xsum, const float* wsum, const float* g_out, float* g_xsum, float* g_wsum,
                            int64_t n, int C, float eps, nfs_stream_t stream) {
  NFS_REQUIRE(xsum && wsum && g_out && g_xsum && g_wsum && n > 0 && C > 0, "nfs_p2g_wavg_finish_bwd: bad argument");
  hipLaunchKernelGGL(wavg_finish_bwd_kernel, dim3(blocks_for(n, 256)), dim3(256), 0, as_stream(stream), xsum, wsum,
                     g_out, g_xsum, g_wsum, n, C, eps);
  return check_launch("nfs_p2g_wavg_finish_bwd");
}

int nfs_g2p_fwd(const float* g, const float* p, float* out, int nd, int X, int Y, int Z, int C, int64_t N, int cubic,
                nfs_stream_t stream) {
  NFS_REQUIRE(g && p && out, "nfs_g2p_fwd: null pointer");
  NFS_REQUIRE((nd == 2 || nd == 3) && X > 0 && Y > 0 && (nd == 2 || Z > 0) && C > 0 && N > 0,
              "nfs_g2p_fwd: bad dimension");
  G2PArgs a{g, p, out, nd, {X, Y, nd == 3 ? Z : 1}, C, cubic ? 1 : 0, N};
  hipLaunchKernelGGL(g2p_kernel, dim3(blocks_for(N, 256)), dim3(256), 0, as_stream(stream), a);
  return check_launch("nfs_g2p_fwd");
}

}  // extern "C"
