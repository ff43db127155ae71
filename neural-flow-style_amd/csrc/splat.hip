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
// nsize, fits 48 KB the block accumulates into LDS (ds_add_f32: neighbouring particles share 18 of their 27 cells) and
// flushes each touched cell ONCE with a global atomic; otherwise (unsorted or very sparse particles: decided per block
// from a min / max reduction of the cell coordinates) it falls back to the per-cell global atomics above.  Same
// arithmetic per contribution.
constexpr int SPL_LDS = 12288;     // floats of LDS accumulators

template <int SPL_PB>              // particles per block
__global__ void __launch_bounds__(256) p2g_fwd_lds_kernel(SplatDev s, const float* __restrict__ p,
                                                          const float* __restrict__ attr, const float* __restrict__ pd,
                                                          float* __restrict__ grid, float* __restrict__ wsum, int N,
                                                          int C, int allow_lds) {
  extern __shared__ float acc[];
  __shared__ int red[6 * 4];
  const int t = threadIdx.x;
  const int64_t a0 = (int64_t)blockIdx.x * SPL_PB;
  Particle P[SPL_PB / 256];
  int lo[3] = {0x7fffffff, 0x7fffffff, 0x7fffffff}, hi[3] = {-1, -1, -1};
#pragma unroll
  for (int j = 0; j < SPL_PB / 256; ++j) {
    const int64_t a = a0 + t + 256 * j;
    P[j].valid = false;
    if (a < N) {
      P[j] = load_particle(s, p, a);
      if (P[j].valid)
#pragma unroll
        for (int k = 0; k < 3; ++k) { lo[k] = min(lo[k], P[j].idx[k]); hi[k] = max(hi[k], P[j].idx[k]); }
    }
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
    if (k < s.nd) {
      any = any && hi[k] >= lo[k];
      lo[k] = max(lo[k] - s.nsize, 0);
      hi[k] = min(hi[k] + s.nsize, s.res[k] - 1);
      ext[k] = hi[k] - lo[k] + 1;
      vol *= ext[k] > 0 ? ext[k] : 1;
    } else {
      lo[k] = 0; hi[k] = 0;
    }
  }
  const int nch = s.mode == 0 ? 1 : (s.mode == 2 ? C + 1 : C);
  const bool use_lds = allow_lds && any && vol * nch <= SPL_LDS;
  const int nvol = (int)vol;
  if (use_lds) {
    for (int i = t; i < nvol * nch; i += 256) acc[i] = 0.f;
    __syncthreads();
  }
  const int span = 2 * s.nsize + 1;
  const int total = s.nd == 2 ? span * span : span * span * span;
#pragma unroll
  for (int j = 0; j < SPL_PB / 256; ++j) {
    if (!P[j].valid) continue;
    const int64_t a = a0 + t + 256 * j;
    float coef = 1.f;
    if (s.mode == 0) coef = s.mass;
    if (s.mode == 1) coef = s.mass / (pd ? pd[a] : s.rest_density);
    for (int o = 0; o < total; ++o) {
      int n[3] = {0, 0, 0};
      if (s.nd == 2) { n[0] = o / span - s.nsize; n[1] = o % span - s.nsize; }
      else { n[0] = o / (span * span) - s.nsize; n[1] = (o / span) % span - s.nsize; n[2] = o % span - s.nsize; }
      float d2 = 0.f;
      int c[3] = {0, 0, 0};
      for (int k = 0; k < s.nd; ++k) {
        const float rr = P[j].r[k] - (float)n[k] * s.cell;
        d2 += rr * rr;
        c[k] = P[j].idx[k] + n[k];
      }
      const float w = cubic_w(sqrtf(d2) / s.h, s.sigma);
      if (w == 0.f) continue;
      if (use_lds) {
        bool in = true;
        for (int k = 0; k < s.nd; ++k) in = in && c[k] >= lo[k] && c[k] <= hi[k];   // (the box is clipped to the grid)
        if (!in) continue;
        float* dst = acc + (((c[0] - lo[0]) * ext[1] + (c[1] - lo[1])) * ext[2] + (c[2] - lo[2])) * nch;
        if (s.mode == 0) {
          atomicAdd(dst, coef * w);
        } else {
          for (int ch = 0; ch < C; ++ch) atomicAdd(dst + ch, coef * w * attr[a * C + ch]);
          if (s.mode == 2) atomicAdd(dst + C, w);
        }
        continue;
      }
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
  if (!use_lds) return;
  __syncthreads();
  for (int i = t; i < nvol; i += 256) {
    int c[3];
    c[2] = lo[2] + i % ext[2];
    const int r = i / ext[2];
    c[1] = lo[1] + r % ext[1];
    c[0] = lo[0] + r / ext[1];
    const float* src = acc + (int64_t)i * nch;
    bool nz = false;
    for (int ch = 0; ch < nch; ++ch) nz = nz || src[ch] != 0.f;
    if (!nz) continue;
    const int64_t cell = cell_index(s, c);
    if (s.mode == 0) {
      atomicAdd(grid + cell, src[0]);
    } else {
      for (int ch = 0; ch < C; ++ch)
        if (src[ch] != 0.f) atomicAdd(grid + cell * C + ch, src[ch]);
      if (s.mode == 2 && src[C] != 0.f) atomicAdd(wsum + cell, src[C]);
    }
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
  hipLaunchKernelGGL(p2g_fwd_lds_kernel<256>, dim3(blocks_for(N, 256)), dim3(256), SPL_LDS * sizeof(float),
                     as_stream(stream), s, p, attr, pd, grid, wsum, N, C, allow_lds);
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
  hipLaunchKernelGGL(p2g_bwd_kernel, dim3(blocks_for(N, 256)), dim3(256), 0, as_stream(stream), s, p, attr, pd, g_grid,
                     g_wsum, g_p, g_attr, g_pd, N, C);
  return check_launch("nfs_p2g_bwd");
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
