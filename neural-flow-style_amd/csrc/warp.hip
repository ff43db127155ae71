// A2/A3/A11: border-replicating trilinear resampling family (transform.py:238-269,
// 343-433, 557-569, 611-628).  One thread per output voxel, W fastest => coalesced
// stores and near-coalesced gathers; coordinates are generated in registers (the
// reference materialises 3 coordinate volumes + 8 index/weight/gather tensors).
// HBM-bound: algorithmic bytes = read volume once + write output once.
#include "common.h"
#include <stdlib.h>

namespace nfs {

enum CoordKind { COORD_EXPLICIT = 0, COORD_ROTATE = 1, COORD_ADVECT = 2 };

struct WarpArgs {
  const float* src;     // fwd: source volume(s); bwd: as fwd (may be null when unused)
  const float* coords;  // explicit [B,3,X,Y,Z] | rot [B,9] | vel [X,Y,Z,3]
  int B, X, Y, Z, C;
  int src_batched;      // 1: src has a batch dim (explicit), 0: one shared volume (rotate/advect)
};

template <int KIND>
__device__ __forceinline__ void coords_at(const WarpArgs& a, int b, int x, int y, int z, int64_t vox,
                                          float& cx, float& cy, float& cz) {
  if (KIND == COORD_EXPLICIT) {
    const int64_t n = (int64_t)a.X * a.Y * a.Z;
    const float* c = a.coords + (int64_t)b * 3 * n;
    cx = c[vox];
    cy = c[n + vox];
    cz = c[2 * n + vox];
  } else if (KIND == COORD_ROTATE) {
    const float* r = a.coords + b * 9;
    const float gx = lin_coord(x, a.X), gy = lin_coord(y, a.Y), gz = lin_coord(z, a.Z);
    cx = r[0] * gx + r[1] * gy + r[2] * gz;
    cy = r[3] * gx + r[4] * gy + r[5] * gz;
    cz = r[6] * gx + r[7] * gy + r[8] * gz;
  } else {
    const float* v = a.coords + vox * 3;
    cx = lin_coord(x, a.X) - v[0];
    cy = lin_coord(y, a.Y) - v[1];
    cz = lin_coord(z, a.Z) - v[2];
  }
}

template <int KIND>
__global__ void __launch_bounds__(256) warp_fwd_kernel(WarpArgs a, float* __restrict__ out) {
  const int64_t n = (int64_t)a.X * a.Y * a.Z;
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= n * a.B) return;
  const int b = (int)(gid / n);
  const int64_t vox = gid - (int64_t)b * n;
  const int z = (int)(vox % a.Z);
  const int y = (int)((vox / a.Z) % a.Y);
  const int x = (int)(vox / ((int64_t)a.Z * a.Y));
  float cx, cy, cz;
  coords_at<KIND>(a, b, x, y, z, vox, cx, cy, cz);
  Tri t; Axis ax, ay, az;
  tri_setup(cx, cy, cz, a.X, a.Y, a.Z, t, ax, ay, az);
  const float* src = a.src + (a.src_batched ? (int64_t)b * n * a.C : 0);
  float* o = out + gid * a.C;
  if (a.C == 1 && a.Z >= 2) {   // scalar volume: x-corner pairs from 8-byte loads
    o[0] = tri_sample_pairs(src, a.Y, a.Z, ax, ay, az);
    return;
  }
  for (int c = 0; c < a.C; ++c) {
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) s += t.w[k] * src[t.o[k] * a.C + c];
    o[c] = s;
  }
}

// adjoint: g_src += scatter(g_out) (float atomics), g_coord written (explicit: [B,3,X,Y,Z];
// advect: g_vel [X,Y,Z,3] = -g_coord).
template <int KIND>
__global__ void __launch_bounds__(256) warp_bwd_kernel(WarpArgs a, const float* __restrict__ g_out,
                                                       float* __restrict__ g_src, float* __restrict__ g_coord) {
  const int64_t n = (int64_t)a.X * a.Y * a.Z;
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= n * a.B) return;
  const int b = (int)(gid / n);
  const int64_t vox = gid - (int64_t)b * n;
  const int z = (int)(vox % a.Z);
  const int y = (int)((vox / a.Z) % a.Y);
  const int x = (int)(vox / ((int64_t)a.Z * a.Y));
  float cx, cy, cz;
  coords_at<KIND>(a, b, x, y, z, vox, cx, cy, cz);
  Tri t; Axis ax, ay, az;
  tri_setup(cx, cy, cz, a.X, a.Y, a.Z, t, ax, ay, az);
  const int64_t boff = a.src_batched ? (int64_t)b * n * a.C : 0;
  const float* go = g_out + gid * a.C;
  float gx = 0.f, gy = 0.f, gz = 0.f;
  const float wx[2] = {1.f - ax.w1, ax.w1}, wy[2] = {1.f - ay.w1, ay.w1}, wz[2] = {1.f - az.w1, az.w1};
  for (int c = 0; c < a.C; ++c) {
    const float g = go[c];
    if (g_src) {
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const float contrib = t.w[k] * g;
        if (contrib != 0.f) atomicAdd(g_src + boff + t.o[k] * a.C + c, contrib);
      }
    }
    if (g_coord) {
      const float* src = a.src + boff;
      float v[8];
      if (a.C == 1 && a.Z >= 2) {
        tri_gather_pairs(src, a.Y, a.Z, ax, ay, az, v);
      } else {
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = src[t.o[k] * a.C + c];
      }
      // k = a*4 + b*2 + c  (x,y,z bits)
      const float dx = wy[0] * wz[0] * (v[4] - v[0]) + wy[0] * wz[1] * (v[5] - v[1]) +
                       wy[1] * wz[0] * (v[6] - v[2]) + wy[1] * wz[1] * (v[7] - v[3]);
      const float dy = wx[0] * wz[0] * (v[2] - v[0]) + wx[0] * wz[1] * (v[3] - v[1]) +
                       wx[1] * wz[0] * (v[6] - v[4]) + wx[1] * wz[1] * (v[7] - v[5]);
      const float dz = wx[0] * wy[0] * (v[1] - v[0]) + wx[0] * wy[1] * (v[3] - v[2]) +
                       wx[1] * wy[0] * (v[5] - v[4]) + wx[1] * wy[1] * (v[7] - v[6]);
      gx += g * dx;
      gy += g * dy;
      gz += g * dz;
    }
  }
  if (g_coord) {
    gx *= (float)(a.X - 1) * 0.5f;
    gy *= (float)(a.Y - 1) * 0.5f;
    gz *= (float)(a.Z - 1) * 0.5f;
    if (KIND == COORD_ADVECT) {
      float* gv = g_coord + vox * 3;
      gv[0] = -gx; gv[1] = -gy; gv[2] = -gz;
    } else {
      float* gc = g_coord + (int64_t)b * 3 * n;
      gc[vox] = gx; gc[n + vox] = gy; gc[2 * n + vox] = gz;
    }
  }
}


// ---- advect, scalar field (the hot case) --------------------------------------------------------
// The generic kernel keeps one voxel per thread in flight and is bound by latency x occupancy
// (~1.9 TB/s).  Here a thread owns 4 W-consecutive voxels: the 12 velocity floats arrive as three
// float4 loads, the 16 x-corner pairs as 8-byte loads, all issued before use, 32-bit indexing.
__device__ __forceinline__ void advect1_stencil(const float* __restrict__ d, int D, int H, int W, int vox, float v0,
                                                float v1, float v2, Axis& az, Axis& ay, Axis& ax) {
  const int w = vox % W;
  const int hh = (vox / W) % H;
  const int z = vox / (W * H);
  az = axis_setup(lin_coord(z, D) - v0, D);
  ay = axis_setup(lin_coord(hh, H) - v1, H);
  ax = axis_setup(lin_coord(w, W) - v2, W);
}

__device__ __forceinline__ void gather_pairs32(const float* __restrict__ vol, int H, int W, const Axis& az,
                                               const Axis& ay, const Axis& ax, float* v) {
  const int xb = min(ax.i0, W - 2);
  const bool x0_lo = ax.i0 == xb, x1_lo = ax.i1 == xb;
  const int iz[2] = {az.i0, az.i1}, iy[2] = {ay.i0, ay.i1};
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const F2u p = *reinterpret_cast<const F2u*>(vol + ((iz[a] * H + iy[b]) * W + xb));
      v[a * 4 + b * 2] = x0_lo ? p.x : p.y;
      v[a * 4 + b * 2 + 1] = x1_lo ? p.x : p.y;
    }
}

template <bool BWD>
__global__ void __launch_bounds__(256) advect1_kernel(const float* __restrict__ d, const float* __restrict__ vel,
                                                      const float* __restrict__ g_out, float* __restrict__ out,
                                                      int D, int H, int W) {
  const int n = D * H * W;
  const int base = (blockIdx.x * blockDim.x + threadIdx.x) * 4;   // n % 4 == 0 (checked by the host)
  if (base >= n) return;
  const float4* v4 = reinterpret_cast<const float4*>(vel) + (size_t)(base / 4) * 3;
  const float4 va = v4[0], vb = v4[1], vc = v4[2];
  const float vv[12] = {va.x, va.y, va.z, va.w, vb.x, vb.y, vb.z, vb.w, vc.x, vc.y, vc.z, vc.w};
  float4 go = make_float4(0.f, 0.f, 0.f, 0.f);
  if (BWD) go = *reinterpret_cast<const float4*>(g_out + base);
  const float gg[4] = {go.x, go.y, go.z, go.w};
  Axis az[4], ay[4], ax[4];
  float cv[4][8];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    advect1_stencil(d, D, H, W, base + j, vv[3 * j], vv[3 * j + 1], vv[3 * j + 2], az[j], ay[j], ax[j]);
    gather_pairs32(d, H, W, az[j], ay[j], ax[j], cv[j]);
  }
  if (!BWD) {
    float r[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float wz[2] = {1.f - az[j].w1, az[j].w1}, wy[2] = {1.f - ay[j].w1, ay[j].w1},
                  wx[2] = {1.f - ax[j].w1, ax[j].w1};
      float sacc = 0.f;
#pragma unroll
      for (int k = 0; k < 8; ++k) sacc += wz[k >> 2] * wy[(k >> 1) & 1] * wx[k & 1] * cv[j][k];
      r[j] = sacc;
    }
    *reinterpret_cast<float4*>(out + base) = make_float4(r[0], r[1], r[2], r[3]);
  } else {
    float gv[12];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float* v = cv[j];
      const float wz[2] = {1.f - az[j].w1, az[j].w1}, wy[2] = {1.f - ay[j].w1, ay[j].w1},
                  wx[2] = {1.f - ax[j].w1, ax[j].w1};
      // k = a*4 + b*2 + c with (a,b,c) = (z,y,x) corner bits
      const float dz = wy[0] * wx[0] * (v[4] - v[0]) + wy[0] * wx[1] * (v[5] - v[1]) +
                       wy[1] * wx[0] * (v[6] - v[2]) + wy[1] * wx[1] * (v[7] - v[3]);
      const float dy = wz[0] * wx[0] * (v[2] - v[0]) + wz[0] * wx[1] * (v[3] - v[1]) +
                       wz[1] * wx[0] * (v[6] - v[4]) + wz[1] * wx[1] * (v[7] - v[5]);
      const float dx = wz[0] * wy[0] * (v[1] - v[0]) + wz[0] * wy[1] * (v[3] - v[2]) +
                       wz[1] * wy[0] * (v[5] - v[4]) + wz[1] * wy[1] * (v[7] - v[6]);
      gv[3 * j] = -gg[j] * dz * ((float)(D - 1) * 0.5f);
      gv[3 * j + 1] = -gg[j] * dy * ((float)(H - 1) * 0.5f);
      gv[3 * j + 2] = -gg[j] * dx * ((float)(W - 1) * 0.5f);
    }
    float4* o4 = reinterpret_cast<float4*>(out) + (size_t)(base / 4) * 3;
    o4[0] = make_float4(gv[0], gv[1], gv[2], gv[3]);
    o4[1] = make_float4(gv[4], gv[5], gv[6], gv[7]);
    o4[2] = make_float4(gv[8], gv[9], gv[10], gv[11]);
  }
}

// ---- output-stationary adjoint of rotate for C = 1 -------------------------------------------
// Global float atomics cap the scatter at ~90 G atomics/s (5.6 ms for 8 views of 200^3).  Here a
// block OWNS a TZ x TY x TX tile of g_d in LDS.  For every view it inverse-maps the tile's
// catchment box (tile +-1 cell; open-ended on volume faces, because out-of-range samples clamp
// onto them) through the affine sample map x = A o + c and visits every sample of g_out inside the
// integer bounding box (x fastest => coalesced reads), recomputes the forward stencil and
// accumulates the corners that fall inside the tile with LDS atomics.  Measured on gfx950
// (tools/lds_atomic_bench.hip): ds_add_f32 sustains only 0.33 lanes/clk/CU while ds_add_u64 runs
// at 9.4 -- so the tile accumulates in 64-bit FIXED POINT: contributions are scaled by 2^k, with k
// chosen from max|g_out| (a streaming pre-pass) so that no voxel sum can overflow; with ~49 bits
// below the largest value this is more accurate than f32 accumulation and, integer adds being
// associative, bit-reproducible.  Four samples per thread are in flight (loads issued before any
// use).  One plain read-modify-write of g_d per tile at the end: no global atomics.
constexpr int RT_Z = 16, RT_Y = 16, RT_X = 32;
constexpr int RT_THREADS = 1024;
constexpr int RT_VMAX = 32;  // views per launch (host loops over chunks)
constexpr int RT_UNROLL = 4;

struct ViewBox {           // per (block, view), in LDS
  int lo[3], ext[3];       // sample bounding box (z,y,x) and extents
  unsigned mx, my;         // magic reciprocals ceil(2^32/ext) for x and y
};

// max |x| over n floats -> *out (as float bits; non-negative floats order like unsigned ints)
__global__ void __launch_bounds__(256) absmax_kernel(const float* __restrict__ x, int64_t n, unsigned* out) {
  __shared__ float red[16];
  float m = 0.f;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x * 4;
  for (int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < n; i += stride) {
    if (i + 3 < n) {
      const float4 v = *reinterpret_cast<const float4*>(x + i);
      m = fmaxf(m, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
    } else {
      for (int64_t j = i; j < n; ++j) m = fmaxf(m, fabsf(x[j]));
    }
  }
  m = block_max(m, red);
  if (threadIdx.x == 0 && m > 0.f) atomicMax(out, __float_as_uint(fminf(m, 3.0e38f)));
}

__global__ void __launch_bounds__(RT_THREADS) rotate_bwd_tiled_kernel(const float* __restrict__ g_out,
                                                                      const float* __restrict__ rot,
                                                                      float* __restrict__ g_d,
                                                                      const unsigned* __restrict__ gmax_bits,
                                                                      float bound_factor, int V, int D, int H,
                                                                      int W, int tiles_y, int tiles_x) {
  __shared__ unsigned long long acc[RT_Z * RT_Y * RT_X];
  __shared__ ViewBox vbox[RT_VMAX];
  const int t = threadIdx.x;
  const int bx = blockIdx.x % tiles_x;
  const int by = (blockIdx.x / tiles_x) % tiles_y;
  const int bz = blockIdx.x / (tiles_x * tiles_y);
  const int z0 = bz * RT_Z, y0 = by * RT_Y, x0 = bx * RT_X;
  const int z1 = min(z0 + RT_Z, D) - 1, y1 = min(y0 + RT_Y, H) - 1, x1 = min(x0 + RT_X, W) - 1;  // inclusive
  for (int i = t; i < RT_Z * RT_Y * RT_X; i += RT_THREADS) acc[i] = 0ull;
  // fixed-point scale 2^k: |any voxel sum| <= bound_factor * max|g_out| must stay below 2^62
  const float gmax = __uint_as_float(*gmax_bits);
  if (!(gmax > 0.f)) return;                      // all-zero (or NaN-free empty) gradient: nothing to add
  int ebound;
  frexpf(gmax * bound_factor, &ebound);           // gmax*bound_factor < 2^ebound
  const int kexp = 62 - ebound;
  const float fscale = ldexpf(1.f, min(max(kexp, -120), 120));
  const float fscale2 = ldexpf(1.f, kexp - min(max(kexp, -120), 120));  // split: 2^k may exceed the float range

  if (t < V) {
    const float* r = rot + t * 9;
    const int n[3] = {D, H, W};
    const int tlo[3] = {z0, y0, x0}, thi[3] = {z1, y1, x1};
    double A[3][3], c[3], big[3];
    for (int a = 0; a < 3; ++a) {
      const double ha = 0.5 * (n[a] - 1);
      double rs = 0.0;
      big[a] = 4.0;
      for (int b = 0; b < 3; ++b) {
        const double sb = n[b] > 1 ? 2.0 / (n[b] - 1) : 0.0;
        A[a][b] = (double)r[a * 3 + b] * sb * ha;
        rs += (double)r[a * 3 + b];
        big[a] += fabs(A[a][b]) * (n[b] - 1);
      }
      c[a] = (1.0 - rs) * ha;
      big[a] += fabs(c[a]);
    }
    const double det = A[0][0] * (A[1][1] * A[2][2] - A[1][2] * A[2][1]) -
                       A[0][1] * (A[1][0] * A[2][2] - A[1][2] * A[2][0]) +
                       A[0][2] * (A[1][0] * A[2][1] - A[1][1] * A[2][0]);
    int lo[3] = {0, 0, 0}, hi[3] = {D - 1, H - 1, W - 1};
    if (fabs(det) > 1e-9) {
      double inv[3][3];
      inv[0][0] = (A[1][1] * A[2][2] - A[1][2] * A[2][1]) / det;
      inv[0][1] = (A[0][2] * A[2][1] - A[0][1] * A[2][2]) / det;
      inv[0][2] = (A[0][1] * A[1][2] - A[0][2] * A[1][1]) / det;
      inv[1][0] = (A[1][2] * A[2][0] - A[1][0] * A[2][2]) / det;
      inv[1][1] = (A[0][0] * A[2][2] - A[0][2] * A[2][0]) / det;
      inv[1][2] = (A[0][2] * A[1][0] - A[0][0] * A[1][2]) / det;
      inv[2][0] = (A[1][0] * A[2][1] - A[1][1] * A[2][0]) / det;
      inv[2][1] = (A[0][1] * A[2][0] - A[0][0] * A[2][1]) / det;
      inv[2][2] = (A[0][0] * A[1][1] - A[0][1] * A[1][0]) / det;
      // box centre / half widths in x-space -> centre / half extents in sample space
      double oc[3] = {0, 0, 0}, oe[3] = {0, 0, 0};
      for (int a = 0; a < 3; ++a) {
        const double xl = (tlo[a] == 0) ? -big[a] : tlo[a] - 1.02;
        const double xh = (thi[a] == n[a] - 1) ? big[a] : thi[a] + 1.02;
        const double mid = 0.5 * (xl + xh) - c[a], half = 0.5 * (xh - xl);
        for (int b = 0; b < 3; ++b) {
          oc[b] += inv[b][a] * mid;
          oe[b] += fabs(inv[b][a]) * half;
        }
      }
      for (int b = 0; b < 3; ++b) {
        // the catchment box already carries a 0.02-cell pad: no extra integer margin needed
        lo[b] = (int)fmin(fmax(floor(oc[b] - oe[b]), 0.0), (double)n[b]);
        hi[b] = (int)fmax(fmin(ceil(oc[b] + oe[b]), (double)(n[b] - 1)), -1.0);
      }
    }
    ViewBox vb;
    for (int b = 0; b < 3; ++b) { vb.lo[b] = lo[b]; vb.ext[b] = max(hi[b] - lo[b] + 1, 0); }
    vb.mx = vb.ext[2] > 1 ? (unsigned)(((1ull << 32) + vb.ext[2] - 1) / vb.ext[2]) : 0u;
    vb.my = vb.ext[1] > 1 ? (unsigned)(((1ull << 32) + vb.ext[1] - 1) / vb.ext[1]) : 0u;
    vbox[t] = vb;
  }
  __syncthreads();

  for (int v = 0; v < V; ++v) {
    const int lz = vbox[v].lo[0], ly = vbox[v].lo[1], lx = vbox[v].lo[2];
    const int ey = vbox[v].ext[1], ex = vbox[v].ext[2];
    const unsigned mx = vbox[v].mx, my = vbox[v].my;
    const int total = vbox[v].ext[0] * ey * ex;
    if (total <= 0) continue;
    const float* r = rot + v * 9;
    const float r0 = r[0], r1 = r[1], r2 = r[2], r3 = r[3], r4 = r[4], r5 = r[5], r6 = r[6], r7 = r[7], r8 = r[8];
    const float* gv = g_out + (int64_t)v * D * H * W;
    for (int base = t; base < total; base += RT_THREADS * RT_UNROLL) {
      float gval[RT_UNROLL];
      int oz[RT_UNROLL], oy[RT_UNROLL], ox[RT_UNROLL];
#pragma unroll
      for (int u = 0; u < RT_UNROLL; ++u) {
        const int i = base + u * RT_THREADS;
        // i = (z*ey + y)*ex + x ; exact magic division (i < 2^22, ext < 2^11)
        const int rowi = ex > 1 ? (int)__umulhi((unsigned)i, mx) : i;
        ox[u] = i - rowi * ex;
        const int zi = ey > 1 ? (int)__umulhi((unsigned)rowi, my) : rowi;
        oy[u] = rowi - zi * ey;
        oz[u] = zi;
        gval[u] = 0.f;
        if (i < total) gval[u] = gv[((int64_t)(lz + oz[u]) * H + (ly + oy[u])) * W + (lx + ox[u])];
      }
#pragma unroll
      for (int u = 0; u < RT_UNROLL; ++u) {
        const float g = gval[u];
        if (g == 0.f) continue;
        const float gz_ = lin_coord(lz + oz[u], D), gy_ = lin_coord(ly + oy[u], H), gx_ = lin_coord(lx + ox[u], W);
        const Axis az = axis_setup(r0 * gz_ + r1 * gy_ + r2 * gx_, D);
        if (az.i1 < z0 || az.i0 > z1) continue;
        const Axis ay = axis_setup(r3 * gz_ + r4 * gy_ + r5 * gx_, H);
        if (ay.i1 < y0 || ay.i0 > y1) continue;
        const Axis ax = axis_setup(r6 * gz_ + r7 * gy_ + r8 * gx_, W);
        if (ax.i1 < x0 || ax.i0 > x1) continue;
        const int iz[2] = {az.i0, az.i1}, iy[2] = {ay.i0, ay.i1}, ix[2] = {ax.i0, ax.i1};
        const float wz[2] = {1.f - az.w1, az.w1}, wy[2] = {1.f - ay.w1, ay.w1}, wx[2] = {1.f - ax.w1, ax.w1};
        const float gs = g * fscale * fscale2;              // fixed-point scale applied once per sample
#pragma unroll
        for (int a = 0; a < 2; ++a) {
          if (iz[a] < z0 || iz[a] > z1) continue;
#pragma unroll
          for (int b = 0; b < 2; ++b) {
            if (iy[b] < y0 || iy[b] > y1) continue;
            const float wzy = wz[a] * wy[b] * gs;
            const int rowi = ((iz[a] - z0) * RT_Y + (iy[b] - y0)) * RT_X - x0;   // 32-bit LDS index
#pragma unroll
            for (int cI = 0; cI < 2; ++cI) {
              if (ix[cI] < x0 || ix[cI] > x1) continue;
              const float c = wzy * wx[cI];
              if (c != 0.f) {
                // float -> 64-bit fixed point in ~7 instructions: a float carries 24 significant bits, so take
                // them as an int32 at exponent s and shift (the library conversion is a ~15-instruction sequence)
                int e;
                (void)frexpf(c, &e);
                const int sh = max(e - 24, 0);
                const long long q = (long long)(int)rintf(ldexpf(c, -sh)) << sh;
                atomicAdd(&acc[rowi + ix[cI]], (unsigned long long)q);
              }
            }
          }
        }
      }
    }
  }
  __syncthreads();
  for (int i = t; i < RT_Z * RT_Y * RT_X; i += RT_THREADS) {
    const int lx_ = i % RT_X, ly_ = (i / RT_X) % RT_Y, lz_ = i / (RT_X * RT_Y);
    const int z = z0 + lz_, y = y0 + ly_, x = x0 + lx_;
    if (z < D && y < H && x < W) {
      const long long q = (long long)acc[i];
      if (q != 0) g_d[((int64_t)z * H + y) * W + x] += (float)ldexp((double)q, -kexp);
    }
  }
}

static int check_dims(int B, int X, int Y, int Z, int C) {
  NFS_REQUIRE(B > 0 && X > 0 && Y > 0 && Z > 0 && C > 0, "warp: non-positive dimension");
  NFS_REQUIRE((int64_t)B * X * Y * Z * C < (int64_t)1 << 40, "warp: tensor too large");
  return NFS_OK;
}

}  // namespace nfs

using namespace nfs;

extern "C" {

int nfs_warp3d_fwd(const float* imgs, const float* coords, float* out, int B, int X, int Y, int Z, int C,
                   nfs_stream_t stream) {
  NFS_REQUIRE(imgs && coords && out, "nfs_warp3d_fwd: null pointer");
  if (int e = check_dims(B, X, Y, Z, C)) return e;
  WarpArgs a{imgs, coords, B, X, Y, Z, C, 1};
  const int64_t n = (int64_t)B * X * Y * Z;
  hipLaunchKernelGGL(warp_fwd_kernel<COORD_EXPLICIT>, dim3(blocks_for(n, 256)), dim3(256), 0, as_stream(stream), a, out);
  return check_launch("nfs_warp3d_fwd");
}

int nfs_warp3d_bwd(const float* imgs, const float* coords, const float* g_out, float* g_imgs_acc, float* g_coords,
                   int B, int X, int Y, int Z, int C, nfs_stream_t stream) {
  NFS_REQUIRE(coords && g_out, "nfs_warp3d_bwd: null pointer");
  NFS_REQUIRE(!g_coords || imgs, "nfs_warp3d_bwd: g_coords needs imgs");
  if (int e = check_dims(B, X, Y, Z, C)) return e;
  WarpArgs a{imgs, coords, B, X, Y, Z, C, 1};
  const int64_t n = (int64_t)B * X * Y * Z;
  hipLaunchKernelGGL(warp_bwd_kernel<COORD_EXPLICIT>, dim3(blocks_for(n, 256)), dim3(256), 0, as_stream(stream), a,
                     g_out, g_imgs_acc, g_coords);
  return check_launch("nfs_warp3d_bwd");
}

int nfs_rotate_fwd(const float* d, const float* rot, float* out, int V, int D, int H, int W, int C,
                   nfs_stream_t stream) {
  NFS_REQUIRE(d && rot && out, "nfs_rotate_fwd: null pointer");
  if (int e = check_dims(V, D, H, W, C)) return e;
  WarpArgs a{d, rot, V, D, H, W, C, 0};
  const int64_t n = (int64_t)V * D * H * W;
  hipLaunchKernelGGL(warp_fwd_kernel<COORD_ROTATE>, dim3(blocks_for(n, 256)), dim3(256), 0, as_stream(stream), a, out);
  return check_launch("nfs_rotate_fwd");
}

int nfs_rotate_bwd(const float* g_out, const float* rot, float* g_d_acc, int V, int D, int H, int W, int C,
                   float* workspace, nfs_stream_t stream) {
  NFS_REQUIRE(g_out && rot && g_d_acc, "nfs_rotate_bwd: null pointer");
  if (int e = check_dims(V, D, H, W, C)) return e;
  if (C == 1 && workspace) {
    const int tz = (D + RT_Z - 1) / RT_Z, ty = (H + RT_Y - 1) / RT_Y, tx = (W + RT_X - 1) / RT_X;
    unsigned* gmax_bits = reinterpret_cast<unsigned*>(workspace);
    if (hipMemsetAsync(gmax_bits, 0, sizeof(unsigned), as_stream(stream)) != hipSuccess) {
      set_error("nfs_rotate_bwd: memset failed");
      return NFS_ELAUNCH;
    }
    const int64_t n = (int64_t)V * D * H * W;
    hipLaunchKernelGGL(absmax_kernel, dim3(2048), dim3(256), 0, as_stream(stream), g_out, n, gmax_bits);
    // a voxel collects, per view, unit total weight from interior samples and at most ~max(D,H,W)
    // clamped samples per face direction; 4*nmax per view is a safe bound on the summed weights
    const int nmax = D > H ? (D > W ? D : W) : (H > W ? H : W);
    const float bound_factor = 4.f * (float)nmax * (float)V + 8.f;
    for (int v0 = 0; v0 < V; v0 += RT_VMAX) {
      const int vn = V - v0 < RT_VMAX ? V - v0 : RT_VMAX;
      hipLaunchKernelGGL(rotate_bwd_tiled_kernel, dim3(tz * ty * tx), dim3(RT_THREADS), 0, as_stream(stream),
                         g_out + (int64_t)v0 * D * H * W, rot + v0 * 9, g_d_acc, gmax_bits, bound_factor, vn, D, H, W,
                         ty, tx);
    }
    return check_launch("nfs_rotate_bwd(tiled)");
  }
  WarpArgs a{nullptr, rot, V, D, H, W, C, 0};
  const int64_t n = (int64_t)V * D * H * W;
  hipLaunchKernelGGL(warp_bwd_kernel<COORD_ROTATE>, dim3(blocks_for(n, 256)), dim3(256), 0, as_stream(stream), a,
                     g_out, g_d_acc, (float*)nullptr);
  return check_launch("nfs_rotate_bwd");
}

int nfs_advect_fwd(const float* d, const float* vel, float* out, int D, int H, int W, int C, nfs_stream_t stream) {
  NFS_REQUIRE(d && vel && out, "nfs_advect_fwd: null pointer");
  if (int e = check_dims(1, D, H, W, C)) return e;
  WarpArgs a{d, vel, 1, D, H, W, C, 0};
  const int64_t n = (int64_t)D * H * W;
  if (C == 1 && W >= 2 && n % 4 == 0 && n < ((int64_t)1 << 30)) {
    hipLaunchKernelGGL(advect1_kernel<false>, dim3(blocks_for(n / 4, 256)), dim3(256), 0, as_stream(stream), d, vel,
                       (const float*)nullptr, out, D, H, W);
    return check_launch("nfs_advect_fwd(x4)");
  }
  hipLaunchKernelGGL(warp_fwd_kernel<COORD_ADVECT>, dim3(blocks_for(n, 256)), dim3(256), 0, as_stream(stream), a, out);
  return check_launch("nfs_advect_fwd");
}

int nfs_advect_bwd(const float* d, const float* vel, const float* g_out, float* g_d_acc, float* g_vel, int D, int H,
                   int W, int C, nfs_stream_t stream) {
  NFS_REQUIRE(vel && g_out, "nfs_advect_bwd: null pointer");
  NFS_REQUIRE(!g_vel || d, "nfs_advect_bwd: g_vel needs d");
  NFS_REQUIRE(g_d_acc || g_vel, "nfs_advect_bwd: nothing to compute");
  if (int e = check_dims(1, D, H, W, C)) return e;
  WarpArgs a{d, vel, 1, D, H, W, C, 0};
  const int64_t n = (int64_t)D * H * W;
  if (C == 1 && !g_d_acc && W >= 2 && n % 4 == 0 && n < ((int64_t)1 << 30)) {   // velocity gradient only: no atomics
    hipLaunchKernelGGL(advect1_kernel<true>, dim3(blocks_for(n / 4, 256)), dim3(256), 0, as_stream(stream), d, vel,
                       g_out, g_vel, D, H, W);
    return check_launch("nfs_advect_bwd(x4)");
  }
  hipLaunchKernelGGL(warp_bwd_kernel<COORD_ADVECT>, dim3(blocks_for(n, 256)), dim3(256), 0, as_stream(stream), a,
                     g_out, g_d_acc, g_vel);
  return check_launch("nfs_advect_bwd");
}

}  // extern "C"
