// 2-D twins of the resampling family and the remaining TNST grid operators:
//   batch_warp2d / _interpolate2d      transform.py:206-236, 280-341   (A2, 2-D; pins the reference's only KAT)
//   advect, 2-D branch, order 1        transform.py:583-588            (A11, 2-D)
//   advect order 2 (MacCormack)        transform.py:570-582 / 590-607  (the reference's clamp is broken
//                                      -- tf.to_int32 of [-1,1] coordinates, d_max[grids] -- here done as intended)
//   curl of a stream function          transform.py:517-555            (SURVEY 8(f)-4), forward and adjoint
// Images are small (128^2 ... 512 x 1024) next to the volumes: one thread per output element, exact reference stencil
// (x0 = floor, x1 = x0 + 1, both clipped to [0, n-1], weight dx = x - float(clipped x0)).
#include "common.h"
#include <cstring>

namespace nfs {

enum Coord2 { C2_EXPLICIT = 0, C2_ADVECT = 1 };

struct Warp2Args {
  const float* src;     // [B,X,Y,C] (explicit: batched; advect: B = 1)
  const float* coords;  // explicit [B,2,X,Y] | vel [X,Y,2]
  int B, X, Y, C;
};

template <int KIND>
__device__ __forceinline__ void coords2_at(const Warp2Args& a, int b, int x, int y, int64_t pix, float& cx, float& cy) {
  if (KIND == C2_EXPLICIT) {
    const int64_t n = (int64_t)a.X * a.Y;
    const float* c = a.coords + (int64_t)b * 2 * n;
    cx = c[pix];
    cy = c[n + pix];
  } else {
    const float* v = a.coords + pix * 2;
    cx = lin_coord(x, a.X) - v[0];
    cy = lin_coord(y, a.Y) - v[1];
  }
}

template <int KIND>
__global__ void __launch_bounds__(256) warp2d_fwd_kernel(Warp2Args a, float* __restrict__ out) {
  const int64_t n = (int64_t)a.X * a.Y;
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= n * a.B) return;
  const int b = (int)(gid / n);
  const int64_t pix = gid - (int64_t)b * n;
  const int y = (int)(pix % a.Y), x = (int)(pix / a.Y);
  float cx, cy;
  coords2_at<KIND>(a, b, x, y, pix, cx, cy);
  const Axis ax = axis_setup(cx, a.X), ay = axis_setup(cy, a.Y);
  const float* src = a.src + (int64_t)b * n * a.C;
  const float w00 = (1.f - ax.w1) * (1.f - ay.w1), w01 = (1.f - ax.w1) * ay.w1, w10 = ax.w1 * (1.f - ay.w1),
              w11 = ax.w1 * ay.w1;
  const int64_t o00 = ((int64_t)ax.i0 * a.Y + ay.i0) * a.C, o01 = ((int64_t)ax.i0 * a.Y + ay.i1) * a.C,
                o10 = ((int64_t)ax.i1 * a.Y + ay.i0) * a.C, o11 = ((int64_t)ax.i1 * a.Y + ay.i1) * a.C;
  float* o = out + gid * a.C;
  for (int c = 0; c < a.C; ++c)   // add_n order of the reference: w00 I00 + w01 I01 + w10 I10 + w11 I11
    o[c] = ((w00 * src[o00 + c] + w01 * src[o01 + c]) + w10 * src[o10 + c]) + w11 * src[o11 + c];
}

// adjoint: g_src += scatter(g_out) (float atomics), g_coord overwritten (explicit [B,2,X,Y]; advect: g_vel = -g_coord)
template <int KIND>
__global__ void __launch_bounds__(256) warp2d_bwd_kernel(Warp2Args a, const float* __restrict__ g_out,
                                                         float* __restrict__ g_src, float* __restrict__ g_coord) {
  const int64_t n = (int64_t)a.X * a.Y;
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= n * a.B) return;
  const int b = (int)(gid / n);
  const int64_t pix = gid - (int64_t)b * n;
  const int y = (int)(pix % a.Y), x = (int)(pix / a.Y);
  float cx, cy;
  coords2_at<KIND>(a, b, x, y, pix, cx, cy);
  const Axis ax = axis_setup(cx, a.X), ay = axis_setup(cy, a.Y);
  const int64_t boff = (int64_t)b * n * a.C;
  const float wx[2] = {1.f - ax.w1, ax.w1}, wy[2] = {1.f - ay.w1, ay.w1};
  const int ix[2] = {ax.i0, ax.i1}, iy[2] = {ay.i0, ay.i1};
  const float* go = g_out + gid * a.C;
  float gx = 0.f, gy = 0.f;
  for (int c = 0; c < a.C; ++c) {
    const float g = go[c];
    float v[2][2];
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int64_t o = boff + ((int64_t)ix[p] * a.Y + iy[q]) * a.C + c;
        if (g_src) {
          const float contrib = wx[p] * wy[q] * g;
          if (contrib != 0.f) atomicAdd(g_src + o, contrib);
        }
        if (g_coord) v[p][q] = a.src[o];
      }
    if (g_coord) {
      gx += g * (wy[0] * (v[1][0] - v[0][0]) + wy[1] * (v[1][1] - v[0][1]));
      gy += g * (wx[0] * (v[0][1] - v[0][0]) + wx[1] * (v[1][1] - v[1][0]));
    }
  }
  if (g_coord) {
    gx *= (float)(a.X - 1) * 0.5f;
    gy *= (float)(a.Y - 1) * 0.5f;
    if (KIND == C2_ADVECT) {
      g_coord[pix * 2] = -gx;
      g_coord[pix * 2 + 1] = -gy;
    } else {
      float* gc = g_coord + (int64_t)b * 2 * n;
      gc[pix] = gx;
      gc[n + pix] = gy;
    }
  }
}

// ---- MacCormack (order 2), 2-D (D == 1) and 3-D ------------------------------------------------------------------
//   d_fwd = advect(d, v)                         (given: nfs_advect_fwd / nfs_advect2d_fwd)
//   d_bwd = advect(d_fwd, -v)                    (grids_ = mgrid + vel)
//   d_adv = d_fwd + (d - d_bwd) / 2
//   limiter: d_min / d_max = extrema of d over the corners of the back-traced interpolation stencil (what the
//   reference's 2x2(x2) max-pool sampled at the back-traced cell means); where d_adv leaves [d_min, d_max] the
//   first-order value d_fwd is kept ("soft clamp", transform.py:578-582 / 604-607).
__global__ void __launch_bounds__(256) maccormack_kernel(const float* __restrict__ d, const float* __restrict__ vel,
                                                         const float* __restrict__ d_fwd, float* __restrict__ out,
                                                         int D, int H, int W, int C, int nd) {
  const int64_t n = (int64_t)D * H * W;
  const int64_t vox = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (vox >= n) return;
  const int x = (int)(vox % W), y = (int)((vox / W) % H), z = (int)(vox / ((int64_t)W * H));
  const float* v = vel + vox * nd;
  const float vz = nd == 3 ? v[0] : 0.f, vy = v[nd - 2], vx = v[nd - 1];
  const float gz = lin_coord(z, D), gy = lin_coord(y, H), gx = lin_coord(x, W);
  Tri tb, tf;
  Axis az, ay, ax;
  tri_setup(gz - vz, gy - vy, gx - vx, D, H, W, tb, az, ay, ax);     // back-trace (stencil of d_fwd; the limiter's cell)
  tri_setup(gz + vz, gy + vy, gx + vx, D, H, W, tf, az, ay, ax);     // forward trace of d_fwd
  for (int c = 0; c < C; ++c) {
    float lo = d[tb.o[0] * C + c], hi = lo, bwd = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const float dv = d[tb.o[k] * C + c];
      lo = fminf(lo, dv);
      hi = fmaxf(hi, dv);
      bwd += tf.w[k] * d_fwd[tf.o[k] * C + c];
    }
    const float f = d_fwd[vox * C + c];
    const float adv = f + (d[vox * C + c] - bwd) * 0.5f;
    out[vox * C + c] = (adv > hi || lo > adv) ? f : adv;
  }
}

// ---- curl of a stream function (forward differences, last slice replicated; transform.py:517-555) ------------------
// 2-D: s [H,W] -> [H,W,2]: u = ds/dy (axis 0), v = -ds/dx (axis 1).  3-D: s [D,H,W,3] -> [D,H,W,3]:
//   u = dw/dy - dv/dz, v = du/dz - dw/dx, w = dv/dx - du/dy   with x = axis W, y = axis H, z = axis D.
__device__ __forceinline__ int fd_lo(int i, int n) { return i < n - 1 ? i : (n >= 2 ? n - 2 : 0); }

__global__ void __launch_bounds__(256) curl_fwd_kernel(const float* __restrict__ s, float* __restrict__ out, int D, int H,
                                                       int W, int nd) {
  const int64_t n = (int64_t)D * H * W;
  const int64_t vox = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (vox >= n) return;
  const int x = (int)(vox % W), y = (int)((vox / W) % H), z = (int)(vox / ((int64_t)W * H));
  const int xl = fd_lo(x, W), yl = fd_lo(y, H), zl = fd_lo(z, D);
  const int dxs = W >= 2, dys = H >= 2, dzs = D >= 2;
  if (nd == 2) {
    // u = s[y+1,x] - s[y,x] at row yl;  v = s[y,x] - s[y,x+1] at column xl
    const float u = dys ? s[(int64_t)(yl + 1) * W + x] - s[(int64_t)yl * W + x] : 0.f;
    const float v = dxs ? s[(int64_t)y * W + xl] - s[(int64_t)y * W + xl + 1] : 0.f;
    out[vox * 2] = u;
    out[vox * 2 + 1] = v;
    return;
  }
  auto at = [&](int zz, int yy, int xx, int c) { return s[(((int64_t)zz * H + yy) * W + xx) * 3 + c]; };
  const float dvdx = dxs ? at(z, y, xl + 1, 1) - at(z, y, xl, 1) : 0.f, dwdx = dxs ? at(z, y, xl + 1, 2) - at(z, y, xl, 2) : 0.f;
  const float dudy = dys ? at(z, yl + 1, x, 0) - at(z, yl, x, 0) : 0.f, dwdy = dys ? at(z, yl + 1, x, 2) - at(z, yl, x, 2) : 0.f;
  const float dudz = dzs ? at(zl + 1, y, x, 0) - at(zl, y, x, 0) : 0.f, dvdz = dzs ? at(zl + 1, y, x, 1) - at(zl, y, x, 1) : 0.f;
  out[vox * 3] = dwdy - dvdz;
  out[vox * 3 + 1] = dudz - dwdx;
  out[vox * 3 + 2] = dvdx - dudy;
}

// adjoint: g_s = curl^T g.  A forward difference taken at cell l = fd_lo(i) contributes +g to s[l+1] and -g to s[l];
// cell i receives from the outputs whose l equals i (outputs i, and n-1 as well when i == n-2) and whose l+1 equals i.
// Written as a gather over the (at most three) contributing outputs per axis: no atomics, deterministic.
__device__ __forceinline__ float fd_adj(const float* __restrict__ g, int64_t base, int64_t stride, int i, int n, int ch,
                                        int nch) {
  // sum over outputs o along this axis: coefficient of s[i] in (s[lo(o)+1] - s[lo(o)])
  if (n < 2) return 0.f;
  float r = 0.f;
  if (i >= 1) {                       // s[i] is the upper sample of outputs with lo == i-1
    r += g[(base + (int64_t)(i - 1) * stride) * nch + ch];
    if (i == n - 1) r += g[(base + (int64_t)(n - 1) * stride) * nch + ch];   // replicated last slice (lo = n-2)
  }
  if (i <= n - 2) {                   // s[i] is the lower sample of outputs with lo == i
    r -= g[(base + (int64_t)i * stride) * nch + ch];
    if (i == n - 2) r -= g[(base + (int64_t)(n - 1) * stride) * nch + ch];
  }
  return r;
}

__global__ void __launch_bounds__(256) curl_bwd_kernel(const float* __restrict__ g, float* __restrict__ gs, int D, int H,
                                                       int W, int nd) {
  const int64_t n = (int64_t)D * H * W;
  const int64_t vox = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (vox >= n) return;
  const int x = (int)(vox % W), y = (int)((vox / W) % H), z = (int)(vox / ((int64_t)W * H));
  const int64_t bx = (int64_t)z * H * W + (int64_t)y * W;     // base with x = 0, stride 1
  const int64_t by = (int64_t)z * H * W + x;                  // base with y = 0, stride W
  const int64_t bz = (int64_t)y * W + x;                      // base with z = 0, stride H*W
  if (nd == 2) {
    // u = +d/dy s, v = -d/dx s
    gs[vox] = fd_adj(g, by, W, y, H, 0, 2) - fd_adj(g, bx, 1, x, W, 1, 2);
    return;
  }
  // u = dw/dy - dv/dz ; v = du/dz - dw/dx ; w = dv/dx - du/dy      (channels of s: 0 = u-, 1 = v-, 2 = w-potential)
  // d/ds0: +d/dz (into v) - d/dy (into w);  d/ds1: -d/dz (into u) + d/dx (into w);  d/ds2: +d/dy (into u) - d/dx (into v)
  gs[vox * 3] = fd_adj(g, bz, (int64_t)H * W, z, D, 1, 3) - fd_adj(g, by, W, y, H, 2, 3);
  gs[vox * 3 + 1] = fd_adj(g, bx, 1, x, W, 2, 3) - fd_adj(g, bz, (int64_t)H * W, z, D, 0, 3);
  gs[vox * 3 + 2] = fd_adj(g, by, W, y, H, 0, 3) - fd_adj(g, bx, 1, x, W, 1, 3);
}

static int check_dims2(int B, int X, int Y, int C) {
  NFS_REQUIRE(B > 0 && X > 0 && Y > 0 && C > 0, "warp2d: non-positive dimension");
  NFS_REQUIRE((int64_t)B * X * Y * C < (int64_t)1 << 40, "warp2d: tensor too large");
  return NFS_OK;
}

}  // namespace nfs

using namespace nfs;

extern "C" {

int nfs_warp2d_fwd(const float* imgs, const float* coords, float* out, int B, int X, int Y, int C, nfs_stream_t stream) {
  NFS_REQUIRE(imgs && coords && out, "nfs_warp2d_fwd: null pointer");
  if (int e = check_dims2(B, X, Y, C)) return e;
  Warp2Args a{imgs, coords, B, X, Y, C};
  hipLaunchKernelGGL(warp2d_fwd_kernel<C2_EXPLICIT>, dim3(blocks_for((int64_t)B * X * Y, 256)), dim3(256), 0,
                     as_stream(stream), a, out);
  return check_launch("nfs_warp2d_fwd");
}

int nfs_warp2d_bwd(const float* imgs, const float* coords, const float* g_out, float* g_imgs_acc, float* g_coords, int B,
                   int X, int Y, int C, nfs_stream_t stream) {
  NFS_REQUIRE(coords && g_out, "nfs_warp2d_bwd: null pointer");
  NFS_REQUIRE(!g_coords || imgs, "nfs_warp2d_bwd: g_coords needs imgs");
  NFS_REQUIRE(g_imgs_acc || g_coords, "nfs_warp2d_bwd: nothing to compute");
  if (int e = check_dims2(B, X, Y, C)) return e;
  Warp2Args a{imgs, coords, B, X, Y, C};
  hipLaunchKernelGGL(warp2d_bwd_kernel<C2_EXPLICIT>, dim3(blocks_for((int64_t)B * X * Y, 256)), dim3(256), 0,
                     as_stream(stream), a, g_out, g_imgs_acc, g_coords);
  return check_launch("nfs_warp2d_bwd");
}

int nfs_advect2d_fwd(const float* d, const float* vel, float* out, int H, int W, int C, nfs_stream_t stream) {
  NFS_REQUIRE(d && vel && out, "nfs_advect2d_fwd: null pointer");
  if (int e = check_dims2(1, H, W, C)) return e;
  Warp2Args a{d, vel, 1, H, W, C};
  hipLaunchKernelGGL(warp2d_fwd_kernel<C2_ADVECT>, dim3(blocks_for((int64_t)H * W, 256)), dim3(256), 0, as_stream(stream),
                     a, out);
  return check_launch("nfs_advect2d_fwd");
}

int nfs_advect2d_bwd(const float* d, const float* vel, const float* g_out, float* g_d_acc, float* g_vel, int H, int W,
                     int C, nfs_stream_t stream) {
  NFS_REQUIRE(vel && g_out, "nfs_advect2d_bwd: null pointer");
  NFS_REQUIRE(!g_vel || d, "nfs_advect2d_bwd: g_vel needs d");
  NFS_REQUIRE(g_d_acc || g_vel, "nfs_advect2d_bwd: nothing to compute");
  if (int e = check_dims2(1, H, W, C)) return e;
  Warp2Args a{d, vel, 1, H, W, C};
  hipLaunchKernelGGL(warp2d_bwd_kernel<C2_ADVECT>, dim3(blocks_for((int64_t)H * W, 256)), dim3(256), 0, as_stream(stream),
                     a, g_out, g_d_acc, g_vel);
  return check_launch("nfs_advect2d_bwd");
}

int nfs_advect_maccormack(const float* d, const float* vel, const float* d_fwd, float* out, int D, int H, int W, int C,
                          int nd, nfs_stream_t stream) {
  NFS_REQUIRE(d && vel && d_fwd && out, "nfs_advect_maccormack: null pointer");
  NFS_REQUIRE(nd == 2 || nd == 3, "nfs_advect_maccormack: nd must be 2 or 3");
  NFS_REQUIRE(nd == 3 || D == 1, "nfs_advect_maccormack: a 2-D field has D == 1");
  NFS_REQUIRE(out != d_fwd && out != d, "nfs_advect_maccormack: out must not alias d or d_fwd");
  if (int e = check_dims2(D, H, W, C)) return e;
  hipLaunchKernelGGL(maccormack_kernel, dim3(blocks_for((int64_t)D * H * W, 256)), dim3(256), 0, as_stream(stream), d, vel,
                     d_fwd, out, D, H, W, C, nd);
  return check_launch("nfs_advect_maccormack");
}

int nfs_curl_fwd(const float* s, float* out, int D, int H, int W, int nd, nfs_stream_t stream) {
  NFS_REQUIRE(s && out, "nfs_curl_fwd: null pointer");
  NFS_REQUIRE(nd == 2 || nd == 3, "nfs_curl_fwd: nd must be 2 or 3");
  NFS_REQUIRE(nd == 3 || D == 1, "nfs_curl_fwd: a 2-D stream function has D == 1");
  if (int e = check_dims2(D, H, W, 1)) return e;
  hipLaunchKernelGGL(curl_fwd_kernel, dim3(blocks_for((int64_t)D * H * W, 256)), dim3(256), 0, as_stream(stream), s, out, D,
                     H, W, nd);
  return check_launch("nfs_curl_fwd");
}

int nfs_curl_bwd(const float* g_out, float* g_s, int D, int H, int W, int nd, nfs_stream_t stream) {
  NFS_REQUIRE(g_out && g_s, "nfs_curl_bwd: null pointer");
  NFS_REQUIRE(nd == 2 || nd == 3, "nfs_curl_bwd: nd must be 2 or 3");
  NFS_REQUIRE(nd == 3 || D == 1, "nfs_curl_bwd: a 2-D stream function has D == 1");
  if (int e = check_dims2(D, H, W, 1)) return e;
  hipLaunchKernelGGL(curl_bwd_kernel, dim3(blocks_for((int64_t)D * H * W, 256)), dim3(256), 0, as_stream(stream), g_out,
                     g_s, D, H, W, nd);
  return check_launch("nfs_curl_bwd");
}

}  // extern "C"

// ---- Laplacian-pyramid gradient normalisation (util.py:57-110) -----------------------------------------
// lap_split: lo = conv(img, k, stride 2, SAME); lo2 = conv_transpose(lo, k * s, shape(img), stride 2); hi = img - lo2
// (s = 5 in 3-D, 4 in 2-D); lap_merge: img = conv_transpose(img, k * s, shape(hi)) + hi; every level is divided by its
// RMS (normalize_std).  k is the 5-tap-per-axis kernel k5x5 / k5x5x5 (util.py:27-46), the same for every channel.
// TF 'SAME' geometry for stride 2, window 5, input n: out = ceil(n / 2), pad_before = (max((out-1)*2 + 5 - n, 0)) / 2.
namespace nfs {

__host__ __device__ __forceinline__ int same_pad_before(int n) {
  const int out = (n + 1) / 2;
  const int total = (out - 1) * 2 + 5 - n;
  return total > 0 ? total / 2 : 0;
}

// out [Do,Ho,Wo,C] = strided correlation of x [D,H,W,C] with k (zero padding).  2-D: D = Do = 1, k [5][5].
__global__ void __launch_bounds__(256) lap_down_kernel(const float* __restrict__ x, const float* __restrict__ k,
                                                       float* __restrict__ out, int D, int H, int W, int C, int nd) {
  const int Do = nd == 3 ? (D + 1) / 2 : 1, Ho = (H + 1) / 2, Wo = (W + 1) / 2;
  const int64_t n = (int64_t)Do * Ho * Wo * C;
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= n) return;
  const int c = (int)(gid % C);
  int64_t v = gid / C;
  const int xo = (int)(v % Wo); v /= Wo;
  const int yo = (int)(v % Ho);
  const int zo = (int)(v / Ho);
  const int pz = nd == 3 ? same_pad_before(D) : 0, py = same_pad_before(H), px = same_pad_before(W);
  const int nz = nd == 3 ? 5 : 1;
  float s = 0.f;
  for (int a = 0; a < nz; ++a) {
    const int z = nd == 3 ? 2 * zo + a - pz : 0;
    if (z < 0 || z >= D) continue;
    for (int b = 0; b < 5; ++b) {
      const int y = 2 * yo + b - py;
      if (y < 0 || y >= H) continue;
      for (int e = 0; e < 5; ++e) {
        const int xx = 2 * xo + e - px;
        if (xx < 0 || xx >= W) continue;
        s += k[(a * 5 + b) * 5 + e] * x[(((int64_t)z * H + y) * W + xx) * C + c];   // nd == 2: a == 0
      }
    }
  }
  out[gid] = s;
}

// The 3-D form of lap_down with the input staged: a block owns 4 x 4 x 16 output voxels and loads the 11 x 11 x 35 input
// voxels (x C) they draw from into LDS once (zeros outside the volume = SAME padding), rows of 35 C consecutive floats;
// 16.5 global loads per output instead of 125, the 125 taps come from LDS and the kernel from LDS too.  Same taps as the
// gather form above (an out-of-range tap adds k * 0); equal to the rounding of the 125-term sum.
constexpr int LD_TZ = 4, LD_TY = 4, LD_TX = 16;
constexpr int LD_IZ = 2 * LD_TZ + 3, LD_IY = 2 * LD_TY + 3, LD_IX = 2 * LD_TX + 3;
template <int C>
__global__ void __launch_bounds__(256) lap_down3_tiled_kernel(const float* __restrict__ x, const float* __restrict__ k,
                                                              float* __restrict__ out, int D, int H, int W, int ntx,
                                                              int nty) {
  __shared__ float tile[LD_IZ * LD_IY * LD_IX * C];
  const int Do = (D + 1) / 2, Ho = (H + 1) / 2, Wo = (W + 1) / 2;
  const int bx = blockIdx.x % ntx, by = (blockIdx.x / ntx) % nty, bz = blockIdx.x / (ntx * nty);
  const int zo0 = bz * LD_TZ, yo0 = by * LD_TY, xo0 = bx * LD_TX;
  const int zi0 = 2 * zo0 - same_pad_before(D), yi0 = 2 * yo0 - same_pad_before(H), xi0 = 2 * xo0 - same_pad_before(W);
  constexpr int ROW = LD_IX * C, NEL = LD_IZ * LD_IY * ROW, BATCH = 10;
  // staging in batches of BATCH loads per thread, all requested before the first is written (clamped addresses, the value
  // outside the volume dropped by a select: as a loop of guarded loads every element was a round trip of its own --
  // 0.159 ms for a 96 MB level)
  for (int i0 = threadIdx.x; i0 < NEL; i0 += 256 * BATCH) {
    float v[BATCH];
#pragma unroll
    for (int j = 0; j < BATCH; ++j) {
      const int i = min(i0 + 256 * j, NEL - 1);
      const int xc = i % ROW, iy = (i / ROW) % LD_IY, iz = i / (ROW * LD_IY);
      const int gz = zi0 + iz, gy = yi0 + iy, gx = xi0 + xc / C;
      const bool in = gz >= 0 && gz < D && gy >= 0 && gy < H && gx >= 0 && gx < W;
      const int cz = min(max(gz, 0), D - 1), cy = min(max(gy, 0), H - 1), cx = min(max(gx, 0), W - 1);
      const float t = x[(((int64_t)cz * H + cy) * W + cx) * C + xc % C];
      v[j] = in ? t : 0.f;
    }
#pragma unroll
    for (int j = 0; j < BATCH; ++j)
      if (i0 + 256 * j < NEL) tile[i0 + 256 * j] = v[j];
  }
  __syncthreads();
  // (the 125 weights: uniform addresses with compile-time offsets = scalar loads, not a second LDS read per tap)
  for (int o = threadIdx.x; o < LD_TZ * LD_TY * LD_TX * C; o += 256) {
    const int c = o % C, xo = (o / C) % LD_TX, yo = (o / (C * LD_TX)) % LD_TY, zo = o / (C * LD_TX * LD_TY);
    if (zo0 + zo >= Do || yo0 + yo >= Ho || xo0 + xo >= Wo) continue;
    const float* t0 = tile + ((2 * zo) * LD_IY + 2 * yo) * ROW + 2 * xo * C + c;
    float s = 0.f;
#pragma unroll
    for (int a = 0; a < 5; ++a)
#pragma unroll
      for (int b = 0; b < 5; ++b)
#pragma unroll
        for (int e = 0; e < 5; ++e) s += k[(a * 5 + b) * 5 + e] * t0[(a * LD_IY + b) * ROW + e * C];
    out[(((int64_t)(zo0 + zo) * Ho + yo0 + yo) * Wo + xo0 + xo) * C + c] = s;
  }
}

// out [D,H,W,C] = scale * conv_transpose(lo [Do,Ho,Wo,C], k) + addend (nullable): the adjoint of lap_down's geometry,
// gathered per output element (taps of matching parity only: <= 3 per axis)
__global__ void __launch_bounds__(256) lap_up_kernel(const float* __restrict__ lo, const float* __restrict__ k,
                                                     const float* __restrict__ addend, float* __restrict__ out, int D,
                                                     int H, int W, int C, int nd, float scale) {
  const int Do = nd == 3 ? (D + 1) / 2 : 1, Ho = (H + 1) / 2, Wo = (W + 1) / 2;
  const int64_t n = (int64_t)D * H * W * C;
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= n) return;
  const int c = (int)(gid % C);
  int64_t v = gid / C;
  const int x = (int)(v % W); v /= W;
  const int y = (int)(v % H);
  const int z = (int)(v / H);
  const int pz = nd == 3 ? same_pad_before(D) : 0, py = same_pad_before(H), px = same_pad_before(W);
  const int nz = nd == 3 ? 5 : 1;
  float s = 0.f;
  // only the taps whose parity matches reach an output element (2 zo = z + pz - a): start at the first such tap and
  // step by two -- at most 3 per axis instead of 5 tests per axis
  const int a0 = nd == 3 ? ((z + pz) & 1) : 0, b0 = (y + py) & 1, e0 = (x + px) & 1;
  for (int a = a0; a < nz; a += 2) {
    const int zo = nd == 3 ? (z + pz - a) >> 1 : 0;
    if (nd == 3 && (z + pz - a < 0 || zo >= Do)) continue;
    for (int b = b0; b < 5; b += 2) {
      const int yo = (y + py - b) >> 1;
      if (y + py - b < 0 || yo >= Ho) continue;
      const float* row = lo + (((int64_t)zo * Ho + yo) * Wo) * C + c;
      const float* kr = k + (a * 5 + b) * 5;
      for (int e = e0; e < 5; e += 2) {
        const int xo = (x + px - e) >> 1;
        if (x + px - e < 0 || xo >= Wo) continue;
        s += kr[e] * row[(int64_t)xo * C];
      }
    }
  }
  s *= scale;
  if (addend) s += addend[gid];
  out[gid] = s;
}

// The 3-D form of lap_up by 2 x 2 x 2 CELLS: the fine voxels 2m - p and 2m - p + 1 of an axis (p = the SAME pad) draw
// from the coarse voxels m, m-1, m-2 with the even taps (k0, k2, k4) and from m, m-1 with the odd ones (k1, k3) -- so a
// thread that owns one cell holds its 3 x 3 x 3 coarse neighbourhood in registers and forms its 8 outputs with
// compile-time tap sets: 125 multiply-adds and 27 LDS reads per 8 outputs and channel, no parity tests, no index
// arithmetic in the sums.  A block = 4 x 4 x 16 cells, the 6 x 6 x 18 coarse voxels (x C) they reach staged in LDS
// (zeros outside the coarse volume: an out-of-range tap adds k * 0; equal to the gather form to rounding).
constexpr int LU_CZ = 4, LU_CY = 4, LU_CX = 16;
template <int C>
// addend_part (nullable): the addend enters as addend / max(rms(addend), eps), its sum of squares given as addend_nparts
// partial sums (every block forms the total itself, in a fixed order); out_part (nullable): this block's sum of out^2
// goes to out_part[blockIdx.x] (fixed order inside the block) -- the RMS normalisation of a pyramid level rides in the
// kernels that produce and consume the level instead of two passes of its own over it
__global__ void __launch_bounds__(256) lap_up3_cell_kernel(const float* __restrict__ lo, const float* __restrict__ k,
                                                           const float* __restrict__ addend, float* __restrict__ out,
                                                           int D, int H, int W, float scale, int nbx, int nby,
                                                           const float* __restrict__ addend_part = nullptr,
                                                           int addend_nparts = 0, double addend_n = 1.0, float eps = 0.f,
                                                           float* __restrict__ out_part = nullptr) {
  constexpr int IZ = LU_CZ + 2, IY = LU_CY + 2, IX = LU_CX + 2, ROW = IX * C;
  __shared__ float tile[IZ * IY * ROW];
  __shared__ float ks[125];
  __shared__ double tot[256];
  __shared__ float red[16];
  float amul = 1.f;
  if (addend_part) {
    double t = 0.0;
    for (int i = threadIdx.x; i < addend_nparts; i += 256) t += (double)addend_part[i];
    tot[threadIdx.x] = t;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
      if ((int)threadIdx.x < w) tot[threadIdx.x] += tot[threadIdx.x + w];
      __syncthreads();
    }
    amul = 1.f / fmaxf(sqrtf((float)(tot[0] / addend_n)), eps);
  }
  const int Do = (D + 1) / 2, Ho = (H + 1) / 2, Wo = (W + 1) / 2;
  const int pz = same_pad_before(D), py = same_pad_before(H), px = same_pad_before(W);
  const int bx = blockIdx.x % nbx, by = (blockIdx.x / nbx) % nby, bz = blockIdx.x / (nbx * nby);
  const int mz0 = (pz >> 1) + bz * LU_CZ, my0 = (py >> 1) + by * LU_CY, mx0 = (px >> 1) + bx * LU_CX;
  if (threadIdx.x < 125) ks[threadIdx.x] = k[threadIdx.x];
  for (int i = threadIdx.x; i < IZ * IY * ROW; i += 256) {
    const int xc = i % ROW, iy = (i / ROW) % IY, iz = i / (ROW * IY);
    const int gz = mz0 - 2 + iz, gy = my0 - 2 + iy, gx = mx0 - 2 + xc / C;
    float v = 0.f;
    if (gz >= 0 && gz < Do && gy >= 0 && gy < Ho && gx >= 0 && gx < Wo)
      v = lo[(((int64_t)gz * Ho + gy) * Wo + gx) * C + xc % C];
    tile[i] = v;
  }
  __syncthreads();
  const int ix = threadIdx.x % LU_CX, iy = (threadIdx.x / LU_CX) % LU_CY, iz = threadIdx.x / (LU_CX * LU_CY);
  const int ze = 2 * (mz0 + iz) - pz, ye = 2 * (my0 + iy) - py, xe = 2 * (mx0 + ix) - px;
  const bool active = !(ze >= D || ye >= H || xe >= W);
  if (!active && !out_part) return;
  float sq = 0.f;
  if (active) {
  // coarse neighbourhood, cq[jz][jy][jx][c] = coarse voxel (m - j) per axis
  float cq[3][3][3][C];
#pragma unroll
  for (int jz = 0; jz < 3; ++jz)
#pragma unroll
    for (int jy = 0; jy < 3; ++jy)
#pragma unroll
      for (int jx = 0; jx < 3; ++jx)
#pragma unroll
        for (int c = 0; c < C; ++c)
          cq[jz][jy][jx][c] = tile[((iz + 2 - jz) * IY + (iy + 2 - jy)) * ROW + (ix + 2 - jx) * C + c];
#pragma unroll
  for (int ez = 0; ez < 2; ++ez)
#pragma unroll
    for (int ey = 0; ey < 2; ++ey)
#pragma unroll
      for (int ex = 0; ex < 2; ++ex) {
        const int z = ze + ez, y = ye + ey, x = xe + ex;
        if (z < 0 || z >= D || y < 0 || y >= H || x < 0 || x >= W) continue;
        const int64_t gid = (((int64_t)z * H + y) * W + x) * C;
#pragma unroll
        for (int c = 0; c < C; ++c) {
          float s = 0.f;
#pragma unroll
          for (int tz = 0; tz < 3 - ez; ++tz)
#pragma unroll
            for (int ty = 0; ty < 3 - ey; ++ty)
#pragma unroll
              for (int tx = 0; tx < 3 - ex; ++tx)
                s += ks[((ez + 2 * tz) * 5 + ey + 2 * ty) * 5 + ex + 2 * tx] * cq[tz][ty][tx][c];
          s *= scale;
          if (addend) s += amul * addend[gid + c];
          out[gid + c] = s;
          sq += s * s;
        }
      }
  }
  if (out_part) {
    sq = block_sum(sq, red);
    if (threadIdx.x == 0) out_part[blockIdx.x] = sq;
  }
}

// The 3-D form of lap_up by ROWS: a block owns one coarse cell row of y (two fine rows), all of x and LR_CZ cells of z
// (2 LR_CZ fine planes); a lane is one float (x, c) of a fine row, so the addend loads and the output stores of a wave are
// 256 contiguous bytes (the cell form's lanes sit 6 floats apart: six partial stores per 384-byte span).  The
// 3 x (LR_CZ + 2) coarse rows the block reaches are staged in LDS once (whole rows: contiguous loads, all requested before
// the first is written); a lane reads its 3 x 3 x (LR_CZ + 2) coarse values from there and forms 2 x 2 LR_CZ outputs.  The
// parity of (y, x) picks one of four weight sets -- taps b = ey + 2 j, e = ex + 2 i, zero where b or e would be 5 -- from an
// LDS table; the z parity is compile-time (taps 0, 2, 4 / 1, 3).  Same taps as the gather form (an out-of-range coarse
// voxel adds k * 0); addend_part / out_part as in the cell form.
constexpr int LR_CZ = 2, LR_T = 320, LR_ROWS = 3 * (LR_CZ + 2), LR_BATCH = 6;
template <int C>
__global__ void __launch_bounds__(LR_T) lap_up3_row_kernel(const float* __restrict__ lo, const float* __restrict__ k,
                                                           const float* __restrict__ addend, float* __restrict__ out,
                                                           int D, int H, int W, float scale, int ncy,
                                                           const float* __restrict__ addend_part, int addend_nparts,
                                                           double addend_n, float eps, float* __restrict__ out_part) {
  extern __shared__ __attribute__((aligned(16))) float cs[];      // [q = coarse plane mz0 - 2 + q][j = coarse row my - j][Wo C]
  __shared__ float wt[2][2][45];
  __shared__ double tot[LR_T];
  __shared__ float red[16];
  const int tid = threadIdx.x;
  float amul = 1.f;
  if (addend_part) {
    double t = 0.0;
    for (int i = tid; i < addend_nparts; i += LR_T) t += (double)addend_part[i];
    tot[tid] = t;
    __syncthreads();
    if (tid < 64) tot[tid] += tot[tid + 256];                       // 320 = 256 + 64 slots, then the tree over 256
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
      if (tid < w) tot[tid] += tot[tid + w];
      __syncthreads();
    }
    amul = 1.f / fmaxf(sqrtf((float)(tot[0] / addend_n)), eps);
  }
  if (tid < 180) {
    const int r = tid % 45, a = r / 9, j = (r / 3) % 3, ii = r % 3, b = tid / 90 + 2 * j, e = (tid / 45) % 2 + 2 * ii;
    (&wt[0][0][0])[tid] = (b < 5 && e < 5) ? k[(a * 5 + b) * 5 + e] : 0.f;
  }
  const int Do = (D + 1) / 2, Ho = (H + 1) / 2, Wo = (W + 1) / 2;
  const int pz = same_pad_before(D), py = same_pad_before(H), px = same_pad_before(W);
  const int RW = Wo * C, NS = LR_ROWS * RW;
  const int by = blockIdx.x % ncy, bz = blockIdx.x / ncy;
  const int my = (py >> 1) + by, mz0 = (pz >> 1) + bz * LR_CZ;
  for (int i0 = tid; i0 < NS; i0 += LR_T * LR_BATCH) {
    float v[LR_BATCH];
#pragma unroll
    for (int b = 0; b < LR_BATCH; ++b) {
      const int i = min(i0 + LR_T * b, NS - 1);
      const int xc = i % RW, row = i / RW, j = row % 3, q = row / 3;
      const int gz = mz0 - 2 + q, gy = my - j;
      const bool in = gz >= 0 && gz < Do && gy >= 0 && gy < Ho;
      const int cz = min(max(gz, 0), Do - 1), cy = min(max(gy, 0), Ho - 1);
      const float t = lo[((int64_t)cz * Ho + cy) * RW + xc];
      v[b] = in ? t : 0.f;
    }
#pragma unroll
    for (int b = 0; b < LR_BATCH; ++b)
      if (i0 + LR_T * b < NS) cs[i0 + LR_T * b] = v[b];
  }
  __syncthreads();
  float sq = 0.f;
  const int64_t plane = (int64_t)H * W * C;
  for (int xc = tid; xc < W * C; xc += LR_T) {
    const int x = xc / C, c = xc % C;
    const int ex = (x + px) & 1, xo0 = (x + px - ex) >> 1;           // coarse voxel of tap i = 0; tap i: xo0 - i
    float cq[LR_CZ + 2][3][3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int gx = xo0 - i;
      const bool in = gx >= 0 && gx < Wo;
      const int off = min(max(gx, 0), Wo - 1) * C + c;
#pragma unroll
      for (int q = 0; q < LR_CZ + 2; ++q)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
          const float t = cs[(q * 3 + j) * RW + off];
          cq[q][j][i] = in ? t : 0.f;
        }
    }
#pragma unroll
    for (int ey = 0; ey < 2; ++ey) {
      const int y = 2 * my - py + ey;
      if (y < 0 || y >= H) continue;
      float w[5][3][3];
#pragma unroll
      for (int a = 0; a < 5; ++a)
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
          for (int i = 0; i < 3; ++i) w[a][j][i] = wt[ey][ex][(a * 3 + j) * 3 + i];
#pragma unroll
      for (int t = 0; t < LR_CZ; ++t)
#pragma unroll
        for (int ez = 0; ez < 2; ++ez) {
          const int z = 2 * (mz0 + t) - pz + ez;
          if (z < 0 || z >= D) continue;
          float s = 0.f;
#pragma unroll
          for (int tz = 0; tz < 3 - ez; ++tz)
#pragma unroll
            for (int j = 0; j < 3; ++j)
#pragma unroll
              for (int i = 0; i < 3; ++i) s += w[ez + 2 * tz][j][i] * cq[t + 2 - tz][j][i];
          s *= scale;
          const int64_t gid = (int64_t)z * plane + (int64_t)y * W * C + xc;
          if (addend) s += amul * addend[gid];       // (all eight requested ahead of their use: measured 10 % slower)
          out[gid] = s;
          sq += s * s;
        }
    }
  }
  if (out_part) {
    sq = block_sum(sq, red);
    if (tid == 0) out_part[blockIdx.x] = sq;
  }
}

// normalize_std / mean-abs normalisation: partial sums (fixed block order => deterministic) then scale
__global__ void __launch_bounds__(256) norm_partial_kernel(const float* __restrict__ x, int64_t n, float* __restrict__ part,
                                                           int use_abs) {
  __shared__ float red[16];
  float s = 0.f;
  const int64_t n4 = ((uintptr_t)x & 15) == 0 ? n / 4 : 0;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    const float4 v = reinterpret_cast<const float4*>(x)[i];
    s += use_abs ? fabsf(v.x) + fabsf(v.y) + fabsf(v.z) + fabsf(v.w) : v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
  }
  for (int64_t i = n4 * 4 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const float v = x[i];
    s += use_abs ? fabsf(v) : v * v;
  }
  s = block_sum(s, red);
  if (threadIdx.x == 0) part[blockIdx.x] = s;
}

__global__ void __launch_bounds__(256) norm_scale_kernel(const float* __restrict__ x, float* __restrict__ out, int64_t n,
                                                         const float* __restrict__ part, int nparts, int use_abs,
                                                         float eps) {
  // every block forms the total itself: each thread a strided slice of the partial sums in double, then a tree over
  // the 256 slots -- a fixed order (the same bits in every block and every run), 4 dependent loads per thread instead
  // of 1024 in one
  __shared__ double tot[256];
  double t = 0.0;
  for (int i = threadIdx.x; i < nparts; i += 256) t += (double)part[i];
  tot[threadIdx.x] = t;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w) tot[threadIdx.x] += tot[threadIdx.x + w];
    __syncthreads();
  }
  const double sum = tot[0];
  const float m = use_abs ? (float)(sum / (double)n) : sqrtf((float)(sum / (double)n));
  const float r = 1.f / fmaxf(m, eps);
  const int64_t n4 = (((uintptr_t)x | (uintptr_t)out) & 15) == 0 ? n / 4 : 0;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    float4 v = reinterpret_cast<const float4*>(x)[i];
    v.x *= r; v.y *= r; v.z *= r; v.w *= r;
    reinterpret_cast<float4*>(out)[i] = v;
  }
  for (int64_t i = n4 * 4 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) out[i] = x[i] * r;
}

}  // namespace nfs

extern "C" {

int nfs_lap_down(const float* x, const float* k, float* out, int D, int H, int W, int C, int nd, nfs_stream_t stream) {
  NFS_REQUIRE(x && k && out, "nfs_lap_down: null pointer");
  NFS_REQUIRE((nd == 2 && D == 1) || nd == 3, "nfs_lap_down: nd must be 2 (D == 1) or 3");
  if (int e = check_dims2(D, H, W, C)) return e;
  const int64_t n = (int64_t)(nd == 3 ? (D + 1) / 2 : 1) * ((H + 1) / 2) * ((W + 1) / 2) * C;
  if (nd == 3 && (C == 1 || C == 3)) {
    const int ntx = ((W + 1) / 2 + LD_TX - 1) / LD_TX, nty = ((H + 1) / 2 + LD_TY - 1) / LD_TY,
              ntz = ((D + 1) / 2 + LD_TZ - 1) / LD_TZ;
    if (C == 1) hipLaunchKernelGGL(lap_down3_tiled_kernel<1>, dim3(ntx * nty * ntz), dim3(256), 0, as_stream(stream), x, k, out, D, H, W, ntx, nty);
    else hipLaunchKernelGGL(lap_down3_tiled_kernel<3>, dim3(ntx * nty * ntz), dim3(256), 0, as_stream(stream), x, k, out, D, H, W, ntx, nty);
    return check_launch("nfs_lap_down");
  }
  hipLaunchKernelGGL(lap_down_kernel, dim3(blocks_for(n, 256)), dim3(256), 0, as_stream(stream), x, k, out, D, H, W, C,
                     nd);
  return check_launch("nfs_lap_down");
}

// the 3-D kernels of lap_up (C = 1 or 3): the row form from 4096 voxels on (NFS_LAP_UP=cell: the 2 x 2 x 2 cell form, from
// 2^21 voxels on, as before round 5); returns the number of blocks (= partial sums), 0 where neither applies
struct LapUpPlan { int kind, blocks, a, b; };          // kind 1: cells (a, b = nbx, nby); 2: rows (a = cell rows of y)
static LapUpPlan lap_up3_plan(int D, int H, int W, int C, int nd) {
  static const bool cell = [] { const char* e = getenv("NFS_LAP_UP"); return e && !strcmp(e, "cell"); }();
  LapUpPlan pl{0, 0, 0, 0};
  if (!(nd == 3 && (C == 1 || C == 3))) return pl;
  const int64_t vox = (int64_t)D * H * W;
  const int pz = same_pad_before(D), py = same_pad_before(H), px = same_pad_before(W);
  // cells m = (p >> 1) ... floor((n - 1 + p) / 2) per axis
  const int ncz = (D - 1 + pz) / 2 - (pz >> 1) + 1, ncy = (H - 1 + py) / 2 - (py >> 1) + 1, ncx = (W - 1 + px) / 2 - (px >> 1) + 1;
  if (cell) {
    if (vox < ((int64_t)1 << 21)) return pl;           // (small volumes: too few blocks)
    const int nbx = (ncx + LU_CX - 1) / LU_CX, nby = (ncy + LU_CY - 1) / LU_CY, nbz = (ncz + LU_CZ - 1) / LU_CZ;
    return LapUpPlan{1, nbx * nby * nbz, nbx, nby};
  }
  const int Wo = (W + 1) / 2;
  // (the staged rows fit LDS: 64 KB is what a launch gets without an opt-in, and the kernel holds ~3.3 KB of static LDS
  // beside them -- weights, the double partial sums, the reduction slots)
  if (vox < 4096 || (int64_t)LR_ROWS * Wo * C * (int64_t)sizeof(float) > 60 * 1024) return pl;
  const int nbz = (ncz + LR_CZ - 1) / LR_CZ;
  return LapUpPlan{2, ncy * nbz, ncy, 0};
}
static void lap_up3_launch(const LapUpPlan& pl, const float* lo, const float* k, float scale, const float* addend, float* out,
                           int D, int H, int W, int C, const float* addend_part, int addend_nparts, double addend_n,
                           float eps, float* out_part, hipStream_t s) {
  if (pl.kind == 1) {
    if (C == 1) hipLaunchKernelGGL(lap_up3_cell_kernel<1>, dim3(pl.blocks), dim3(256), 0, s, lo, k, addend, out, D, H, W, scale, pl.a, pl.b, addend_part, addend_nparts, addend_n, eps, out_part);
    else hipLaunchKernelGGL(lap_up3_cell_kernel<3>, dim3(pl.blocks), dim3(256), 0, s, lo, k, addend, out, D, H, W, scale, pl.a, pl.b, addend_part, addend_nparts, addend_n, eps, out_part);
  } else {
    const size_t lds = (size_t)LR_ROWS * ((W + 1) / 2) * C * sizeof(float);
    if (C == 1) hipLaunchKernelGGL(lap_up3_row_kernel<1>, dim3(pl.blocks), dim3(LR_T), lds, s, lo, k, addend, out, D, H, W, scale, pl.a, addend_part, addend_nparts, addend_n, eps, out_part);
    else hipLaunchKernelGGL(lap_up3_row_kernel<3>, dim3(pl.blocks), dim3(LR_T), lds, s, lo, k, addend, out, D, H, W, scale, pl.a, addend_part, addend_nparts, addend_n, eps, out_part);
  }
}

int nfs_lap_up(const float* lo, const float* k, float scale, const float* addend, float* out, int D, int H, int W, int C,
               int nd, nfs_stream_t stream) {
  NFS_REQUIRE(lo && k && out, "nfs_lap_up: null pointer");
  NFS_REQUIRE((nd == 2 && D == 1) || nd == 3, "nfs_lap_up: nd must be 2 (D == 1) or 3");
  if (int e = check_dims2(D, H, W, C)) return e;
  const int64_t n = (int64_t)D * H * W * C;
  const LapUpPlan pl = lap_up3_plan(D, H, W, C, nd);
  if (pl.kind) {
    lap_up3_launch(pl, lo, k, scale, addend, out, D, H, W, C, nullptr, 0, 1.0, 0.f, nullptr, as_stream(stream));
    return check_launch("nfs_lap_up");
  }
  hipLaunchKernelGGL(lap_up_kernel, dim3(blocks_for(n, 256)), dim3(256), 0, as_stream(stream), lo, k, addend, out, D, H, W,
                     C, nd, scale);
  return check_launch("nfs_lap_up");
}

int nfs_lap_up_rms_parts(int D, int H, int W, int C, int nd) { return lap_up3_plan(D, H, W, C, nd).blocks; }

int nfs_lap_up_rms(const float* lo, const float* k, float scale, const float* addend, const float* addend_part,
                   int addend_nparts, int64_t addend_n, float eps, float* out, float* out_part, int D, int H, int W, int C,
                   int nd, nfs_stream_t stream) {
  NFS_REQUIRE(lo && k && out, "nfs_lap_up_rms: null pointer");
  NFS_REQUIRE(!addend_part || (addend && addend_nparts > 0 && addend_n > 0),
              "nfs_lap_up_rms: addend partial sums without an addend, a count or a length");
  const LapUpPlan pl = lap_up3_plan(D, H, W, C, nd);
  NFS_REQUIRE(pl.kind, "nfs_lap_up_rms: no 3-D kernel instance for this shape (nfs_lap_up_rms_parts() == 0): use nfs_lap_up + "
                       "nfs_normalize_mean");
  lap_up3_launch(pl, lo, k, scale, addend, out, D, H, W, C, addend_part, addend_nparts, (double)addend_n, eps, out_part,
                 as_stream(stream));
  return check_launch("nfs_lap_up_rms");
}

int nfs_normalize_mean(const float* x, float* out, int64_t n, int use_abs, float eps, float* workspace, int ws_floats,
                       nfs_stream_t stream) {
  NFS_REQUIRE(x && out && workspace, "nfs_normalize_mean: null pointer");
  NFS_REQUIRE(n > 0 && ws_floats >= 64, "nfs_normalize_mean: n > 0 and a workspace of >= 64 floats are needed");
  int parts = ws_floats < 1024 ? ws_floats : 1024;
  const unsigned need = blocks_for(n, 256 * 8);
  if ((unsigned)parts > need) parts = (int)need;
  hipLaunchKernelGGL(norm_partial_kernel, dim3(parts), dim3(256), 0, as_stream(stream), x, n, workspace, use_abs);
  hipLaunchKernelGGL(norm_scale_kernel, dim3(blocks_for(n, 256 * 8)), dim3(256), 0, as_stream(stream), x, out, n, workspace,
                     parts, use_abs, eps);
  return check_launch("nfs_normalize_mean");
}

}  // extern "C"
