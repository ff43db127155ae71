// A4: emission-absorption ray integral (styler_3p.py:147-158), its adjoint, the
// global-max normalisation (158) and the fused rotate+render pair that never
// materialises the [V,D,H,W] rotated volume.
//
// Layout: the ray axis is D (slowest), one thread per (view, h, w) ray, w fastest:
// every z step of a wave reads 64 consecutive floats (256 B) of a z-plane.
// HBM-bound.  Algorithmic bytes: fwd 4*V*D*H*W + 4*V*H*W, bwd 8*V*D*H*W + 4*V*H*W;
// fused fwd 4*V*G^3 + 4*V*G^2, fused bwd 8*V*G^3 + 4*V*G^2.
#include "common.h"

namespace nfs {

// ---- un-fused ----------------------------------------------------------------------
__global__ void __launch_bounds__(256) render_fwd_kernel(const float* __restrict__ d, float* __restrict__ img,
                                                         float* __restrict__ raysum, int V, int D, int HW, float tau,
                                                         int liquid) {
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= (int64_t)V * HW) return;
  const int v = (int)(gid / HW);
  const int px = (int)(gid - (int64_t)v * HW);
  const float* col = d + (int64_t)v * D * HW + px;
  float acc = 0.f, I = 0.f;
  // T[z] = exp(-tau * sum_{z' >= z} d[z'])  (reverse cumsum incl. own cell, styler_3p.py:155)
#pragma unroll 4
  for (int z = D - 1; z >= 0; --z) {
    const float s = col[(int64_t)z * HW];
    acc += s;
    I += s * expf(-acc * tau);
  }
  img[gid] = liquid ? 1.f - expf(-acc * tau) : I;
  if (raysum) raysum[gid] = acc;
}

// dI/dd[k] = T[k] - tau * sum_{z<=k} d[z] T[z]; liquid: tau * exp(-tau * sum)
// g_d may alias d (in-place): every thread reads d[z] of its own column before writing g_d[z].
__global__ void __launch_bounds__(256) render_bwd_kernel(const float* d,
                                                         const float* __restrict__ raysum,
                                                         const float* __restrict__ g_img, float* g_d,
                                                         int V, int D, int HW, float tau, int liquid) {
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= (int64_t)V * HW) return;
  const int v = (int)(gid / HW);
  const int px = (int)(gid - (int64_t)v * HW);
  const int64_t base = (int64_t)v * D * HW + px;
  const float total = raysum[gid];
  const float g = g_img[gid];
  if (liquid) {
    const float gd = g * tau * expf(-total * tau);
    for (int z = 0; z < D; ++z) g_d[base + (int64_t)z * HW] = gd;
    return;
  }
  float prefix = 0.f, P = 0.f;
#pragma unroll 4
  for (int z = 0; z < D; ++z) {
    const float s = d[base + (int64_t)z * HW];
    const float T = expf(-(total - prefix) * tau);
    P += s * T;
    g_d[base + (int64_t)z * HW] = g * (T - tau * P);
    prefix += s;
  }
}

// ---- fused rotate + render ---------------------------------------------------------
struct RayGeom {
  float bx, by, bz;  // coordinate of the sample at z-index 0 contribution of (h,w)
  float sx, sy, sz;  // first column of R
};

__device__ __forceinline__ void ray_coords(const float* r, int D, int H, int W, int zi, int h, int w, float& cx,
                                           float& cy, float& cz) {
  const float gx = lin_coord(zi, D), gy = lin_coord(h, H), gz = lin_coord(w, W);
  cx = r[0] * gx + r[1] * gy + r[2] * gz;
  cy = r[3] * gx + r[4] * gy + r[5] * gz;
  cz = r[6] * gx + r[7] * gy + r[8] * gz;
}

__device__ __forceinline__ float tri_sample1(const float* __restrict__ vol, const Tri& t) {
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < 8; ++k) s += t.w[k] * vol[t.o[k]];
  return s;
}

__global__ void __launch_bounds__(256) rotate_render_fwd_kernel(const float* __restrict__ d,
                                                                const float* __restrict__ rot,
                                                                float* __restrict__ img, float* __restrict__ raysum,
                                                                float* __restrict__ d_rot, int V, int D, int H,
                                                                int W, float tau, int liquid) {
  const int HW = H * W;
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= (int64_t)V * HW) return;
  const int v = (int)(gid / HW);
  const int px = (int)(gid - (int64_t)v * HW);
  const int h = px / W, w = px - h * W;
  float r[9];
#pragma unroll
  for (int i = 0; i < 9; ++i) r[i] = rot[v * 9 + i];
  float acc = 0.f, I = 0.f;
  if (W >= 2) {
    // 4 samples (16 paired gathers) are issued before the serial transmittance update consumes them: a
    // one-sample loop is bound by the L2 round trip of each step, not by bandwidth
    int z = D - 1;
    for (; z >= 3; z -= 4) {
      float sv[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        float cx, cy, cz;
        ray_coords(r, D, H, W, z - u, h, w, cx, cy, cz);
        const Axis az = axis_setup(cx, D), ay = axis_setup(cy, H), ax = axis_setup(cz, W);
        sv[u] = tri_sample_pairs(d, H, W, az, ay, ax);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (d_rot) d_rot[((int64_t)v * D + (z - u)) * HW + px] = sv[u];  // rotated volume kept for the adjoint
        acc += sv[u];
        I += sv[u] * expf(-acc * tau);
      }
    }
    for (; z >= 0; --z) {
      float cx, cy, cz;
      ray_coords(r, D, H, W, z, h, w, cx, cy, cz);
      const Axis az = axis_setup(cx, D), ay = axis_setup(cy, H), ax = axis_setup(cz, W);
      const float sone = tri_sample_pairs(d, H, W, az, ay, ax);
      if (d_rot) d_rot[((int64_t)v * D + z) * HW + px] = sone;
      acc += sone;
      I += sone * expf(-acc * tau);
    }
  } else {
    for (int z = D - 1; z >= 0; --z) {
      float cx, cy, cz;
      ray_coords(r, D, H, W, z, h, w, cx, cy, cz);
      Tri t; Axis ax, ay, az;
      tri_setup(cx, cy, cz, D, H, W, t, ax, ay, az);
      const float sone = tri_sample1(d, t);
      if (d_rot) d_rot[((int64_t)v * D + z) * HW + px] = sone;
      acc += sone;
      I += sone * expf(-acc * tau);
    }
  }
  img[gid] = liquid ? 1.f - expf(-acc * tau) : I;
  if (raysum) raysum[gid] = acc;
}

__global__ void __launch_bounds__(256) rotate_render_bwd_kernel(const float* __restrict__ d,
                                                                const float* __restrict__ rot,
                                                                const float* __restrict__ raysum,
                                                                const float* __restrict__ g_img,
                                                                float* __restrict__ g_d, int V, int D, int H, int W,
                                                                float tau, int liquid) {
  const int HW = H * W;
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= (int64_t)V * HW) return;
  const int v = (int)(gid / HW);
  const int px = (int)(gid - (int64_t)v * HW);
  const int h = px / W, w = px - h * W;
  float r[9];
#pragma unroll
  for (int i = 0; i < 9; ++i) r[i] = rot[v * 9 + i];
  const float total = raysum[gid];
  const float g = g_img[gid];
  if (g == 0.f) return;
  const float gl = g * tau * expf(-total * tau);
  float prefix = 0.f, P = 0.f;
  for (int z = 0; z < D; ++z) {
    float cx, cy, cz;
    ray_coords(r, D, H, W, z, h, w, cx, cy, cz);
    Tri t; Axis ax, ay, az;
    tri_setup(cx, cy, cz, D, H, W, t, ax, ay, az);
    float gs;
    if (liquid) {
      gs = gl;
    } else {
      const float s = tri_sample1(d, t);
      const float T = expf(-(total - prefix) * tau);
      P += s * T;
      gs = g * (T - tau * P);
      prefix += s;
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const float c = t.w[k] * gs;
      if (c != 0.f) atomicAdd(g_d + t.o[k], c);
    }
  }
}

// ---- d /= reduce_max(d) ------------------------------------------------------------
// one 1024-thread block per group (a group is <= a few 10^5 pixels)
__global__ void __launch_bounds__(1024) maxnorm_fwd_kernel(const float* __restrict__ img, float* __restrict__ out,
                                                           float* __restrict__ gmax, int n) {
  __shared__ float red[16];
  const float* x = img + (int64_t)blockIdx.x * n;
  float* o = out + (int64_t)blockIdx.x * n;
  float m = -INFINITY;
  for (int i = threadIdx.x; i < n; i += blockDim.x) m = fmaxf(m, x[i]);
  m = block_max(m, red);
  if (threadIdx.x == 0) gmax[blockIdx.x] = m;
  for (int i = threadIdx.x; i < n; i += blockDim.x) o[i] = x[i] / m;
}

// y = x/m, m = max(x):  g_x[i] = g_y[i]/m - [x[i]==m]/ties * sum_j g_y[j] x[j] / m^2
__global__ void __launch_bounds__(1024) maxnorm_bwd_kernel(const float* __restrict__ img,
                                                           const float* __restrict__ gmax,
                                                           const float* __restrict__ g_out,
                                                           float* __restrict__ g_img, int n) {
  __shared__ float red[16];
  const float* x = img + (int64_t)blockIdx.x * n;
  const float* gy = g_out + (int64_t)blockIdx.x * n;
  float* gx = g_img + (int64_t)blockIdx.x * n;
  const float m = gmax[blockIdx.x];
  float s = 0.f, ties = 0.f;
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    const float xi = x[i];
    s += gy[i] * xi;
    ties += (xi == m) ? 1.f : 0.f;
  }
  s = block_sum(s, red);
  ties = block_sum(ties, red);
  const float corr = s / (m * m) / ties;
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    float g = gy[i] / m;
    if (x[i] == m) g -= corr;
    gx[i] = g;
  }
}

}  // namespace nfs

using namespace nfs;

extern "C" {

int nfs_render_fwd(const float* d, float* img, float* raysum, int V, int D, int H, int W, float tau, int liquid,
                   nfs_stream_t stream) {
  NFS_REQUIRE(d && img, "nfs_render_fwd: null pointer");
  NFS_REQUIRE(V > 0 && D > 0 && H > 0 && W > 0, "nfs_render_fwd: non-positive dimension");
  const int64_t n = (int64_t)V * H * W;
  hipLaunchKernelGGL(render_fwd_kernel, dim3(blocks_for(n, 256)), dim3(256), 0, as_stream(stream), d, img, raysum, V, D,
                     H * W, tau, liquid);
  return check_launch("nfs_render_fwd");
}

int nfs_render_bwd(const float* d, const float* raysum, const float* g_img, float* g_d, int V, int D, int H, int W,
                   float tau, int liquid, nfs_stream_t stream) {
  NFS_REQUIRE(d && raysum && g_img && g_d, "nfs_render_bwd: null pointer");
  NFS_REQUIRE(V > 0 && D > 0 && H > 0 && W > 0, "nfs_render_bwd: non-positive dimension");
  const int64_t n = (int64_t)V * H * W;
  hipLaunchKernelGGL(render_bwd_kernel, dim3(blocks_for(n, 256)), dim3(256), 0, as_stream(stream), d, raysum, g_img, g_d,
                     V, D, H * W, tau, liquid);
  return check_launch("nfs_render_bwd");
}

int nfs_rotate_render_fwd(const float* d, const float* rot, float* img, float* raysum, float* d_rot, int V, int D,
                          int H, int W, float tau, int liquid, nfs_stream_t stream) {
  NFS_REQUIRE(d && rot && img, "nfs_rotate_render_fwd: null pointer");
  NFS_REQUIRE(V > 0 && D > 0 && H > 0 && W > 0, "nfs_rotate_render_fwd: non-positive dimension");
  const int64_t n = (int64_t)V * H * W;
  hipLaunchKernelGGL(rotate_render_fwd_kernel, dim3(blocks_for(n, 256)), dim3(256), 0, as_stream(stream), d, rot, img,
                     raysum, d_rot, V, D, H, W, tau, liquid);
  return check_launch("nfs_rotate_render_fwd");
}

int nfs_rotate_render_bwd(const float* d, const float* rot, const float* raysum, const float* g_img, float* g_d_acc,
                          int V, int D, int H, int W, float tau, int liquid, nfs_stream_t stream) {
  NFS_REQUIRE(d && rot && raysum && g_img && g_d_acc, "nfs_rotate_render_bwd: null pointer");
  NFS_REQUIRE(V > 0 && D > 0 && H > 0 && W > 0, "nfs_rotate_render_bwd: non-positive dimension");
  const int64_t n = (int64_t)V * H * W;
  hipLaunchKernelGGL(rotate_render_bwd_kernel, dim3(blocks_for(n, 256)), dim3(256), 0, as_stream(stream), d, rot, raysum,
                     g_img, g_d_acc, V, D, H, W, tau, liquid);
  return check_launch("nfs_rotate_render_bwd");
}

int nfs_maxnorm_fwd(const float* img, float* out, float* gmax, int G, int n, nfs_stream_t stream) {
  NFS_REQUIRE(img && out && gmax, "nfs_maxnorm_fwd: null pointer");
  NFS_REQUIRE(G > 0 && n > 0, "nfs_maxnorm_fwd: non-positive size");
  hipLaunchKernelGGL(maxnorm_fwd_kernel, dim3(G), dim3(1024), 0, as_stream(stream), img, out, gmax, n);
  return check_launch("nfs_maxnorm_fwd");
}

int nfs_maxnorm_bwd(const float* img, const float* gmax, const float* g_out, float* g_img, int G, int n,
                    nfs_stream_t stream) {
  NFS_REQUIRE(img && gmax && g_out && g_img, "nfs_maxnorm_bwd: null pointer");
  NFS_REQUIRE(G > 0 && n > 0, "nfs_maxnorm_bwd: non-positive size");
  hipLaunchKernelGGL(maxnorm_bwd_kernel, dim3(G), dim3(1024), 0, as_stream(stream), img, gmax, g_out, g_img, n);
  return check_launch("nfs_maxnorm_bwd");
}

}  // extern "C"
