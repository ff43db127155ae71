// A2/A3/A11: border-replicating trilinear resampling family (transform.py:238-269,
// 343-433, 557-569, 611-628).  One thread per output voxel, W fastest => coalesced
// stores and near-coalesced gathers; coordinates are generated in registers (the
// reference materialises 3 coordinate volumes + 8 index/weight/gather tensors).
// HBM-bound: algorithmic bytes = read volume once + write output once.
#include "common.h"

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
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] = src[t.o[k] * a.C + c];
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
                   nfs_stream_t stream) {
  NFS_REQUIRE(g_out && rot && g_d_acc, "nfs_rotate_bwd: null pointer");
  if (int e = check_dims(V, D, H, W, C)) return e;
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
  hipLaunchKernelGGL(warp_bwd_kernel<COORD_ADVECT>, dim3(blocks_for(n, 256)), dim3(256), 0, as_stream(stream), a,
                     g_out, g_d_acc, g_vel);
  return check_launch("nfs_advect_bwd");
}

}  // extern "C"
