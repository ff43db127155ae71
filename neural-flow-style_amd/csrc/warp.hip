// A2/A3/A11: border-replicating trilinear resampling family (transform.py:238-269,
// 343-433, 557-569, 611-628).  One thread per output voxel, W fastest => coalesced
// stores and near-coalesced gathers; coordinates are generated in registers (the
// reference materialises 3 coordinate volumes + 8 index/weight/gather tensors).
// HBM-bound: algorithmic bytes = read volume once + write output once.
#include "common.h"
#include <stdlib.h>

namespace nfs {

typedef float f32x2 __attribute__((ext_vector_type(2)));

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
// Scalar-field advect, 4 W-consecutive voxels per thread, lean stencil: the back-traced voxel coordinate is
// x_a = index_a - vel_a * (n_a - 1)/2 (one FMA; lin_coord(i) - vel mapped to voxel space), border replication by
// clamping it to [0, n-1], base cell min(floor, n-2), weight in [0,1]; the four row-pairs of the stencil are
// dword-aligned 8-byte loads.  D, H, W >= 2.  The first form (per-corner index clamps, two runtime integer
// divisions per voxel) spent ~150 VALU instructions per voxel on a 20-byte-per-voxel stream.
// adv_next (nullable, MODE 2): advect(d, UPDATED vel) of the same voxels -- the next iteration's forward sample, formed
// while the new velocity is still in registers (advect reads the velocity of its own voxel only; the gathered density is
// constant): the forward advect launch of the next iteration and its 96 MB velocity read disappear.  Same arithmetic
// as MODE 0 on the stored velocity: bit-identical to running nfs_advect_fwd afterwards.
struct AdamFused { float* m; float* v; float lr_t, b1, b2, eps; float* adv_next = nullptr;
                   unsigned long long* live = nullptr; unsigned long long* ever = nullptr; };

// live mask (nullable; whole-volume launches only): bit i of the mask = "the eight density corners the back-traced point
// of voxel i interpolates are NOT all equal".  Where they are equal the sample does not depend on the coordinate and the
// velocity gradient of that voxel is g * 0 whatever g is, so the adjoint chain above it (rotate, smooth) need not
// produce g there (rotate_bwd_tiled_kernel skips the tiles / clips to the box that matter).  A wave's lanes hold 64
// consecutive voxels per j: one ballot = one 64-bit word of the mask.
__device__ __forceinline__ bool corners_differ(const F2u (&p)[4]) {
  const float a = p[0].x;
  return !(p[0].y == a && p[1].x == a && p[1].y == a && p[2].x == a && p[2].y == a && p[3].x == a && p[3].y == a);
}

// MODE 0: forward; 1: velocity gradient -> out; 2: velocity gradient consumed on the spot by the TF-Adam update of
// the velocity itself (vel, m, v updated in place: every thread reads and writes only its own 4 voxels of them;
// the 96 MB gradient never goes to HBM)
struct __attribute__((packed, aligned(4))) F3u { float x, y, z; };

// Slab form (zoff, Dfull): vel / g_out / out / the moments hold only the D planes [zoff, zoff + D) of a volume of Dfull
// planes, d is the WHOLE density (the back-traced points leave the slab); zoff = 0, Dfull = D is the whole volume.
// EVER (MODE 2 + LIVE, volumes below 2^31 / 12 voxels): ad.ever is set and the streamed accesses are predicated per lane
template <int MODE, bool LIVE = false, bool EVER = false>   // LIVE: ad.live is set (a compile-time switch: the mask code out of the plain kernels)
__global__ void __launch_bounds__(256) advect1_kernel(const float* __restrict__ d, const float* vel,
                                                      const float* __restrict__ g_out, float* out,
                                                      int D, int H, int W, AdamFused ad, int zoff, int Dfull) {
  constexpr bool BWD = MODE != 0;
  const int n = D * H * W;
  // a wave owns 256 consecutive voxels and lane l takes l, l+64, l+128, l+192: every streamed access (the
  // 12-byte velocity / moment vectors, g_out, out) is then one contiguous 768- or 256-byte run per instruction.
  // (Four consecutive voxels per lane with float4 accesses put the lanes 48 bytes apart: 24 cache lines per
  // instruction, a third of each used.)
  const int lane = threadIdx.x & 63;
  // contiguous z-slab per XCD (workgroups are dealt round-robin to the 8 XCDs): the gathered density planes of
  // neighbouring rows then come through one L2 instead of being fetched into all eight
  // (not for the fused Adam variant: it is dominated by the streamed moments, and eight distant streams measured
  // slower than one front, 0.128 vs 0.112 ms)
  const unsigned per_xcd = gridDim.x / 8;
  const unsigned lb = MODE == 2 ? blockIdx.x : (blockIdx.x % 8) * per_xcd + blockIdx.x / 8;
  const int first = (lb * blockDim.x + (threadIdx.x - lane)) * 4 + lane;
  if (first - lane >= n) return;
  [[maybe_unused]] unsigned long long ew[4] = {~0ull, ~0ull, ~0ull, ~0ull};
  if constexpr (MODE == 2 && LIVE) {
    // `ever` (nullable): bit i = voxel i has been live in SOME iteration since the Adam moments were zeroed.  Where it never
    // was, every velocity gradient so far was an exact zero: m = v = +0, and this iteration's is zero again when the
    // CURRENT mask (`live` on entry: the mask of the forward sample this gradient belongs to) has the bit clear too --
    // ApplyAdam then leaves m, v and the velocity exactly as they are (b1 0 + (1 - b1) (+-0) = +0, x - lr 0 / (0 + eps) = x),
    // so the next forward sample and its mask bit do not change either.  A wave whose 256 voxels are all like that has
    // nothing to read and nothing to write: it returns here.  (Everything else runs the unchanged path: bit-identical.)
    if (ad.ever) {
      const int wbase = __builtin_amdgcn_readfirstlane(first - lane);
      unsigned long long any = 0ull;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int b = wbase + 64 * j;
        ew[j] = 0ull;
        if (b < n) {
          const unsigned long long e0 = ad.ever[b >> 6], e = e0 | ad.live[b >> 6];
          if (e != e0 && lane == 0) ad.ever[b >> 6] = e;
          any |= e;
          ew[j] = e;
        }
      }
      if (any == 0ull) return;
    }
  }
  // EVER: in a wave that runs, the lanes whose voxel never was live take part in nothing either -- their streamed loads
  // get an offset beyond the buffer (the hardware answers zeros without touching memory: exactly what m and v hold
  // there), their stores likewise, their mask bit stays clear.  Traffic then follows the ever-live cache lines, not the
  // ever-live waves.
  [[maybe_unused]] bool act[4] = {true, true, true, true};
  [[maybe_unused]] const __amdgpu_buffer_rsrc_t vel_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(vel), 0, EVER ? (uint32_t)n * 12u : 0u, 0x00020000);
  [[maybe_unused]] const __amdgpu_buffer_rsrc_t m_rs = __builtin_amdgcn_make_buffer_rsrc(ad.m, 0, EVER ? (uint32_t)n * 12u : 0u, 0x00020000);
  [[maybe_unused]] const __amdgpu_buffer_rsrc_t u_rs = __builtin_amdgcn_make_buffer_rsrc(ad.v, 0, EVER ? (uint32_t)n * 12u : 0u, 0x00020000);
  [[maybe_unused]] const __amdgpu_buffer_rsrc_t g_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g_out), 0, EVER ? (uint32_t)n * 4u : 0u, 0x00020000);
  [[maybe_unused]] const __amdgpu_buffer_rsrc_t adv_rs = __builtin_amdgcn_make_buffer_rsrc(ad.adv_next, 0, EVER ? (uint32_t)n * 4u : 0u, 0x00020000);
  typedef unsigned u32x3_t __attribute__((ext_vector_type(3)));
  [[maybe_unused]] auto ld3 = [](__amdgpu_buffer_rsrc_t r, uint32_t o) {
    const u32x3_t q = __builtin_amdgcn_raw_buffer_load_b96(r, o, 0, 0);
    return F3u{__uint_as_float(q.x), __uint_as_float(q.y), __uint_as_float(q.z)};
  };
  [[maybe_unused]] auto st3 = [](F3u f, __amdgpu_buffer_rsrc_t r, uint32_t o) {
    const u32x3_t q = {__float_as_uint(f.x), __float_as_uint(f.y), __float_as_uint(f.z)};
    __builtin_amdgcn_raw_buffer_store_b96(q, r, o, 0, 0);
  };
  const F3u* v3 = reinterpret_cast<const F3u*>(vel);
  const F3u* m3 = reinterpret_cast<const F3u*>(ad.m);
  const F3u* u3 = reinterpret_cast<const F3u*>(ad.v);
  F3u vv[4], mm[4], uu[4];
  float gg[4];
  bool ok[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int idx = first + 64 * j;
    ok[j] = idx < n;
    const int ic = ok[j] ? idx : n - 1;
    if constexpr (EVER) {
      act[j] = ok[j] && ((ew[j] >> lane) & 1ull);
      const uint32_t o12 = act[j] ? (uint32_t)idx * 12u : 0x80000000u, o4 = act[j] ? (uint32_t)idx * 4u : 0x80000000u;
      vv[j] = ld3(vel_rs, o12);
      gg[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(g_rs, o4, 0, 0));
      mm[j] = ld3(m_rs, o12);
      uu[j] = ld3(u_rs, o12);
      continue;
    }
    vv[j] = v3[ic];
    if (BWD) gg[j] = g_out[ic];
    if (MODE == 2) {   // the Adam moments are streamed: ask for them with the velocity, before the gather
      mm[j] = m3[ic];
      uu[j] = u3[ic];
    }
  }
  const float hz = 0.5f * (float)(Dfull - 1), hy = 0.5f * (float)(H - 1), hx = 0.5f * (float)(W - 1);
  const float nz1 = (float)(Dfull - 1), ny1 = (float)(H - 1), nx1 = (float)(W - 1);
  const unsigned uW = (unsigned)W, uHW = (unsigned)(H * W);
  const int f0 = min(first, n - 1);
  int w = f0 % W;
  const int t2 = f0 / W;
  int h = t2 % H, z = t2 / H + zoff;
  const int zlast = zoff + D - 1;
  F2u p[4][4];
  float wz[4], wy[4], wx[4], mz[4], my[4], mx[4];
  [[maybe_unused]] const int w0 = w, h0 = h, z0 = z;   // MODE 2 with adv_next: the walk is repeated for the second sample
  [[maybe_unused]] unsigned ob[4];                    // ... and the base cell of the first sample is remembered
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float xz = fmaf(-vv[j].x, hz, (float)z), xy = fmaf(-vv[j].y, hy, (float)h),
                xx = fmaf(-vv[j].z, hx, (float)w);
    const float cz = __builtin_amdgcn_fmed3f(xz, 0.f, nz1), cy = __builtin_amdgcn_fmed3f(xy, 0.f, ny1),
                cx = __builtin_amdgcn_fmed3f(xx, 0.f, nx1);
    const float bz = fminf(floorf(cz), nz1 - 1.f), by = fminf(floorf(cy), ny1 - 1.f), bx = fminf(floorf(cx), nx1 - 1.f);
    wz[j] = cz - bz; wy[j] = cy - by; wx[j] = cx - bx;
    if (BWD) {   // outside the volume both clipped corners coincide: no dependence on the coordinate
      mz[j] = (xz >= 0.f && xz < nz1) ? hz : 0.f;
      my[j] = (xy >= 0.f && xy < ny1) ? hy : 0.f;
      mx[j] = (xx >= 0.f && xx < nx1) ? hx : 0.f;
    }
    const unsigned o = (unsigned)(int)bz * uHW + (unsigned)(int)by * uW + (unsigned)(int)bx;
    if (MODE == 2) ob[j] = o;
    p[j][0] = *reinterpret_cast<const F2u*>(d + o);
    p[j][1] = *reinterpret_cast<const F2u*>(d + o + uW);
    p[j][2] = *reinterpret_cast<const F2u*>(d + o + uHW);
    p[j][3] = *reinterpret_cast<const F2u*>(d + o + uHW + uW);
    // next voxel of this lane: 64 further on (clamped at the end of the volume; those results are not stored)
    w += 64;
    while (w >= W) { w -= W; if (++h == H) { h = 0; if (z < zlast) ++z; } }
  }
  if (!BWD) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float a00 = fmaf(wx[j], p[j][0].y - p[j][0].x, p[j][0].x), a01 = fmaf(wx[j], p[j][1].y - p[j][1].x, p[j][1].x);
      const float a10 = fmaf(wx[j], p[j][2].y - p[j][2].x, p[j][2].x), a11 = fmaf(wx[j], p[j][3].y - p[j][3].x, p[j][3].x);
      const float b0 = fmaf(wy[j], a01 - a00, a00), b1 = fmaf(wy[j], a11 - a10, a10);
      if (ok[j]) out[first + 64 * j] = fmaf(wz[j], b1 - b0, b0);
      if constexpr (LIVE) {
        const unsigned long long lv = __ballot(ok[j] && corners_differ(p[j]));
        if (lane == 0 && first + 64 * j < n) ad.live[(first + 64 * j) >> 6] = lv;
      }
    }
  } else {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float e00 = p[j][0].y - p[j][0].x, e01 = p[j][1].y - p[j][1].x, e10 = p[j][2].y - p[j][2].x,
                  e11 = p[j][3].y - p[j][3].x;
      const float a00 = fmaf(wx[j], e00, p[j][0].x), a01 = fmaf(wx[j], e01, p[j][1].x);
      const float a10 = fmaf(wx[j], e10, p[j][2].x), a11 = fmaf(wx[j], e11, p[j][3].x);
      // d(sample)/d(coordinate) along each axis: difference of the two faces, interpolated in the other two
      const float f0 = fmaf(wy[j], e01 - e00, e00), f1 = fmaf(wy[j], e11 - e10, e10);
      const float dx = fmaf(wz[j], f1 - f0, f0);
      const float g0 = a01 - a00, g1 = a11 - a10;
      const float dy = fmaf(wz[j], g1 - g0, g0);
      const float b0 = fmaf(wy[j], g0, a00), b1 = fmaf(wy[j], g1, a10);
      const float dz = b1 - b0;
      // coordinate = index - vel * (n-1)/2  =>  d/dvel = -(n-1)/2 * d/dcoordinate (mz/my/mx carry the factor)
      const float gv[3] = {-gg[j] * dz * mz[j], -gg[j] * dy * my[j], -gg[j] * dx * mx[j]};
      const int idx = first + 64 * j;
      if (MODE == 1) {
        if (ok[j]) reinterpret_cast<F3u*>(out)[idx] = F3u{gv[0], gv[1], gv[2]};
      } else {
        // TF ApplyAdam, same arithmetic as adam_kernel (field.hip); out == vel
        float xs[3] = {vv[j].x, vv[j].y, vv[j].z}, ms[3] = {mm[j].x, mm[j].y, mm[j].z},
              us[3] = {uu[j].x, uu[j].y, uu[j].z};
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          ms[c] = ad.b1 * ms[c] + (1.f - ad.b1) * gv[c];
          us[c] = ad.b2 * us[c] + (1.f - ad.b2) * gv[c] * gv[c];
          xs[c] -= ad.lr_t * ms[c] / (sqrtf(us[c]) + ad.eps);
        }
        vv[j] = F3u{xs[0], xs[1], xs[2]};
        if constexpr (EVER) {
          const uint32_t o12 = act[j] ? (uint32_t)idx * 12u : 0x80000000u;
          st3(vv[j], vel_rs, o12);
          st3(F3u{ms[0], ms[1], ms[2]}, m_rs, o12);
          st3(F3u{us[0], us[1], us[2]}, u_rs, o12);
        } else if (ok[j]) {
          reinterpret_cast<F3u*>(out)[idx] = vv[j];
          reinterpret_cast<F3u*>(ad.m)[idx] = F3u{ms[0], ms[1], ms[2]};
          reinterpret_cast<F3u*>(ad.v)[idx] = F3u{us[0], us[1], us[2]};
        }
      }
    }
    if (MODE == 2 && ad.adv_next) {
      // the next iteration's forward sample from the updated velocity (the same lines as the MODE 0 body above; the
      // voxel indices are walked again rather than kept: twelve registers more cost the kernel a wave per SIMD)
      w = w0; h = h0; z = z0;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float xz = fmaf(-vv[j].x, hz, (float)z), xy = fmaf(-vv[j].y, hy, (float)h), xx = fmaf(-vv[j].z, hx, (float)w);
        const float cz = __builtin_amdgcn_fmed3f(xz, 0.f, nz1), cy = __builtin_amdgcn_fmed3f(xy, 0.f, ny1),
                    cx = __builtin_amdgcn_fmed3f(xx, 0.f, nx1);
        const float bz = fminf(floorf(cz), nz1 - 1.f), by = fminf(floorf(cy), ny1 - 1.f), bx = fminf(floorf(cx), nx1 - 1.f);
        wz[j] = cz - bz; wy[j] = cy - by; wx[j] = cx - bx;
        const unsigned o = (unsigned)(int)bz * uHW + (unsigned)(int)by * uW + (unsigned)(int)bx;
        // One Adam step moves the back-traced point by a fraction of a cell: its base cell is almost always the one the
        // adjoint just gathered, whose eight corners are still in registers -- only the weights change.  The second
        // gather (a further memory round trip in every wave's life: 0.128 -> 0.166 ms when it was unconditional) is
        // taken only by the lanes whose point crossed a cell face.
        if (o != ob[j]) {
          p[j][0] = *reinterpret_cast<const F2u*>(d + o);
          p[j][1] = *reinterpret_cast<const F2u*>(d + o + uW);
          p[j][2] = *reinterpret_cast<const F2u*>(d + o + uHW);
          p[j][3] = *reinterpret_cast<const F2u*>(d + o + uHW + uW);
        }
        w += 64;
        while (w >= W) { w -= W; if (++h == H) { h = 0; if (z < zlast) ++z; } }
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float a00 = fmaf(wx[j], p[j][0].y - p[j][0].x, p[j][0].x), a01 = fmaf(wx[j], p[j][1].y - p[j][1].x, p[j][1].x);
        const float a10 = fmaf(wx[j], p[j][2].y - p[j][2].x, p[j][2].x), a11 = fmaf(wx[j], p[j][3].y - p[j][3].x, p[j][3].x);
        const float b0 = fmaf(wy[j], a01 - a00, a00), b1 = fmaf(wy[j], a11 - a10, a10);
        if constexpr (EVER)
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, fmaf(wz[j], b1 - b0, b0)), adv_rs,
                                                act[j] ? (uint32_t)(first + 64 * j) * 4u : 0x80000000u, 0, 0);
        else if (ok[j]) ad.adv_next[first + 64 * j] = fmaf(wz[j], b1 - b0, b0);
        if constexpr (LIVE) {
          const unsigned long long lv = __ballot(ok[j] && (!EVER || act[j]) && corners_differ(p[j]));
          if (lane == 0 && first + 64 * j < n) ad.live[(first + 64 * j) >> 6] = lv;
        }
      }
    }
  }
}


// ---- one step of _transport (styler_base.py:59-89) fused with the temporal filter's accumulation -------------------
// out = w_g * advect(g, scale * u) + w_add * addend   for a C-channel grid field g [D,H,W,C] (C = 3: the stylisation
// velocity, C = 1: a density), u [D,H,W,3] the simulation velocity of the frame being crossed (scale = +1 forwards,
// -1 backwards, +-|b-a| for the one-step form).  HBM-bound: reads u (12 B), gathers g (4C B compulsory), reads the
// addend (4C B), writes out (4C B) per voxel.  A wave owns 128 consecutive voxels, lane l takes l and l+64 (every
// streamed 12-byte vector access is one contiguous run per instruction); lean stencil as in advect1_kernel; the two
// W-adjacent corners of a row come from one 8C-byte access (dword-aligned dwordx2 / two dwordx3).
template <int C> struct __attribute__((packed, aligned(4))) Vec { float v[C]; };

template <int C>
__global__ void __launch_bounds__(256) transport_step_kernel(const float* __restrict__ g, const float* __restrict__ u,
                                                             const float* __restrict__ addend, float* __restrict__ out,
                                                             int D, int H, int W, float scale, float w_g, float w_add) {
  constexpr int NV = 2;
  const int n = D * H * W;
  const int lane = threadIdx.x & 63;
  const unsigned per_xcd = gridDim.x / 8;
  const unsigned lb = (blockIdx.x % 8) * per_xcd + blockIdx.x / 8;     // contiguous z-slab per XCD (shared planes)
  const int first = (lb * blockDim.x + (threadIdx.x - lane)) * NV + lane;
  if (first - lane >= n) return;
  const F3u* u3 = reinterpret_cast<const F3u*>(u);
  const Vec<C>* gC = reinterpret_cast<const Vec<C>*>(g);
  F3u vv[NV];
  Vec<C> add[NV];
  bool ok[NV];
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    const int idx = first + 64 * j;
    ok[j] = idx < n;
    const int ic = ok[j] ? idx : n - 1;
    vv[j] = u3[ic];
    if (addend) add[j] = reinterpret_cast<const Vec<C>*>(addend)[ic];
  }
  const float hz = 0.5f * (float)(D - 1) * scale, hy = 0.5f * (float)(H - 1) * scale, hx = 0.5f * (float)(W - 1) * scale;
  const float nz1 = (float)(D - 1), ny1 = (float)(H - 1), nx1 = (float)(W - 1);
  const unsigned uW = (unsigned)W, uHW = (unsigned)(H * W);
  const int f0 = min(first, n - 1);
  int w = f0 % W;
  const int t2 = f0 / W;
  int h = t2 % H, z = t2 / H;
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    const float xz = fmaf(-vv[j].x, hz, (float)z), xy = fmaf(-vv[j].y, hy, (float)h), xx = fmaf(-vv[j].z, hx, (float)w);
    const float cz = __builtin_amdgcn_fmed3f(xz, 0.f, nz1), cy = __builtin_amdgcn_fmed3f(xy, 0.f, ny1),
                cx = __builtin_amdgcn_fmed3f(xx, 0.f, nx1);
    const float bz = fminf(floorf(cz), nz1 - 1.f), by = fminf(floorf(cy), ny1 - 1.f), bx = fminf(floorf(cx), nx1 - 1.f);
    const float wz = cz - bz, wy = cy - by, wx = cx - bx;
    const unsigned o = (unsigned)(int)bz * uHW + (unsigned)(int)by * uW + (unsigned)(int)bx;
    const Vec<C> p00a = gC[o], p00b = gC[o + 1], p01a = gC[o + uW], p01b = gC[o + uW + 1];
    const Vec<C> p10a = gC[o + uHW], p10b = gC[o + uHW + 1], p11a = gC[o + uHW + uW], p11b = gC[o + uHW + uW + 1];
    Vec<C> r;
#pragma unroll
    for (int c = 0; c < C; ++c) {
      const float a00 = fmaf(wx, p00b.v[c] - p00a.v[c], p00a.v[c]), a01 = fmaf(wx, p01b.v[c] - p01a.v[c], p01a.v[c]);
      const float a10 = fmaf(wx, p10b.v[c] - p10a.v[c], p10a.v[c]), a11 = fmaf(wx, p11b.v[c] - p11a.v[c], p11a.v[c]);
      const float b0 = fmaf(wy, a01 - a00, a00), b1 = fmaf(wy, a11 - a10, a10);
      float s = w_g * fmaf(wz, b1 - b0, b0);
      if (addend) s = fmaf(w_add, add[j].v[c], s);
      r.v[c] = s;
    }
    if (ok[j]) reinterpret_cast<Vec<C>*>(out)[first + 64 * j] = r;
    w += 64;
    while (w >= W) { w -= W; if (++h == H) { h = 0; if (z < D - 1) ++z; } }
  }
}

// generic channel count (and degenerate volumes): one thread per voxel on the exact reference stencil
__global__ void __launch_bounds__(256) transport_step_generic_kernel(const float* __restrict__ g,
                                                                     const float* __restrict__ u,
                                                                     const float* __restrict__ addend,
                                                                     float* __restrict__ out, int D, int H, int W, int C,
                                                                     float scale, float w_g, float w_add) {
  const int64_t n = (int64_t)D * H * W;
  const int64_t vox = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (vox >= n) return;
  const int x = (int)(vox % W), y = (int)((vox / W) % H), z = (int)(vox / ((int64_t)W * H));
  const float* v = u + vox * 3;
  Tri t; Axis az, ay, ax;
  tri_setup(lin_coord(z, D) - scale * v[0], lin_coord(y, H) - scale * v[1], lin_coord(x, W) - scale * v[2], D, H, W, t,
            az, ay, ax);
  for (int c = 0; c < C; ++c) {
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) s += t.w[k] * g[t.o[k] * C + c];
    s *= w_g;
    if (addend) s = fmaf(w_add, addend[vox * C + c], s);
    out[vox * C + c] = s;
  }
}

// ---- output-stationary adjoint of rotate for C = 1 -------------------------------------------
// Global float atomics cap the scatter at ~90 G atomics/s (5.6 ms for 8 views of 200^3), and LDS float
// atomics are no better: measured on gfx950 (tools/lds_atomic_bench.hip) ds_add_f32 sustains 0.33
// lanes/clk/CU while ds_add_u64 runs at 9.4.  So the tile accumulates in 64-bit FIXED POINT:
// contributions are scaled by 2^k, with k chosen from max|g_out| (a streaming pre-pass) so that no voxel
// sum can overflow; with ~49 bits below the largest value this is more accurate than f32 accumulation and,
// integer adds being associative, bit-reproducible.
// A block owns RT_TZ x RT_TY x RT_TX voxels of g_d and keeps them, plus a one-cell halo on every side, in
// LDS as 64-bit fixed-point accumulators (LDS float atomics run at 1/40 of the integer rate on gfx950).
// It walks, view by view, the output-lattice rows that cross its catchment (the samples whose base cell
// lies in [tile_lo - 1, tile_hi]), each row clipped to the exact x-interval; a sample in the catchment adds
// all 8 corners with compile-time LDS offsets and no per-corner tests (the halo absorbs the corners that
// belong to neighbouring tiles; only the owned cells are written back).  No global atomics; the integer
// sums make the result independent of the traversal order.
#ifndef NFS_RT_TZ
// (tile shape x lanes per lattice row swept on 8 views of 200^3, tools/rot_tile_sweep.sh: 14x12x40 x 8 0.331 ms, x 4 0.305,
// x 2 0.357, x 16 0.42; 14x14x34 x 4 0.296; 16x12x34 x 4 0.307; 10x10x40 x 4 0.324; 12x12x48 x 4 0.41 (240-wide for 200);
// the lattice rows of one (tile, view) are ~20 samples long on average and many are much shorter, so narrow groups
// idle less in the row tails)
#define NFS_RT_TZ 14
#define NFS_RT_TY 14
#define NFS_RT_TX 34
#define NFS_RT_GROUP 4
#endif
constexpr int RT_TZ = NFS_RT_TZ, RT_TY = NFS_RT_TY, RT_TX = NFS_RT_TX;
constexpr int RT_LZ = RT_TZ + 2, RT_LY = RT_TY + 2, RT_LX = RT_TX + 2;
#ifndef NFS_RT_THREADS
#define NFS_RT_THREADS 1024
#endif
constexpr int RT_THREADS = NFS_RT_THREADS;
constexpr int RT_GROUP = NFS_RT_GROUP;   // lanes that share one lattice row
constexpr int RT_VMAX = 32;   // views per launch (host loops over chunks)

struct ViewRows {             // per (block, view), in LDS
  // voxel-space sample coordinate of axis a at lattice point (oz,oy,ox): c + a0*oz + a1*oy + s*ox
  float a0[3], a1[3], s[3], inv_s[3], c[3];
  float lo[3], hi[3];         // catchment (padded) in voxel space
  int z_lo, y_lo, x_lo, x_hi, ey, rows;
  unsigned my;                // ceil(2^32 / ey)
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

// catchment of one (tile, view).  One lane per view runs this while the other 1023 threads of the block wait, so its
// latency is a fixed cost of every block (measured in round 4, tools/render_family_by_views.py: the kernel took
// 42 us + 30 us per view at 200^3 -- the first version of this function, 21 double-precision divisions and 148 bytes of
// scratch under the kernel's 64-register cap, was most of the 42).  Now: double precision where it sets the floats the
// sample loop reads (A, c, 1 / A[a][2]: bit-identical to before), one reciprocal of the determinant for the lattice box
// (a bound, padded by 0.05 cells), every array fully unrolled and the outputs stored as soon as they are final: inlined
// it fits the 64 registers without scratch (out of line the ABI's callee-saved registers alone cost 16 scratch round
// trips).  200^3: 71.8 -> 62.6 us at one view, 278.7 -> 271.5 at eight, bit-identical; with the sample loop compiled out
// the kernel (zero fill of the accumulators, this function, write-back) takes 16.7 us, of which this function 1.3.
__device__ __forceinline__ void rotate_tile_catchment(const float* r, int D, int H, int W, int z0, int y0,
                                                                 int x0, int z1, int y1, int x1, ViewRows* out) {
  const int n[3] = {D, H, W};
  const int tlo[3] = {z0, y0, x0}, thi[3] = {z1, y1, x1};
  float rr[9];
#pragma unroll
  for (int i = 0; i < 9; ++i) rr[i] = r[i];
  double sb[3];
#pragma unroll
  for (int b = 0; b < 3; ++b) sb[b] = n[b] > 1 ? 2.0 / (n[b] - 1) : 0.0;
  // cmin/cmax: the range the (unclamped) sample coordinate of axis a takes over the whole output lattice;
  // a border tile also catches the samples clamped onto it, so its catchment extends to that range
  double A[3][3], mid[3], half[3];
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const double ha = 0.5 * (n[a] - 1);
    double rs = 0.0, mn = 0.0, mx = 0.0;
#pragma unroll
    for (int b = 0; b < 3; ++b) {
      A[a][b] = (double)rr[a * 3 + b] * sb[b] * ha;
      rs += (double)rr[a * 3 + b];
      mn += fmin(A[a][b] * (n[b] - 1), 0.0);
      mx += fmax(A[a][b] * (n[b] - 1), 0.0);
    }
    const double c = (1.0 - rs) * ha;
    // base cell floor(u) in [tlo-1, thi]  <=>  u in [tlo-1, thi+1); 0.05-cell pad covers the float evaluation
    const double xl = (tlo[a] == 0) ? fmin(c + mn - 0.5, -1.05) : tlo[a] - 1.05;
    const double xh = (thi[a] == n[a] - 1) ? fmax(c + mx + 0.5, n[a] + 0.05) : thi[a] + 1.05;
    mid[a] = 0.5 * (xl + xh) - c;
    half[a] = 0.5 * (xh - xl);
    out->a0[a] = (float)A[a][0];
    out->a1[a] = (float)A[a][1];
    out->s[a] = (float)A[a][2];
    out->inv_s[a] = fabs(A[a][2]) > 1e-6 ? (float)(1.0 / A[a][2]) : 0.f;
    out->c[a] = (float)c;
    out->lo[a] = (float)xl;
    out->hi[a] = (float)xh;
  }
  const double k00 = A[1][1] * A[2][2] - A[1][2] * A[2][1], k01 = A[0][2] * A[2][1] - A[0][1] * A[2][2],
               k02 = A[0][1] * A[1][2] - A[0][2] * A[1][1];
  const double det = A[0][0] * k00 + A[1][0] * k01 + A[2][0] * k02;
  int lo[3] = {0, 0, 0}, hi[3] = {D - 1, H - 1, W - 1};
  if (fabs(det) > 1e-9) {
    const double rd = 1.0 / det;
    // row b of the inverse = cofactors / det; box centre / half widths in voxel space -> centre / half extents in
    // lattice space (the reciprocal instead of nine divisions moves the box by ~1e-13 cells: it is a padded bound)
    double iv[3][3];
    iv[0][0] = k00; iv[0][1] = k01; iv[0][2] = k02;
    iv[1][0] = A[1][2] * A[2][0] - A[1][0] * A[2][2];
    iv[1][1] = A[0][0] * A[2][2] - A[0][2] * A[2][0];
    iv[1][2] = A[0][2] * A[1][0] - A[0][0] * A[1][2];
    iv[2][0] = A[1][0] * A[2][1] - A[1][1] * A[2][0];
    iv[2][1] = A[0][1] * A[2][0] - A[0][0] * A[2][1];
    iv[2][2] = A[0][0] * A[1][1] - A[0][1] * A[1][0];
#pragma unroll
    for (int b = 0; b < 3; ++b) {
      const double oc = (iv[b][0] * mid[0] + iv[b][1] * mid[1] + iv[b][2] * mid[2]) * rd;
      const double oe = (fabs(iv[b][0]) * half[0] + fabs(iv[b][1]) * half[1] + fabs(iv[b][2]) * half[2]) * fabs(rd) + 1e-9;
      lo[b] = (int)fmin(fmax(floor(oc - oe), 0.0), (double)n[b]);
      hi[b] = (int)fmax(fmin(ceil(oc + oe), (double)(n[b] - 1)), -1.0);
    }
  }
  const int ez = max(hi[0] - lo[0] + 1, 0), ey = max(hi[1] - lo[1] + 1, 0);
  out->z_lo = lo[0]; out->y_lo = lo[1]; out->x_lo = lo[2]; out->x_hi = hi[2];
  out->ey = ey;
  out->rows = hi[2] >= lo[2] ? ez * ey : 0;
  out->my = ey > 1 ? 0xFFFFFFFFu / (unsigned)ey + 1u : 0u;     // ceil(2^32 / ey), ey >= 2
}

// ---- which tiles, and how much of each, a live-mask launch accumulates ------------------------------------------------
// live (nfs_advect_*_live): bit (z H + y) W + x set = the velocity gradient of voxel (z, y, x) can be non-zero.  The
// linear stencil between the rotate adjoint's g_d and the advect adjoint (the smoothing) reads g_d within `dil` cells of
// such a voxel, and nothing else of g_d is ever multiplied by a non-zero: a tile needs the sums of the bounding box of
// its voxels within `dil` of a live voxel, no more.  One wave per tile finds that box; a second, one-block launch sorts
// the tiles by decreasing work into `order` -- the hardware starts blocks in index order, and with boxes of very
// different sizes (and two fifths of the tiles empty on a smoke density) the long ones must not start last.
// Workspace (ints): [LWS_BOXES + 8 t ...] box of tile t = (z0, z1, y0, y1, x0, x1, work, -), [LWS_ORDER(ntiles) + i] the
// tile block i of the adjoint takes.
constexpr int LWS_BOXES = 8;
#define LWS_ORDER(ntiles) (nfs::LWS_BOXES + 8 * (ntiles))
constexpr int LB_THREADS = 1024, LB_WAVES = LB_THREADS / 64, LB_CLASSES = 64;

// a WAVE per tile, four (z, y) rows of the tile's dilated box per lane
__global__ void __launch_bounds__(LB_THREADS) rotate_live_boxes_kernel(const unsigned long long* __restrict__ live, int D,
                                                                       int H, int W, int tiles_y, int tiles_x, int dil,
                                                                       int ntiles, int* __restrict__ ws) {
  const int t = threadIdx.x, lane = t & 63;
  const int tile = blockIdx.x * LB_WAVES + (t >> 6);
  if (tile < ntiles) {
    const int bx = tile % tiles_x, by = (tile / tiles_x) % tiles_y, bz = tile / (tiles_x * tiles_y);
    const int Tz0 = bz * RT_TZ, Ty0 = by * RT_TY, Tx0 = bx * RT_TX;
    const int Tz1 = min(Tz0 + RT_TZ, D) - 1, Ty1 = min(Ty0 + RT_TY, H) - 1, Tx1 = min(Tx0 + RT_TX, W) - 1;
    const int ez0 = max(Tz0 - dil, 0), ez1 = min(Tz1 + dil, D - 1), ey0 = max(Ty0 - dil, 0), ey1 = min(Ty1 + dil, H - 1);
    const int ex0 = max(Tx0 - dil, 0), ex1 = min(Tx1 + dil, W - 1);
    const int nry = ey1 - ey0 + 1, len = ex1 - ex0 + 1, nrows = (ez1 - ez0 + 1) * nry;   // len <= RT_TX + 2 dil <= 63
    int b0 = 0x7fffffff, b1 = -1, b2 = 0x7fffffff, b3 = -1, b4 = 0x7fffffff, b5 = -1;
    for (int r = lane; r < nrows; r += 64) {
      const int rz = ez0 + r / nry, ry = ey0 + r % nry;
      const int64_t bit0 = ((int64_t)rz * H + ry) * W + ex0;
      const int sh = (int)(bit0 & 63);
      unsigned long long bits = live[bit0 >> 6] >> sh;
      if (sh + len > 64) bits |= live[(bit0 >> 6) + 1] << (64 - sh);
      bits &= (1ull << len) - 1ull;
      if (bits) {
        b0 = min(b0, rz); b1 = max(b1, rz); b2 = min(b2, ry); b3 = max(b3, ry);
        b4 = min(b4, ex0 + (int)__builtin_ctzll(bits)); b5 = max(b5, ex0 + 63 - (int)__builtin_clzll(bits));
      }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      b0 = min(b0, __shfl_xor(b0, o, 64)); b1 = max(b1, __shfl_xor(b1, o, 64));
      b2 = min(b2, __shfl_xor(b2, o, 64)); b3 = max(b3, __shfl_xor(b3, o, 64));
      b4 = min(b4, __shfl_xor(b4, o, 64)); b5 = max(b5, __shfl_xor(b5, o, 64));
    }
    if (lane == 0) {
      int* b = ws + LWS_BOXES + tile * 8;
      int work = 0;
      int z0 = Tz0, z1 = Tz1, y0 = Ty0, y1 = Ty1, x0 = Tx0, x1 = Tx1;
      if (b1 >= 0) {
        z0 = max(Tz0, b0 - dil); z1 = min(Tz1, b1 + dil);
        y0 = max(Ty0, b2 - dil); y1 = min(Ty1, b3 + dil);
        x0 = max(Tx0, b4 - dil); x1 = min(Tx1, b5 + dil);
        // samples whose base cell lies in [lo - 1, hi] per axis; a box on the volume's border also catches the samples
        // clamped onto it (a few cells' worth at these view angles)
        const int cz = z1 - z0 + 2 + (z0 == 0 ? 6 : 0) + (z1 == D - 1 ? 6 : 0);
        const int cy = y1 - y0 + 2 + (y0 == 0 ? 6 : 0) + (y1 == H - 1 ? 6 : 0);
        const int cx = x1 - x0 + 2 + (x0 == 0 ? 6 : 0) + (x1 == W - 1 ? 6 : 0);
        work = cz * cy * cx;
      }
      b[0] = z0; b[1] = z1; b[2] = y0; b[3] = y1; b[4] = x0; b[5] = x1; b[6] = work; b[7] = 0;
    }
  }
}

// ... and the tiles by decreasing work: one block, counting sort by work class (a launch of its own: the boxes come from
// 85 blocks on all XCDs, and a ticket + __threadfence() in their kernel -- the "last block sorts" form this replaced --
// cost 30 us of cache maintenance; a kernel boundary costs 3)
__global__ void __launch_bounds__(LB_THREADS) rotate_live_order_kernel(int ntiles, int* __restrict__ ws) {
  __shared__ int hist[LB_CLASSES + 1];
  const int t = threadIdx.x;
  const int* boxes = ws + LWS_BOXES;
  constexpr int WMAX = (RT_TZ + 13) * (RT_TY + 13) * (RT_TX + 13);
  for (int i = t; i <= LB_CLASSES; i += LB_THREADS) hist[i] = 0;
  __syncthreads();
  auto cls = [](int work) {   // 0 = most work ... LB_CLASSES - 1 = least, LB_CLASSES = none
    return work == 0 ? LB_CLASSES : LB_CLASSES - 1 - min((int)(((int64_t)work * LB_CLASSES) / (WMAX + 1)), LB_CLASSES - 1);
  };
  constexpr int PER = 4;                                   // tiles per thread and round (4096 per round)
  int* order = ws + LWS_ORDER(ntiles);
  int done = 0;                                            // tiles placed by earlier rounds (volumes of > 4096 tiles)
  for (int base = 0; base < ntiles; base += PER * LB_THREADS) {
    int c[PER];
#pragma unroll
    for (int j = 0; j < PER; ++j) {
      const int i = base + j * LB_THREADS + t;
      c[j] = i < ntiles ? cls(boxes[i * 8 + 6]) : -1;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < PER; ++j) if (c[j] >= 0) atomicAdd(&hist[c[j]], 1);
    __syncthreads();
    if (t == 0) {
      int run = done;
      for (int k = 0; k <= LB_CLASSES; ++k) { const int n = hist[k]; hist[k] = run; run += n; }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < PER; ++j) if (c[j] >= 0) order[atomicAdd(&hist[c[j]], 1)] = base + j * LB_THREADS + t;
    __syncthreads();
    done = min(base + PER * LB_THREADS, ntiles);
    for (int i = t; i <= LB_CLASSES; i += LB_THREADS) hist[i] = 0;
  }
}

// COEF: g_out holds, per sample, the render's u (nfs_rotate_render_fwd with u_rot) and `ab` the per-(view, depth
// segment, ray) pair (A, B) of nfs_render_ray_coef: the sample's gradient is A u - B, formed here instead of by a
// render-adjoint pass over the whole rotated volume (K4a: 512 MB of traffic and a launch at eight views)
template <bool COEF>
__global__ void __launch_bounds__(RT_THREADS, 8) rotate_bwd_tiled_kernel(const float* __restrict__ g_out,
                                                                         const float* __restrict__ rot,
                                                                         float* __restrict__ g_d,
                                                                         const unsigned* __restrict__ gmax_bits,
                                                                         float bound_factor, int V, int D, int H,
                                                                         int W, int tiles_y, int tiles_x, int overwrite,
                                                                         int order, const float2* __restrict__ ab,
                                                                         int nseg, int seg_len, int nbounds,
                                                                         const int* __restrict__ lws) {
  __shared__ unsigned long long acc[RT_LZ * RT_LY * RT_LX];
  __shared__ ViewRows vrows[RT_VMAX];
  [[maybe_unused]] __shared__ float gred[RT_THREADS / 64];
  const int t = threadIdx.x;
  // Which tile a block takes.  order 0 (round 1-3): x-border tiles first, then z-border, then the interior (border tiles
  // also collect the clamped out-of-volume samples and run longest; starting them first keeps them off the tail), dealt
  // round-robin to the XCDs by the hardware -- neighbouring tiles, whose catchments overlap by a third, then sit on
  // different L2s and every XCD fetches its own copy of the shared rows (PMC: 608 MB fetched for 320 MB algorithmic).
  // order 1 / 2 (NFS_RT_XCD; measured in round 4, tools/rot_xcd_ab.sh, NOT the default): every XCD owns a contiguous range
  // of (z, y) tile columns (1: z-major, 2: y-major) with all their x tiles, x-border tiles first inside the XCD, so that
  // shared catchment rows are L2 hits: FETCH_SIZE 340 -> 308 / 290 MB (raw counter) but 0.297 -> 0.308 ms -- the kernel is
  // bound by its VALU + LDS-atomic stream, not by the fetch, and the round-robin deal balances the slow border tiles better.
  int bz, by, bx;
  const int* lbox = nullptr;
  if (lws) {
    // live-mask launch: rotate_live_boxes_kernel has left, per tile, the box to accumulate and, in `order`, the tiles by
    // decreasing work -- the longest blocks start first, the tiles nothing reads come last and return at once
    const int ntiles = gridDim.x;
    const int tile = lws[LWS_ORDER(ntiles) + blockIdx.x];
    lbox = lws + LWS_BOXES + tile * 8;
    bx = tile % tiles_x;
    by = (tile / tiles_x) % tiles_y;
    bz = tile / (tiles_x * tiles_y);
  } else if (order == 0) {
    const int tiles_z = gridDim.x / (tiles_x * tiles_y);
    by = blockIdx.x % tiles_y;
    const int oz_ = (blockIdx.x / tiles_y) % tiles_z;
    const int ox_ = blockIdx.x / (tiles_y * tiles_z);
    bz = oz_ == 0 ? 0 : (oz_ == 1 ? tiles_z - 1 : oz_ - 1);
    bx = ox_ == 0 ? 0 : (ox_ == 1 ? tiles_x - 1 : ox_ - 1);
  } else {
    const int tiles_z = (D + RT_TZ - 1) / RT_TZ;
    const int cols = tiles_z * tiles_y, cpx = (cols + 7) / 8;
    const int xcd = blockIdx.x % 8, i = blockIdx.x / 8;
    const int c0 = xcd * cpx, nc = min(cpx, cols - c0);
    if (nc <= 0 || i >= nc * tiles_x) return;
    const int ox_ = i / nc, col = c0 + i % nc;
    bx = ox_ == 0 ? 0 : (ox_ == 1 ? tiles_x - 1 : ox_ - 1);
    if (order == 1) { bz = col / tiles_y; by = col % tiles_y; } else { by = col / tiles_z; bz = col % tiles_z; }
  }
  // the tile (T*) and the box of it this block accumulates (z0 ... x1, inclusive): the whole tile, or -- with a live
  // mask -- the bounding box of the voxels whose gradient anything downstream reads.  What is written outside that box
  // (zeros) meets an exact zero factor downstream; inside it every sample that touches a voxel is still visited and the
  // integer sums do not depend on the order: the velocity gradient is bit-identical to the unmasked launch
  // (tests/test_dead_skip_gpu.py).
  const int Tz0 = bz * RT_TZ, Ty0 = by * RT_TY, Tx0 = bx * RT_TX;
  const int Tz1 = min(Tz0 + RT_TZ, D) - 1, Ty1 = min(Ty0 + RT_TY, H) - 1, Tx1 = min(Tx0 + RT_TX, W) - 1;
  int z0 = Tz0, y0 = Ty0, x0 = Tx0, z1 = Tz1, y1 = Ty1, x1 = Tx1;
  bool dead = false;
  if (lbox) {
    z0 = lbox[0]; z1 = lbox[1]; y0 = lbox[2]; y1 = lbox[3]; x0 = lbox[4]; x1 = lbox[5];   // (scalar loads: uniform address)
    dead = lbox[6] == 0;
  }
  if (!dead)
    for (int i = t; i < RT_LZ * RT_LY * RT_LX; i += RT_THREADS) acc[i] = 0ull;
  // fixed-point scale 2^k: |any voxel sum| <= bound_factor * max|g_out| must stay below 2^62
  float gmax = 0.f;
  if (dead) {                                     // (uniform over the block)
  } else if (COEF) {                              // the maximum of the coefficient kernel's per-block bounds
    const float* bounds = reinterpret_cast<const float*>(gmax_bits);
    float m = 0.f;
    for (int i = t; i < nbounds; i += RT_THREADS) m = fmaxf(m, bounds[i]);
    m = wave_max(m);
    if ((t & 63) == 0) gred[t >> 6] = m;
    __syncthreads();
    gmax = gred[0];
#pragma unroll
    for (int i = 1; i < RT_THREADS / 64; ++i) gmax = fmaxf(gmax, gred[i]);
  } else {
    gmax = __uint_as_float(*gmax_bits);
  }
  if (dead || !(gmax > 0.f)) {                    // nothing downstream reads this tile / all-zero gradient
    if (overwrite)
      for (int i = t; i < RT_TZ * RT_TY * RT_TX; i += RT_THREADS) {
        const int lx_ = i % RT_TX, ly_ = (i / RT_TX) % RT_TY, lz_ = i / (RT_TX * RT_TY);
        const int z = Tz0 + lz_, y = Ty0 + ly_, x = Tx0 + lx_;
        if (z < D && y < H && x < W) g_d[((int64_t)z * H + y) * W + x] = 0.f;
      }
    return;
  }
  int ebound;
  frexpf(gmax * bound_factor, &ebound);           // gmax*bound_factor < 2^ebound
  const int kexp = 62 - ebound;
  // samples are scaled by 2^(k-32): the integer part of a product is the high word, the fraction the low word
  const int ks = kexp - 32;
  const float fscale = ldexpf(1.f, min(max(ks, -120), 120));
  const float fscale2 = ldexpf(1.f, ks - min(max(ks, -120), 120));  // split: 2^ks may exceed the float range

  if (t < V) rotate_tile_catchment(rot + t * 9, D, H, W, z0, y0, x0, z1, y1, x1, &vrows[t]);
  __syncthreads();

  const int grp = t / RT_GROUP, gl = t % RT_GROUP;
  constexpr int NGRP = RT_THREADS / RT_GROUP;
  const unsigned nz = (unsigned)(z1 - z0 + 1), ny = (unsigned)(y1 - y0 + 1), nx = (unsigned)(x1 - x0 + 1);
  const float3 dm1 = make_float3((float)(D - 1), (float)(H - 1), (float)(W - 1));
  const float oz1 = (float)(1 - z0), oy1 = (float)(1 - y0), ox1 = (float)(1 - x0);
  for (int v = 0; v < V; ++v) {
    const ViewRows& vr = vrows[v];
    const int rows = vr.rows;
    if (rows <= 0) continue;
    const float* gv = g_out + (int64_t)v * D * H * W;
    for (int row = grp; row < rows; row += NGRP) {
      const int zi = vr.ey > 1 ? (int)__umulhi((unsigned)row, vr.my) : row;
      const int oy = vr.y_lo + (row - zi * vr.ey), oz = vr.z_lo + zi;
      // exact x-interval of this row inside the catchment slab of every axis
      float ta = (float)vr.x_lo, tb = (float)vr.x_hi;
      const float fz = (float)oz, fy = (float)oy;
      float pv[3];                 // voxel coordinate of the row's sample ox = 0, per axis
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        const float p = vr.c[a] + vr.a0[a] * fz + vr.a1[a] * fy;
        pv[a] = p;
        if (vr.inv_s[a] != 0.f) {
          const float u0 = (vr.lo[a] - p) * vr.inv_s[a], u1 = (vr.hi[a] - p) * vr.inv_s[a];
          ta = fmaxf(ta, fminf(u0, u1) - 0.01f);
          tb = fminf(tb, fmaxf(u0, u1) + 0.01f);
        } else if (p < vr.lo[a] || p > vr.hi[a]) {
          tb = ta - 2.f;
        }
      }
      const int xa = (int)ceilf(ta), xb = (int)floorf(tb);
      if (xb < xa) continue;
      float sz = vr.s[0], sy = vr.s[1], sx = vr.s[2];
      // COEF: settle the three LDS reads HERE -- left to the compiler, their first use inside the sample loop puts an
      // s_waitcnt lgkmcnt(0) into every iteration, which also waits for the previous sample's eight LDS atomics.
      // Measured on one box (8 views of 200^3): COEF 324 -> 311 us with it, the plain instance 271 -> 278: that one keeps
      // the compiler's waits.
      if constexpr (COEF) asm volatile("" : "+v"(sz), "+v"(sy), "+v"(sx));
      const float* grow = gv + ((int64_t)oz * H + oy) * W;
      // (COEF: the ray of a sample is its (oy, ox); its depth segment, far end first, is wave-uniform per row)
      [[maybe_unused]] const float2* abrow =
          COEF ? ab + (((int64_t)v * nseg + (D - 1 - oz) / seg_len) * H + oy) * W : nullptr;
      // the next sample's operands are loaded one iteration ahead and stay RAW until they are used (forming A u - B at
      // load time would put a full wait behind every load)
      int ox = xa + gl;
      float g = 0.f;
      [[maybe_unused]] float2 cab = make_float2(0.f, 0.f);
      if (ox <= xb) {
        g = grow[ox];
        if (COEF) cab = abrow[ox];
      }
      while (ox <= xb) {
        const int oxn = ox + RT_GROUP;
        float gn = 0.f;
        [[maybe_unused]] float2 cabn = make_float2(0.f, 0.f);
        if (oxn <= xb) {                       // (one exec-mask block around both loads)
          gn = grow[oxn];
          if (COEF) cabn = abrow[oxn];
        }
        if (COEF) g = fmaf(cab.x, g, -cab.y);
        if (g != 0.f) {
          // lean stencil (same form as the ray march): the sample's voxel coordinate is affine in ox; border
          // replication = clamp the coordinate; base cell floor(c), weights (1-w, w).  A sample clamped onto the far
          // border has base cell n-1 and w = 0: the whole weight on voxel n-1 and an exact zero into the halo cell
          // beyond it (adding zero changes nothing in the fixed-point sum).
          const float fx = (float)ox;
          const float cz = __builtin_amdgcn_fmed3f(fmaf(sz, fx, pv[0]), 0.f, dm1.x);
          const float cy = __builtin_amdgcn_fmed3f(fmaf(sy, fx, pv[1]), 0.f, dm1.y);
          const float cx = __builtin_amdgcn_fmed3f(fmaf(sx, fx, pv[2]), 0.f, dm1.z);
          const float bz = floorf(cz), by_ = floorf(cy), bx_ = floorf(cx);
          const float wz = cz - bz, wy = cy - by_, wx = cx - bx_;
          // halo'd local base cell (small integers: exact in float)
          const int lz = (int)(bz + oz1), ly = (int)(by_ + oy1), lx = (int)(bx_ + ox1);
          if ((unsigned)lz <= nz && (unsigned)ly <= ny && (unsigned)lx <= nx) {
            const float gs = g * fscale * fscale2;
            const float wz1 = gs * wz, wz0 = gs - wz1;
            // packed f32 (v_pk_*_f32, two results per instruction): (w01, w11) = (wz0, wz1) * wy; (w00, w10) = rest
            const f32x2 wzp = {wz0, wz1};
            const f32x2 wy1 = wzp * wy, wy0 = wzp - wy1;
            const f32x2 ax = {1.f - wx, wx};
            f32x2 c00, c01, c10, c11;      // (x, x+1) corner pairs of the rows (z,y), (z,y+1), (z+1,y), (z+1,y+1)
            asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,1]" : "=v"(c00) : "v"(wy0), "v"(ax));
            asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,1]" : "=v"(c01) : "v"(wy1), "v"(ax));
            asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[1,1]" : "=v"(c10) : "v"(wy0), "v"(ax));
            asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[1,1]" : "=v"(c11) : "v"(wy1), "v"(ax));
            // LDS byte address of the base cell: two full-rate 24-bit multiply-adds (the compiler's own choice for
            // this expression is a quarter-rate v_mad_u64_u32)
            unsigned zy, cell_b;
            asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(zy) : "v"(lz), "n"(RT_LY), "v"(ly));
            asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(cell_b) : "v"(zy), "s"(RT_LX * 8), "v"(lx * 8));
            unsigned long long* cell = reinterpret_cast<unsigned long long*>(reinterpret_cast<char*>(acc) + cell_b);
// float -> 64-bit two's-complement fixed point: high word = floor (v_cvt_flr_i32_f32), low word = fract * 2^32
// (v_fract_f32 is < 1 by construction; the scaling is one packed multiply per corner pair; v_cvt_u32_f32 saturates)
#define NFS_RT_ADD2(off_, c_)                                                                      \
  {                                                                                                \
    unsigned lo0_, hi0_, lo1_, hi1_;                                                               \
    asm("v_cvt_flr_i32_f32 %0, %1" : "=v"(hi0_) : "v"((c_).x));                                    \
    asm("v_cvt_flr_i32_f32 %0, %1" : "=v"(hi1_) : "v"((c_).y));                                    \
    f32x2 fr_ = {__builtin_amdgcn_fractf((c_).x), __builtin_amdgcn_fractf((c_).y)};                \
    fr_ = fr_ * 4294967296.f;                                                                      \
    asm("v_cvt_u32_f32 %0, %1" : "=v"(lo0_) : "v"(fr_.x));                                         \
    asm("v_cvt_u32_f32 %0, %1" : "=v"(lo1_) : "v"(fr_.y));                                         \
    atomicAdd(cell + (off_), ((unsigned long long)hi0_ << 32) | lo0_);                             \
    atomicAdd(cell + (off_) + 1, ((unsigned long long)hi1_ << 32) | lo1_);                         \
  }
            NFS_RT_ADD2(0, c00)
            NFS_RT_ADD2(RT_LX, c01)
            NFS_RT_ADD2(RT_LY * RT_LX, c10)
            NFS_RT_ADD2(RT_LY * RT_LX + RT_LX, c11)
#undef NFS_RT_ADD2
          }
        }
        g = gn;
        if (COEF) cab = cabn;
        ox = oxn;
      }
    }
  }
  __syncthreads();
  for (int i = t; i < RT_TZ * RT_TY * RT_TX; i += RT_THREADS) {
    const int lx_ = i % RT_TX, ly_ = (i / RT_TX) % RT_TY, lz_ = i / (RT_TX * RT_TY);
    const int z = Tz0 + lz_, y = Ty0 + ly_, x = Tx0 + lx_;
    if (z < D && y < H && x < W) {
      // (the accumulators are indexed from the box's origin; the tile's voxels outside the box hold no sum)
      const bool in = z >= z0 && z <= z1 && y >= y0 && y <= y1 && x >= x0 && x <= x1;
      const long long q = in ? (long long)acc[((z - z0 + 1) * RT_LY + (y - y0) + 1) * RT_LX + (x - x0) + 1] : 0ll;
      // the tiles partition the volume: with `overwrite` every voxel is written exactly once (no zero fill before)
      if (overwrite) g_d[((int64_t)z * H + y) * W + x] = q != 0 ? (float)ldexp((double)q, -kexp) : 0.f;
      else if (q != 0) g_d[((int64_t)z * H + y) * W + x] += (float)ldexp((double)q, -kexp);
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
                   float* workspace, const float* g_max, int overwrite, nfs_stream_t stream) {
  NFS_REQUIRE(g_out && rot && g_d_acc, "nfs_rotate_bwd: null pointer");
  if (int e = check_dims(V, D, H, W, C)) return e;
  if (C == 1 && workspace) {
    const int tz = (D + RT_TZ - 1) / RT_TZ, ty = (H + RT_TY - 1) / RT_TY, tx = (W + RT_TX - 1) / RT_TX;
    const unsigned* gmax_bits = reinterpret_cast<const unsigned*>(g_max);
    if (!gmax_bits) {                       // no max |g_out| supplied: streaming pre-pass
      unsigned* wb = reinterpret_cast<unsigned*>(workspace);
      zero_words(wb, 1, as_stream(stream));                              // (a kernel, not a memset node: common.h)
      const int64_t n = (int64_t)V * D * H * W;
      hipLaunchKernelGGL(absmax_kernel, dim3(2048), dim3(256), 0, as_stream(stream), g_out, n, wb);
      gmax_bits = wb;
    }
    // a voxel collects, per view, unit total weight from interior samples and at most ~max(D,H,W)
    // clamped samples per face direction; 4*nmax per view is a safe bound on the summed weights
    const int nmax = D > H ? (D > W ? D : W) : (H > W ? H : W);
    const float bound_factor = 4.f * (float)nmax * (float)V + 8.f;
    static const int order = [] { const char* e = getenv("NFS_RT_XCD"); return e ? atoi(e) : 0; }();
    const int grid = order == 0 ? tz * ty * tx : 8 * ((tz * ty + 7) / 8) * tx;
    for (int v0 = 0; v0 < V; v0 += RT_VMAX) {
      const int vn = V - v0 < RT_VMAX ? V - v0 : RT_VMAX;
      hipLaunchKernelGGL(rotate_bwd_tiled_kernel<false>, dim3(grid), dim3(RT_THREADS), 0, as_stream(stream),
                         g_out + (int64_t)v0 * D * H * W, rot + v0 * 9, g_d_acc, gmax_bits, bound_factor, vn, D, H, W,
                         ty, tx, (overwrite && v0 == 0) ? 1 : 0, order, (const float2*)nullptr, 1, D, 0,
                         (const int*)nullptr);
    }
    return check_launch("nfs_rotate_bwd(tiled)");
  }
  NFS_REQUIRE(!overwrite, "nfs_rotate_bwd: overwrite needs the tiled adjoint (C == 1 and a workspace)");
  WarpArgs a{nullptr, rot, V, D, H, W, C, 0};
  const int64_t n = (int64_t)V * D * H * W;
  hipLaunchKernelGGL(warp_bwd_kernel<COORD_ROTATE>, dim3(blocks_for(n, 256)), dim3(256), 0, as_stream(stream), a,
                     g_out, g_d_acc, (float*)nullptr);
  return check_launch("nfs_rotate_bwd");
}

// the tiled adjoint on the render's u volume and per-(view, segment, ray) coefficients: sample gradient = A u - B
// (nfs_rotate_render_fwd_coef / nfs_render_ray_coef); bounds [nbounds]: per-block bounds on max |sample gradient|, whose
// maximum sets the fixed-point scale
static int rotate_bwd_coef_impl(const char* who, const float* u_rot, const float* ab, const float* rot, float* g_d_acc, int V,
                                int D, int H, int W, int nseg, int seg_len, const float* bounds, int nbounds, int overwrite,
                                const unsigned long long* live, int dilate, int* lws, nfs_stream_t stream) {
  NFS_REQUIRE(u_rot && ab && rot && g_d_acc && bounds, "nfs_rotate_bwd_coef: null pointer");
  NFS_REQUIRE(nbounds > 0, "nfs_rotate_bwd_coef: no bounds");
  if (int e = check_dims(V, D, H, W, 1)) return e;
  NFS_REQUIRE(nseg > 0 && seg_len > 0 && (int64_t)nseg * seg_len >= D, "nfs_rotate_bwd_coef: the segments do not cover D");
  NFS_REQUIRE(!live || (dilate >= 0 && RT_TX + 2 * dilate <= 63), "nfs_rotate_bwd_coef_live: dilate out of range");
  const int tz = (D + RT_TZ - 1) / RT_TZ, ty = (H + RT_TY - 1) / RT_TY, tx = (W + RT_TX - 1) / RT_TX;
  const int nmax = D > H ? (D > W ? D : W) : (H > W ? H : W);
  const float bound_factor = 4.f * (float)nmax * (float)V + 8.f;
  static const int order = [] { const char* e = getenv("NFS_RT_XCD"); return e ? atoi(e) : 0; }();
  const int grid = (live || order == 0) ? tz * ty * tx : 8 * ((tz * ty + 7) / 8) * tx;
  if (live)
    hipLaunchKernelGGL(rotate_live_boxes_kernel, dim3((tz * ty * tx + LB_WAVES - 1) / LB_WAVES), dim3(LB_THREADS), 0,
                       as_stream(stream), live, D, H, W, ty, tx, dilate, tz * ty * tx, lws);
  if (live)
    hipLaunchKernelGGL(rotate_live_order_kernel, dim3(1), dim3(LB_THREADS), 0, as_stream(stream), tz * ty * tx, lws);
  for (int v0 = 0; v0 < V; v0 += RT_VMAX) {
    const int vn = V - v0 < RT_VMAX ? V - v0 : RT_VMAX;
    hipLaunchKernelGGL(rotate_bwd_tiled_kernel<true>, dim3(grid), dim3(RT_THREADS), 0, as_stream(stream),
                       u_rot + (int64_t)v0 * D * H * W, rot + v0 * 9, g_d_acc, reinterpret_cast<const unsigned*>(bounds),
                       bound_factor, vn, D, H, W, ty, tx, (overwrite && v0 == 0) ? 1 : 0, order,
                       reinterpret_cast<const float2*>(ab) + (int64_t)v0 * nseg * H * W, nseg, seg_len, nbounds,
                       live ? (const int*)lws : (const int*)nullptr);
  }
  return check_launch(who);
}

int nfs_rotate_bwd_coef(const float* u_rot, const float* ab, const float* rot, float* g_d_acc, int V, int D, int H, int W,
                        int nseg, int seg_len, const float* bounds, int nbounds, int overwrite, nfs_stream_t stream) {
  return rotate_bwd_coef_impl("nfs_rotate_bwd_coef", u_rot, ab, rot, g_d_acc, V, D, H, W, nseg, seg_len, bounds, nbounds,
                              overwrite, nullptr, 0, nullptr, stream);
}

// ... restricted to what a velocity variable can use: `live` is the mask nfs_advect_fwd_live / nfs_advect_bwd_adam_fwd_live
// wrote for the CURRENT velocity ([nfs_live_mask_words] 64-bit words, bit = linear voxel index), `dilate` the reach of
// the linear stencil between g_d and the advect adjoint (1 for the 3x3x3 smoothing, 0 without it).  Inside a tile's box
// (the bounding box of its voxels within `dilate` of a live voxel) g_d is bit-identical to nfs_rotate_bwd_coef; outside
// the boxes it is written as zeros (overwrite) or left alone (accumulate) -- values that only ever meet a zero factor.
// A tile without a box returns before its sample loop; the others run longest first.
int nfs_rotate_bwd_coef_live(const float* u_rot, const float* ab, const float* rot, float* g_d_acc, int V, int D, int H,
                             int W, int nseg, int seg_len, const float* bounds, int nbounds, int overwrite,
                             const unsigned long long* live, int dilate, int* workspace, nfs_stream_t stream) {
  NFS_REQUIRE(live && workspace, "nfs_rotate_bwd_coef_live: null mask / workspace");
  return rotate_bwd_coef_impl("nfs_rotate_bwd_coef_live", u_rot, ab, rot, g_d_acc, V, D, H, W, nseg, seg_len, bounds,
                              nbounds, overwrite, live, dilate, workspace, stream);
}

// ints of that workspace: a box per tile, the tiles in the order the adjoint's blocks take them (written by every launch
// before it is read: no initialisation needed)
int nfs_rotate_live_workspace_ints(int D, int H, int W) {
  if (D <= 0 || H <= 0 || W <= 0) return 0;
  const int nt = ((D + RT_TZ - 1) / RT_TZ) * ((H + RT_TY - 1) / RT_TY) * ((W + RT_TX - 1) / RT_TX);
  return LWS_ORDER(nt) + nt;
}

// 64-bit words of the live mask of a [D,H,W] volume (one bit per voxel, rounded up to whole waves of 256 voxels)
int nfs_live_mask_words(int D, int H, int W) {
  const int64_t n = (int64_t)D * H * W;
  return (D > 0 && H > 0 && W > 0 && n < ((int64_t)1 << 30)) ? (int)((n + 255) / 256 * 4) : 0;
}

int nfs_advect_fwd(const float* d, const float* vel, float* out, int D, int H, int W, int C, nfs_stream_t stream) {
  NFS_REQUIRE(d && vel && out, "nfs_advect_fwd: null pointer");
  if (int e = check_dims(1, D, H, W, C)) return e;
  WarpArgs a{d, vel, 1, D, H, W, C, 0};
  const int64_t n = (int64_t)D * H * W;
  if (C == 1 && W >= 2 && H >= 2 && D >= 2 && n % 4 == 0 && n < ((int64_t)1 << 30)) {
    hipLaunchKernelGGL(advect1_kernel<0>, dim3((blocks_for(n, 1024) + 7) / 8 * 8), dim3(256), 0, as_stream(stream), d, vel,
                       (const float*)nullptr, out, D, H, W, AdamFused{}, 0, D);
    return check_launch("nfs_advect_fwd(x4)");
  }
  hipLaunchKernelGGL(warp_fwd_kernel<COORD_ADVECT>, dim3(blocks_for(n, 256)), dim3(256), 0, as_stream(stream), a, out);
  return check_launch("nfs_advect_fwd");
}

// nfs_advect_fwd for a scalar field + the live mask of the back-traced stencils (see AdamFused::live): refuses the shapes
// the four-voxel kernel does not take (NFS_EINVAL; callers then run without a mask)
int nfs_advect_fwd_live(const float* d, const float* vel, float* out, unsigned long long* live, int D, int H, int W,
                        nfs_stream_t stream) {
  NFS_REQUIRE(d && vel && out && live, "nfs_advect_fwd_live: null pointer");
  if (int e = check_dims(1, D, H, W, 1)) return e;
  const int64_t n = (int64_t)D * H * W;
  NFS_REQUIRE(W >= 2 && H >= 2 && D >= 2 && n % 4 == 0 && n < ((int64_t)1 << 30),
              "nfs_advect_fwd_live: needs D, H, W >= 2 and D*H*W %% 4 == 0");
  AdamFused ad{};
  ad.live = live;
  hipLaunchKernelGGL((advect1_kernel<0, true>), dim3((blocks_for(n, 1024) + 7) / 8 * 8), dim3(256), 0, as_stream(stream), d, vel,
                     (const float*)nullptr, out, D, H, W, ad, 0, D);
  return check_launch("nfs_advect_fwd_live");
}

int nfs_advect_bwd(const float* d, const float* vel, const float* g_out, float* g_d_acc, float* g_vel, int D, int H,
                   int W, int C, nfs_stream_t stream) {
  NFS_REQUIRE(vel && g_out, "nfs_advect_bwd: null pointer");
  NFS_REQUIRE(!g_vel || d, "nfs_advect_bwd: g_vel needs d");
  NFS_REQUIRE(g_d_acc || g_vel, "nfs_advect_bwd: nothing to compute");
  if (int e = check_dims(1, D, H, W, C)) return e;
  WarpArgs a{d, vel, 1, D, H, W, C, 0};
  const int64_t n = (int64_t)D * H * W;
  if (C == 1 && !g_d_acc && W >= 2 && H >= 2 && D >= 2 && n % 4 == 0 && n < ((int64_t)1 << 30)) {   // velocity gradient only: no atomics
    hipLaunchKernelGGL(advect1_kernel<1>, dim3((blocks_for(n, 1024) + 7) / 8 * 8), dim3(256), 0, as_stream(stream), d, vel,
                       g_out, g_vel, D, H, W, AdamFused{}, 0, D);
    return check_launch("nfs_advect_bwd(x4)");
  }
  hipLaunchKernelGGL(warp_bwd_kernel<COORD_ADVECT>, dim3(blocks_for(n, 256)), dim3(256), 0, as_stream(stream), a,
                     g_out, g_d_acc, g_vel);
  return check_launch("nfs_advect_bwd");
}

// velocity gradient of advect + TF ApplyAdam on the velocity in one pass (scalar field, C = 1)
int nfs_advect_bwd_adam(const float* d, float* vel, const float* g_out, float* m, float* v, int D, int H, int W,
                        float lr_t, float beta1, float beta2, float eps, nfs_stream_t stream) {
  NFS_REQUIRE(d && vel && g_out && m && v, "nfs_advect_bwd_adam: null pointer");
  if (int e = check_dims(1, D, H, W, 1)) return e;
  const int64_t n = (int64_t)D * H * W;
  NFS_REQUIRE(W >= 2 && H >= 2 && D >= 2 && n % 4 == 0 && n < ((int64_t)1 << 30),
              "nfs_advect_bwd_adam: needs D, H, W >= 2 and D*H*W %% 4 == 0 (use nfs_advect_bwd + nfs_adam_tf_step)");
  hipLaunchKernelGGL(advect1_kernel<2>, dim3((blocks_for(n, 1024) + 7) / 8 * 8), dim3(256), 0, as_stream(stream), d, vel, g_out, vel,
                     D, H, W, AdamFused{m, v, lr_t, beta1, beta2, eps}, 0, D);
  return check_launch("nfs_advect_bwd_adam");
}

// ... and the NEXT iteration's forward advect of the updated velocity in the same pass (adv_next [D,H,W], see AdamFused)
int nfs_advect_bwd_adam_fwd(const float* d, float* vel, const float* g_out, float* m, float* v, float* adv_next, int D, int H,
                            int W, float lr_t, float beta1, float beta2, float eps, nfs_stream_t stream) {
  NFS_REQUIRE(d && vel && g_out && m && v && adv_next, "nfs_advect_bwd_adam_fwd: null pointer");
  NFS_REQUIRE(adv_next != d && adv_next != g_out, "nfs_advect_bwd_adam_fwd: adv_next must not alias d or g_out");
  if (int e = check_dims(1, D, H, W, 1)) return e;
  const int64_t n = (int64_t)D * H * W;
  NFS_REQUIRE(W >= 2 && H >= 2 && D >= 2 && n % 4 == 0 && n < ((int64_t)1 << 30),
              "nfs_advect_bwd_adam_fwd: needs D, H, W >= 2 and D*H*W %% 4 == 0");
  hipLaunchKernelGGL(advect1_kernel<2>, dim3((blocks_for(n, 1024) + 7) / 8 * 8), dim3(256), 0, as_stream(stream), d, vel, g_out, vel,
                     D, H, W, AdamFused{m, v, lr_t, beta1, beta2, eps, adv_next}, 0, D);
  return check_launch("nfs_advect_bwd_adam_fwd");
}

// ... and the live mask of that next forward sample (for the next iteration's nfs_rotate_bwd_coef_live)
int nfs_advect_bwd_adam_fwd_live(const float* d, float* vel, const float* g_out, float* m, float* v, float* adv_next,
                                 unsigned long long* live_next, int D, int H, int W, float lr_t, float beta1, float beta2,
                                 float eps, nfs_stream_t stream) {
  NFS_REQUIRE(d && vel && g_out && m && v && adv_next && live_next, "nfs_advect_bwd_adam_fwd_live: null pointer");
  NFS_REQUIRE(adv_next != d && adv_next != g_out, "nfs_advect_bwd_adam_fwd_live: adv_next must not alias d or g_out");
  if (int e = check_dims(1, D, H, W, 1)) return e;
  const int64_t n = (int64_t)D * H * W;
  NFS_REQUIRE(W >= 2 && H >= 2 && D >= 2 && n % 4 == 0 && n < ((int64_t)1 << 30),
              "nfs_advect_bwd_adam_fwd_live: needs D, H, W >= 2 and D*H*W %% 4 == 0");
  hipLaunchKernelGGL((advect1_kernel<2, true>), dim3((blocks_for(n, 1024) + 7) / 8 * 8), dim3(256), 0, as_stream(stream), d, vel, g_out, vel,
                     D, H, W, AdamFused{m, v, lr_t, beta1, beta2, eps, adv_next, live_next}, 0, D);
  return check_launch("nfs_advect_bwd_adam_fwd_live");
}

// ... skipping the waves whose voxels have never been live (see `ever` in advect1_kernel): `live` holds the mask of the
// CURRENT forward sample on entry and that of the next one on return, `ever` [nfs_live_mask_words] is the OR of every mask
// since m and v were zeroed (zero it with them; the caller must not use it once anything else has written m or v).
// Bit-identical to nfs_advect_bwd_adam_fwd_live.
int nfs_advect_bwd_adam_fwd_live_ever(const float* d, float* vel, const float* g_out, float* m, float* v, float* adv_next,
                                      unsigned long long* live, unsigned long long* ever, int D, int H, int W, float lr_t,
                                      float beta1, float beta2, float eps, nfs_stream_t stream) {
  NFS_REQUIRE(d && vel && g_out && m && v && adv_next && live && ever, "nfs_advect_bwd_adam_fwd_live_ever: null pointer");
  NFS_REQUIRE(adv_next != d && adv_next != g_out, "nfs_advect_bwd_adam_fwd_live_ever: adv_next must not alias d or g_out");
  if (int e = check_dims(1, D, H, W, 1)) return e;
  const int64_t n = (int64_t)D * H * W;
  NFS_REQUIRE(W >= 2 && H >= 2 && D >= 2 && n % 4 == 0 && n < ((int64_t)1 << 30),
              "nfs_advect_bwd_adam_fwd_live_ever: needs D, H, W >= 2 and D*H*W %% 4 == 0");
  // (per-lane predication goes through buffer descriptors: 32-bit byte offsets, 12 n < 2^31; larger volumes skip by waves only)
  static const bool lanes = [] { const char* e = getenv("NFS_EVER_LANES"); return !(e && atoi(e) == 0); }();
  if (lanes && n * 12 < ((int64_t)1 << 31))
    hipLaunchKernelGGL((advect1_kernel<2, true, true>), dim3((blocks_for(n, 1024) + 7) / 8 * 8), dim3(256), 0, as_stream(stream), d, vel,
                       g_out, vel, D, H, W, AdamFused{m, v, lr_t, beta1, beta2, eps, adv_next, live, ever}, 0, D);
  else
    hipLaunchKernelGGL((advect1_kernel<2, true>), dim3((blocks_for(n, 1024) + 7) / 8 * 8), dim3(256), 0, as_stream(stream), d, vel, g_out,
                       vel, D, H, W, AdamFused{m, v, lr_t, beta1, beta2, eps, adv_next, live, ever}, 0, D);
  return check_launch("nfs_advect_bwd_adam_fwd_live_ever");
}

// Slab forms (view-sharded runs shard the replicated field work over D-slabs, engine.GridStylizer): d is the whole
// [D,H,W] density, vel / out / g_out / m / v hold the nz planes [z0, z0 + nz) only.  Same arithmetic per voxel as the
// whole-volume entry points (the same kernel with a plane offset).
int nfs_advect_fwd_slab(const float* d, const float* vel, float* out, int D, int H, int W, int z0, int nz,
                        nfs_stream_t stream) {
  NFS_REQUIRE(d && vel && out, "nfs_advect_fwd_slab: null pointer");
  if (int e = check_dims(1, D, H, W, 1)) return e;
  NFS_REQUIRE(z0 >= 0 && nz >= 1 && z0 + nz <= D, "nfs_advect_fwd_slab: slab outside the volume");
  const int64_t n = (int64_t)nz * H * W;
  NFS_REQUIRE(W >= 2 && H >= 2 && D >= 2 && n % 4 == 0 && (int64_t)D * H * W < ((int64_t)1 << 30),
              "nfs_advect_fwd_slab: needs D, H, W >= 2 and nz*H*W %% 4 == 0");
  hipLaunchKernelGGL(advect1_kernel<0>, dim3((blocks_for(n, 1024) + 7) / 8 * 8), dim3(256), 0, as_stream(stream), d, vel,
                     (const float*)nullptr, out, nz, H, W, AdamFused{}, z0, D);
  return check_launch("nfs_advect_fwd_slab");
}

int nfs_advect_bwd_adam_slab(const float* d, float* vel, const float* g_out, float* m, float* v, int D, int H, int W,
                             int z0, int nz, float lr_t, float beta1, float beta2, float eps, nfs_stream_t stream) {
  NFS_REQUIRE(d && vel && g_out && m && v, "nfs_advect_bwd_adam_slab: null pointer");
  if (int e = check_dims(1, D, H, W, 1)) return e;
  NFS_REQUIRE(z0 >= 0 && nz >= 1 && z0 + nz <= D, "nfs_advect_bwd_adam_slab: slab outside the volume");
  const int64_t n = (int64_t)nz * H * W;
  NFS_REQUIRE(W >= 2 && H >= 2 && D >= 2 && n % 4 == 0 && (int64_t)D * H * W < ((int64_t)1 << 30),
              "nfs_advect_bwd_adam_slab: needs D, H, W >= 2 and nz*H*W %% 4 == 0");
  hipLaunchKernelGGL(advect1_kernel<2>, dim3((blocks_for(n, 1024) + 7) / 8 * 8), dim3(256), 0, as_stream(stream), d, vel, g_out,
                     vel, nz, H, W, AdamFused{m, v, lr_t, beta1, beta2, eps}, z0, D);
  return check_launch("nfs_advect_bwd_adam_slab");
}

int nfs_advect_bwd_adam_fwd_slab(const float* d, float* vel, const float* g_out, float* m, float* v, float* adv_next, int D,
                                 int H, int W, int z0, int nz, float lr_t, float beta1, float beta2, float eps,
                                 nfs_stream_t stream) {
  NFS_REQUIRE(d && vel && g_out && m && v && adv_next, "nfs_advect_bwd_adam_fwd_slab: null pointer");
  NFS_REQUIRE(adv_next != d && adv_next != g_out, "nfs_advect_bwd_adam_fwd_slab: adv_next must not alias d or g_out");
  if (int e = check_dims(1, D, H, W, 1)) return e;
  NFS_REQUIRE(z0 >= 0 && nz >= 1 && z0 + nz <= D, "nfs_advect_bwd_adam_fwd_slab: slab outside the volume");
  const int64_t n = (int64_t)nz * H * W;
  NFS_REQUIRE(W >= 2 && H >= 2 && D >= 2 && n % 4 == 0 && (int64_t)D * H * W < ((int64_t)1 << 30),
              "nfs_advect_bwd_adam_fwd_slab: needs D, H, W >= 2 and nz*H*W %% 4 == 0");
  hipLaunchKernelGGL(advect1_kernel<2>, dim3((blocks_for(n, 1024) + 7) / 8 * 8), dim3(256), 0, as_stream(stream), d, vel, g_out,
                     vel, nz, H, W, AdamFused{m, v, lr_t, beta1, beta2, eps, adv_next}, z0, D);
  return check_launch("nfs_advect_bwd_adam_fwd_slab");
}

// one step of StylerBase._transport (styler_base.py:59-89) with the temporal filter's weighted accumulation fused in
int nfs_transport_step(const float* g, const float* u, float scale, float w_g, const float* addend, float w_addend,
                       float* out, int D, int H, int W, int C, nfs_stream_t stream) {
  NFS_REQUIRE(g && u && out, "nfs_transport_step: null pointer");
  NFS_REQUIRE(g != out, "nfs_transport_step: out must not alias g (it is a gather)");
  if (int e = check_dims(1, D, H, W, C)) return e;
  const int64_t n = (int64_t)D * H * W;
  if ((C == 1 || C == 3) && W >= 2 && H >= 2 && D >= 2 && n < ((int64_t)1 << 29)) {
    const unsigned blocks = (blocks_for(n, 512) + 7) / 8 * 8;
    if (C == 3)
      hipLaunchKernelGGL(transport_step_kernel<3>, dim3(blocks), dim3(256), 0, as_stream(stream), g, u, addend, out, D, H,
                         W, scale, w_g, w_addend);
    else
      hipLaunchKernelGGL(transport_step_kernel<1>, dim3(blocks), dim3(256), 0, as_stream(stream), g, u, addend, out, D, H,
                         W, scale, w_g, w_addend);
    return check_launch("nfs_transport_step");
  }
  hipLaunchKernelGGL(transport_step_generic_kernel, dim3(blocks_for(n, 256)), dim3(256), 0, as_stream(stream), g, u,
                     addend, out, D, H, W, C, scale, w_g, w_addend);
  return check_launch("nfs_transport_step(generic)");
}

}  // extern "C"
