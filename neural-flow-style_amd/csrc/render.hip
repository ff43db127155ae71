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
  if (liquid >= 2) {
    // ray modes of north_star beside the reference's two: 2 = reduce_max along the ray (the line the reference keeps
    // commented out, styler_3p.py:149), 3 = reduce_mean.  One thread per ray, neighbouring threads on W: the reduction is
    // a register loop over coalesced planes.  raysum carries what the adjoint needs: the maximum / the sum.
    float m = -INFINITY, sum = 0.f;
#pragma unroll 4
    for (int z = 0; z < D; ++z) {
      const float s = col[(int64_t)z * HW];
      m = fmaxf(m, s);
      sum += s;
    }
    img[gid] = liquid == 2 ? m : sum / (float)D;
    if (raysum) raysum[gid] = liquid == 2 ? m : sum;
    return;
  }
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
                                                         int V, int D, int HW, float tau, int liquid,
                                                         unsigned* __restrict__ gmax_bits) {
  __shared__ float red[16];
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  float amax = 0.f;
  if (gid < (int64_t)V * HW) {
    const int v = (int)(gid / HW);
    const int px = (int)(gid - (int64_t)v * HW);
    const int64_t base = (int64_t)v * D * HW + px;
    const float total = raysum[gid];
    const float g = g_img[gid];
    const float ntau = -tau * 1.44269504088896341f;                 // exp(-tau a) = exp2(ntau a)
    if (liquid == 2) {
      // reduce_max: the gradient goes to the cells that hold the maximum, split equally among ties (TF's rule); the
      // column is read once to count them and once more as it is overwritten (g_d may be d itself)
      float ties = 0.f;
      for (int z = 0; z < D; ++z) ties += d[base + (int64_t)z * HW] == total ? 1.f : 0.f;
      const float gd = g / ties;
      for (int z = 0; z < D; ++z) {
        const float sv = d[base + (int64_t)z * HW];
        g_d[base + (int64_t)z * HW] = sv == total ? gd : 0.f;
      }
      amax = fabsf(gd);
    } else if (liquid == 3) {
      const float gd = g / (float)D;                                  // reduce_mean
      for (int z = 0; z < D; ++z) g_d[base + (int64_t)z * HW] = gd;
      amax = fabsf(gd);
    } else if (liquid) {
      const float gd = g * tau * __builtin_amdgcn_exp2f(total * ntau);
      for (int z = 0; z < D; ++z) g_d[base + (int64_t)z * HW] = gd;
      amax = fabsf(gd);
    } else {
      // g_d may be d itself (in-place adjoint), so the compiler cannot lift a load above the store before it:
      // batches of RB samples, the next batch's loads issued before this batch's stores, keep 8-16 loads in
      // flight per ray (one thread per ray has only ~5 waves per SIMD to hide the HBM latency with)
      constexpr int RB = 8;
      float prefix = 0.f, P = 0.f;
      auto step = [&](float s, int z) __attribute__((always_inline)) {
        const float T = __builtin_amdgcn_exp2f((total - prefix) * ntau);
        P = fmaf(s, T, P);
        const float o = g * (T - tau * P);
        g_d[base + (int64_t)z * HW] = o;
        amax = fmaxf(amax, fabsf(o));
        prefix += s;
      };
      const int nfull = D / RB;
      float cur[RB], nxt[RB];
      if (nfull > 0) {
#pragma unroll
        for (int u = 0; u < RB; ++u) cur[u] = d[base + (int64_t)u * HW];
      }
      for (int b = 0; b < nfull; ++b) {
        const int z0 = b * RB;
        if (b + 1 < nfull) {
#pragma unroll
          for (int u = 0; u < RB; ++u) nxt[u] = d[base + (int64_t)(z0 + RB + u) * HW];
        }
#pragma unroll
        for (int u = 0; u < RB; ++u) step(cur[u], z0 + u);
#pragma unroll
        for (int u = 0; u < RB; ++u) cur[u] = nxt[u];
      }
      for (int z = nfull * RB; z < D; ++z) step(d[base + (int64_t)z * HW], z);
    }
  }
  // optional by-product: max |g_d| (as float bits) for the fixed-point scale of the rotate adjoint that follows
  if (gmax_bits) {
    amax = block_max(amax, red);
    if (threadIdx.x == 0 && amax > 0.f) atomicMax(gmax_bits, __float_as_uint(fminf(amax, 3.0e38f)));
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

// ---- lean trilinear sampler for the ray march -------------------------------------------------------------
// The march is VALU-bound (64 M samples x ~120 instructions at 200^3 x 8 views), so the per-sample arithmetic is
// stripped to the minimum that keeps the reference semantics (transform.py:395-417):
//  * the voxel coordinate of a sample is affine in the step index: x_a(z) = A_a z + B_a, one FMA per axis;
//  * border replication = clamp the coordinate to [0, n-1] first (outside the volume the two clipped corners
//    coincide and their weights sum to 1, so the value is the border voxel's either way), base cell
//    min(floor(x), n-2), weight x - base in [0,1]: no per-corner index clamps, the x-pair is always in-row;
//  * 32-bit unsigned offsets (a 200^3 volume is 32 MB), lerp form a + w (b - a);
//  * exp(-tau acc) on the hardware exp2 (v_exp_f32, <= 2 ulp).
struct RayAffine { float az, bz, ay, by, ax, bx; };   // voxel coords along the ray: (az z + bz, ay z + by, ax z + bx)

__device__ __forceinline__ RayAffine ray_affine(const float* r, int D, int H, int W, int h, int w) {
  const float gy = lin_coord(h, H), gx = lin_coord(w, W);
  const float step = D > 1 ? 2.f / (float)(D - 1) : 0.f;
  const float hz = 0.5f * (float)(D - 1), hy = 0.5f * (float)(H - 1), hx = 0.5f * (float)(W - 1);
  RayAffine q;
  q.az = r[0] * step * hz; q.bz = (r[1] * gy + r[2] * gx - r[0] + 1.f) * hz;
  q.ay = r[3] * step * hy; q.by = (r[4] * gy + r[5] * gx - r[3] + 1.f) * hy;
  q.ax = r[6] * step * hx; q.bx = (r[7] * gy + r[8] * gx - r[6] + 1.f) * hx;
  return q;
}

struct VolDims { float nz1, ny1, nx1, nz2, ny2, nx2; unsigned W, HW; };

__device__ __forceinline__ float lean_sample(const float* __restrict__ vol, const VolDims& n, const RayAffine& q,
                                             float zf) {
  const float cz = __builtin_amdgcn_fmed3f(fmaf(q.az, zf, q.bz), 0.f, n.nz1);
  const float cy = __builtin_amdgcn_fmed3f(fmaf(q.ay, zf, q.by), 0.f, n.ny1);
  const float cx = __builtin_amdgcn_fmed3f(fmaf(q.ax, zf, q.bx), 0.f, n.nx1);
  const float fz = fminf(floorf(cz), n.nz2), fy = fminf(floorf(cy), n.ny2), fx = fminf(floorf(cx), n.nx2);
  const float wz = cz - fz, wy = cy - fy, wx = cx - fx;
  const unsigned base = (unsigned)(int)fz * n.HW + (unsigned)(int)fy * n.W + (unsigned)(int)fx;
  const F2u p00 = *reinterpret_cast<const F2u*>(vol + base);
  const F2u p01 = *reinterpret_cast<const F2u*>(vol + base + n.W);
  const F2u p10 = *reinterpret_cast<const F2u*>(vol + base + n.HW);
  const F2u p11 = *reinterpret_cast<const F2u*>(vol + base + n.HW + n.W);
  const float a00 = fmaf(wx, p00.y - p00.x, p00.x), a01 = fmaf(wx, p01.y - p01.x, p01.x);
  const float a10 = fmaf(wx, p10.y - p10.x, p10.x), a11 = fmaf(wx, p11.y - p11.x, p11.x);
  const float b0 = fmaf(wy, a01 - a00, a00), b1 = fmaf(wy, a11 - a10, a10);
  return fmaf(wz, b1 - b0, b0);
}

// The same sampler split in two, for the march that keeps the previous sample's four texel pairs in registers:
// a ray moves ~1 cell in z per step and changes its (y, x) cell only every 6-60 steps, so the upper plane of
// sample z is the lower plane of sample z+1 most of the time and only 2 of the 4 pair loads are new.
struct LeanCoord {
  unsigned base;
  float wz, wy, wx;
};
// Batch form: sample u of a batch sits at z - u; its affine offsets (b - u a) are precomputed, and the cell
// index uses 24-bit multiplies (full rate; exact for H W < 2^24, which the launcher's D H W < 2^30 implies)
struct RayOff {
  float bz, by, bx;
};
__device__ __forceinline__ LeanCoord lean_coord24(const VolDims& n, const RayAffine& q, const RayOff& o, float zf) {
  const float cz = __builtin_amdgcn_fmed3f(fmaf(q.az, zf, o.bz), 0.f, n.nz1);
  const float cy = __builtin_amdgcn_fmed3f(fmaf(q.ay, zf, o.by), 0.f, n.ny1);
  const float cx = __builtin_amdgcn_fmed3f(fmaf(q.ax, zf, o.bx), 0.f, n.nx1);
  const float fz = fminf(floorf(cz), n.nz2), fy = fminf(floorf(cy), n.ny2), fx = fminf(floorf(cx), n.nx2);
  const unsigned base = __umul24((unsigned)(int)fz, n.HW) + __umul24((unsigned)(int)fy, n.W) + (unsigned)(int)fx;
  return LeanCoord{base, cz - fz, cy - fy, cx - fx};
}
__device__ __forceinline__ F2u buf_load_f2(__amdgpu_buffer_rsrc_t rsrc, unsigned voff, unsigned soff) {
  return __builtin_bit_cast(F2u, __builtin_amdgcn_raw_buffer_load_b64(rsrc, voff, soff, 0));
}
// l/u := previous sample's texel pairs on the lanes that do not load them: on `shift` lanes (plane advanced by one)
// the new upper plane is the old lower plane; on `same` lanes everything carries over.
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void masked_carry(F2u& l0, F2u& l1, F2u& u0, F2u& u1, const F2u& pl0, const F2u& pl1,
                                             const F2u& pu0, const F2u& pu1, uint64_t shift, uint64_t same) {
  f32x2 a = __builtin_bit_cast(f32x2, l0), b = __builtin_bit_cast(f32x2, l1), c = __builtin_bit_cast(f32x2, u0),
        e = __builtin_bit_cast(f32x2, u1);
  uint64_t saved;
  asm volatile(
      "s_mov_b64 %[sv], exec\n\t"
      "s_mov_b64 exec, %[msh]\n\t"
      "v_mov_b64 %[u0], %[pl0]\n\t"
      "v_mov_b64 %[u1], %[pl1]\n\t"
      "s_mov_b64 exec, %[msa]\n\t"
      "v_mov_b64 %[l0], %[pl0]\n\t"
      "v_mov_b64 %[l1], %[pl1]\n\t"
      "v_mov_b64 %[u0], %[pu0]\n\t"
      "v_mov_b64 %[u1], %[pu1]\n\t"
      "s_mov_b64 exec, %[sv]"
      : [l0] "+v"(a), [l1] "+v"(b), [u0] "+v"(c), [u1] "+v"(e), [sv] "=&s"(saved)
      : [pl0] "v"(__builtin_bit_cast(f32x2, pl0)), [pl1] "v"(__builtin_bit_cast(f32x2, pl1)),
        [pu0] "v"(__builtin_bit_cast(f32x2, pu0)), [pu1] "v"(__builtin_bit_cast(f32x2, pu1)), [msh] "s"(shift),
        [msa] "s"(same));
  l0 = __builtin_bit_cast(F2u, a);
  l1 = __builtin_bit_cast(F2u, b);
  u0 = __builtin_bit_cast(F2u, c);
  u1 = __builtin_bit_cast(F2u, e);
}
__device__ __forceinline__ float lean_blend(const LeanCoord& c, F2u p00, F2u p01, F2u p10, F2u p11) {
  const float a00 = fmaf(c.wx, p00.y - p00.x, p00.x), a01 = fmaf(c.wx, p01.y - p01.x, p01.x);
  const float a10 = fmaf(c.wx, p10.y - p10.x, p10.x), a11 = fmaf(c.wx, p11.y - p11.x, p11.x);
  const float b0 = fmaf(c.wy, a01 - a00, a00), b1 = fmaf(c.wy, a11 - a10, a10);
  return fmaf(c.wz, b1 - b0, b0);
}

// Segmented variant (D >= 16, H, W >= 2): a block = 64 rays x 4 depth segments (one wave per segment, far
// segment first).  One thread per ray leaves only V*H*W = 320 k threads for a 200-step serial march; splitting
// the ray four ways quadruples the loads in flight (one view per GPU: 0.074 -> 0.041 ms).  Segments combine exactly:
//   I = sum_s exp(-tau * P_s) * I_s,   P_s = sum of the ray sums of the segments farther than s,
// because the transmittance of a sample is exp(-tau (P_s + local suffix sum)).
//
// UOUT (transmittance mode, round 4): instead of the samples, d_rot receives per sample
//   u = t + tau * i,   t = exp(-tau * local suffix sum incl. the sample),  i = the segment's image sum BEFORE the sample,
// and seg_out the per-(view, segment, ray) triple (segment ray sum, segment image sum, max |u|).  With
// E_s = exp(-tau P_s) and F_s = sum over the farther segments of E_s' I_s', the image gradient of a sample is
//   dI/ds = T - tau * sum_{z' <= z} s T = E_s u - tau (I - F_s):
// affine in u with per-(ray, segment) coefficients (render_ray_coef_kernel), so the rotate adjoint forms it from u on
// the fly and the render-adjoint pass over the rotated volume (K4a) is not needed.
constexpr int RR_SEG = 4;
template <bool REUSE, bool UOUT = false>
__global__ void __launch_bounds__(256, 4) rotate_render_fwd_seg_kernel(const float* __restrict__ d,
                                                                    const float* __restrict__ rot,
                                                                    float* __restrict__ img,
                                                                    float* __restrict__ raysum,
                                                                    float* __restrict__ d_rot, int V, int D, int H,
                                                                    int W, float tau, int liquid, int lxb, int band,
                                                                    float* __restrict__ seg_out = nullptr) {
  __shared__ float seg_sum[RR_SEG][64], seg_I[RR_SEG][64];
  const int HW = H * W;
  // a wave is one segment: telling the compiler so keeps the depth index and the plane offsets in scalar registers
  const int lane = threadIdx.x & 63, seg = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t total = (int64_t)V * HW;
  // XCD-aware order: consecutive workgroups go round-robin to the 8 XCDs; give each XCD a contiguous range of
  // rays (one view at V = 8), so that the planes its waves walk through stay in its own 4 MB L2 (with the plain
  // order every XCD sweeps all views at once: 54 % L2 hit rate, 1.2 GB fetched for a 32 MB volume)
  const unsigned per_xcd = gridDim.x / 8;
  const unsigned logical = (blockIdx.x % 8) * per_xcd + blockIdx.x / 8;
  bool live;
  int v, h, w;
  if (lxb == 0) {                                                 // 64 consecutive pixels (may wrap to the next row)
    const int64_t gid_raw = (int64_t)logical * 64 + lane;
    live = gid_raw < total;
    const int64_t g = live ? gid_raw : total - 1;
    v = (int)(g / HW);
    const int p = (int)(g - (int64_t)v * HW);
    h = p / W;
    w = p - h * W;
  } else {
    // a wave is a 2^lxb x 2^(6-lxb) pixel tile: under a rotation the 64 x 1 strip spreads over ~64 sin(theta)
    // source planes, one cache line each; a tile spreads over 2^lxb sin(theta) planes and its rows share lines
    // (row y+1 of one lane row is row y of the next), so the gather touches ~40 % fewer lines
    const int tw = 1 << lxb, th = 64 >> lxb;
    const int ntx = (W + tw - 1) >> lxb, nty = (H + th - 1) / th;
    const unsigned tiles = (unsigned)ntx * nty;
    unsigned vv, t;
    if (band) {
      // all XCDs on the same view at the same time, XCD k on the k-th band of its tiles: every L2 serves a
      // 1/8 band of the volume instead of the whole of it, and the kept-volume stores form one front
      const unsigned per_band = (tiles + 7) / 8, s_ = blockIdx.x / 8;
      vv = s_ / per_band;
      t = (blockIdx.x % 8) * per_band + (s_ - vv * per_band);
    } else {
      vv = logical / tiles;
      t = logical - vv * tiles;
    }
    const int ty = t / ntx, tx = t - ty * ntx;
    const int hh = ty * th + (lane >> lxb), ww = tx * tw + (lane & (tw - 1));
    live = vv < (unsigned)V && t < tiles && hh < H && ww < W;
    v = min((int)vv, V - 1);
    h = min(hh, H - 1);
    w = min(ww, W - 1);
  }
  const int px = h * W + w;
  const int64_t gid = (int64_t)v * HW + px;
  float r[9];
#pragma unroll
  for (int i = 0; i < 9; ++i) r[i] = rot[v * 9 + i];
  const RayAffine q = ray_affine(r, D, H, W, h, w);
  const VolDims n{(float)(D - 1), (float)(H - 1), (float)(W - 1), (float)(D - 2), (float)(H - 2), (float)(W - 2),
                  (unsigned)W, (unsigned)HW};
  const float ntau = -tau * 1.44269504088896341f;                 // exp(-tau a) = exp2(ntau a)
  const int L = (D + RR_SEG - 1) / RR_SEG;
  const int zhi = D - 1 - seg * L, zlo = max(zhi - L + 1, 0);     // segment 0 is the far end
  float* drow = d_rot ? d_rot + (int64_t)v * D * HW + px : nullptr;
  float acc = 0.f, I = 0.f;
  [[maybe_unused]] float umax = 0.f;
  int z = zhi;
  // REUSE path: buffer addressing (32-bit byte offsets; the launcher checks the sizes) so that the +W / +HW
  // neighbours and the output plane cost scalar offsets instead of 64-bit vector adds
  [[maybe_unused]] const __amdgpu_buffer_rsrc_t vol_rsrc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(d), 0, 0xffffffff, 0x00020000);
  [[maybe_unused]] const __amdgpu_buffer_rsrc_t rot_rsrc =
      __builtin_amdgcn_make_buffer_rsrc(d_rot ? d_rot : img, 0, 0xffffffff, 0x00020000);
  [[maybe_unused]] const unsigned so_w = n.W * 4u, so_hw = n.HW * 4u, so_hww = (n.HW + n.W) * 4u;
  [[maybe_unused]] const unsigned rot_vo = ((unsigned)v * (unsigned)D * n.HW + (unsigned)px) * 4u;
  [[maybe_unused]] RayOff qo[4];                                  // offsets of the four samples of a batch
#pragma unroll
  for (int u = 0; u < 4; ++u) qo[u] = RayOff{q.bz - (float)u * q.az, q.by - (float)u * q.ay, q.bx - (float)u * q.ax};
  unsigned pbase = 0xffffffffu;                                   // previous sample's cell and its texel pairs
  F2u L0{0.f, 0.f}, L1{0.f, 0.f}, U0{0.f, 0.f}, U1{0.f, 0.f};     // L: plane bz (rows by, by+1), U: plane bz+1
  constexpr int NB = 4;                                           // samples per batch
  for (; z >= zlo + NB - 1; z -= NB) {
    float sv[NB];
    if constexpr (REUSE) {
      LeanCoord c[NB];
      bool same[NB], shift[NB];
      F2u l0[NB], l1[NB], u0[NB], u1[NB];
      unsigned pb = pbase;
      const float zf = (float)z;
#pragma unroll
      for (int u = 0; u < NB; ++u) {
        c[u] = lean_coord24(n, q, qo[u], zf);
        same[u] = c[u].base == pb;
        shift[u] = c[u].base + n.HW == pb;
        pb = c[u].base;
      }
#pragma unroll
      for (int u = 0; u < NB; ++u) {
        if (!same[u]) {
          const unsigned vo = c[u].base * 4u;
          l0[u] = buf_load_f2(vol_rsrc, vo, 0);
          l1[u] = buf_load_f2(vol_rsrc, vo, so_w);
          if (!shift[u]) {
            u0[u] = buf_load_f2(vol_rsrc, vo, so_hw);
            u1[u] = buf_load_f2(vol_rsrc, vo, so_hww);
          }
        }
      }
#pragma unroll
      for (int u = 0; u < NB; ++u) {
        // the carry-over as masked register moves (2 for a plane advance, 4 for an unchanged cell); written as
        // a branch the compiler turns it into ~23 copies per sample
        masked_carry(l0[u], l1[u], u0[u], u1[u], L0, L1, U0, U1, __builtin_amdgcn_ballot_w64(shift[u]),
                     __builtin_amdgcn_ballot_w64(same[u]));
        L0 = l0[u]; L1 = l1[u]; U0 = u0[u]; U1 = u1[u];
        sv[u] = lean_blend(c[u], l0[u], l1[u], u0[u], u1[u]);
      }
      pbase = pb;
      // what the kept volume receives: the samples (stored before the serial transmittance chain), or (UOUT) u
      auto keep = [&](const float* ov) {
        if (d_rot && live) {
#pragma unroll
          for (int u = 0; u < NB; ++u) {  // plane offset is wave-uniform: it rides in the scalar offset of the store
            if (lxb && lxb < 5)           // half-line pieces: let the L2 merge them with the neighbouring tile's
              __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, ov[u]), rot_rsrc, rot_vo,
                                                    (unsigned)(z - u) * n.HW * 4u, 0);
            else
              __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, ov[u]), rot_rsrc, rot_vo,
                                                    (unsigned)(z - u) * n.HW * 4u, 2 /* nt */);
          }
        }
      };
      if constexpr (!UOUT) keep(sv);
      [[maybe_unused]] float uu[NB];
#pragma unroll
      for (int u = 0; u < NB; ++u) {
        acc += sv[u];
        const float t = __builtin_amdgcn_exp2f(acc * ntau);
        if constexpr (UOUT) {
          uu[u] = fmaf(tau, I, t);
          umax = fmaxf(umax, fabsf(uu[u]));
        }
        I = fmaf(sv[u], t, I);
      }
      if constexpr (UOUT) keep(uu);
      continue;
    } else {
#pragma unroll
      for (int u = 0; u < NB; ++u) sv[u] = lean_sample(d, n, q, (float)(z - u));
    }
#pragma unroll
    for (int u = 0; u < NB; ++u) {
      acc += sv[u];
      const float t = __builtin_amdgcn_exp2f(acc * ntau);
      float o = sv[u];
      if (UOUT) {
        o = fmaf(tau, I, t);
        umax = fmaxf(umax, fabsf(o));
      }
      // rotated volume kept for the adjoint: streaming store, must not evict the volume from L2
      if (drow && live) __builtin_nontemporal_store(o, drow + (int64_t)(z - u) * HW);
      I = fmaf(sv[u], t, I);
    }
  }
  for (; z >= zlo; --z) {
    const float sone = lean_sample(d, n, q, (float)z);
    acc += sone;
    const float t = __builtin_amdgcn_exp2f(acc * ntau);
    float o = sone;
    if (UOUT) {
      o = fmaf(tau, I, t);
      umax = fmaxf(umax, fabsf(o));
    }
    if (drow && live) __builtin_nontemporal_store(o, drow + (int64_t)z * HW);
    I = fmaf(sone, t, I);
  }
  if (UOUT && live) {                                             // [3][V][RR_SEG][HW]
    const int64_t plane = (int64_t)V * RR_SEG * HW, at = ((int64_t)v * RR_SEG + seg) * HW + px;
    seg_out[at] = acc;
    seg_out[plane + at] = I;
    seg_out[2 * plane + at] = umax;
  }
  seg_sum[seg][lane] = acc;
  seg_I[seg][lane] = I;
  __syncthreads();
  if (seg != 0 || !live) return;
  float P = 0.f, Itot = 0.f;
#pragma unroll
  for (int s2 = 0; s2 < RR_SEG; ++s2) {
    Itot = fmaf(__builtin_amdgcn_exp2f(P * ntau), seg_I[s2][lane], Itot);
    P += seg_sum[s2][lane];
  }
  img[gid] = liquid ? 1.f - __builtin_amdgcn_exp2f(P * ntau) : Itot;
  if (raysum) raysum[gid] = P;
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
// ---- multi-block max-normalisation (groups of >= 16 k elements: one 1024-thread block per group took 20-30 us
// on 8 of 256 CUs) -------------------------------------------------------------------------------------------
constexpr int MN_NB = 32;          // blocks per group

// float max through integer atomics: non-negative floats order like signed ints, negative ones inversely like
// unsigned ints; the slot is initialised to -inf by the host (hipMemsetD32Async)
__device__ __forceinline__ void atomic_max_float(float* addr, float v) {
  if (v >= 0.f) atomicMax(reinterpret_cast<int*>(addr), __float_as_int(v));
  else atomicMin(reinterpret_cast<unsigned*>(addr), __float_as_uint(v));
}

// (a kernel, not hipMemsetD32Async: as a memset NODE of a captured hipGraph the -inf fill was seen to run out of order with
// the atomics behind it -- two ranks sharing a GPU, 64^3 sequence: losses off by +361 M from the first replay on)
__global__ void maxnorm_init_kernel(float* __restrict__ gmax, int G) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < G) gmax[i] = -INFINITY;
}

__global__ void __launch_bounds__(256) maxnorm_max_kernel(const float* __restrict__ img, float* __restrict__ gmax, int n) {
  __shared__ float red[16];
  const float* x = img + (int64_t)blockIdx.y * n;
  float m = -INFINITY;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) m = fmaxf(m, x[i]);
  m = block_max(m, red);
  if (threadIdx.x == 0) atomic_max_float(gmax + blockIdx.y, m);
}

__global__ void __launch_bounds__(256) maxnorm_div_kernel(const float* __restrict__ img, const float* __restrict__ gmax,
                                                          float* __restrict__ out, int n) {
  const float m = gmax[blockIdx.y];
  const float* x = img + (int64_t)blockIdx.y * n;
  float* o = out + (int64_t)blockIdx.y * n;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) o[i] = x[i] / m;
}

// adjoint, phase 1: per-block partial sums of g*x and of the tie count -> part[group][block][2]
__global__ void __launch_bounds__(256) maxnorm_bwd_part_kernel(const float* __restrict__ img,
                                                               const float* __restrict__ gmax,
                                                               const float* __restrict__ g_out,
                                                               float* __restrict__ part, int n) {
  __shared__ float red[16];
  const float* x = img + (int64_t)blockIdx.y * n;
  const float* gy = g_out + (int64_t)blockIdx.y * n;
  const float m = gmax[blockIdx.y];
  float s = 0.f, ties = 0.f;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const float xi = x[i];
    s += gy[i] * xi;
    ties += (xi == m) ? 1.f : 0.f;
  }
  s = block_sum(s, red);
  ties = block_sum(ties, red);
  if (threadIdx.x == 0) {
    part[((int64_t)blockIdx.y * MN_NB + blockIdx.x) * 2] = s;
    part[((int64_t)blockIdx.y * MN_NB + blockIdx.x) * 2 + 1] = ties;
  }
}

// phase 2: every thread sums the MN_NB partials in the same order (deterministic), then applies
__global__ void __launch_bounds__(256) maxnorm_bwd_apply_kernel(const float* __restrict__ img,
                                                                const float* __restrict__ gmax,
                                                                const float* __restrict__ g_out,
                                                                const float* __restrict__ part,
                                                                float* __restrict__ g_img, int n) {
  const float* x = img + (int64_t)blockIdx.y * n;
  const float* gy = g_out + (int64_t)blockIdx.y * n;
  float* gx = g_img + (int64_t)blockIdx.y * n;
  const float m = gmax[blockIdx.y];
  float s = 0.f, ties = 0.f;
#pragma unroll 8
  for (int k = 0; k < MN_NB; ++k) {
    s += part[((int64_t)blockIdx.y * MN_NB + k) * 2];
    ties += part[((int64_t)blockIdx.y * MN_NB + k) * 2 + 1];
  }
  const float corr = s / (m * m) / ties;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    float g = gy[i] / m;
    if (x[i] == m) g -= corr;
    gx[i] = g;
  }
}

// ---- max-normalisation fused with the loss-net input (styler_3p.py:158 + styler_base.py:41-45, vgg.py:50-53) ------------
// The grey render at the loss net's own size (resize_scale 1): x[.,c] = (img / max) * 255 - mean[c] in one pass, and the
// adjoint g_img = adjoint_maxnorm(255 * (g_x[.,0] + g_x[.,1] + g_x[.,2])) in two (partial sums, apply) -- the same
// arithmetic (to float32 rounding) as nfs_maxnorm_fwd + nfs_loss_net_input_fwd and nfs_loss_net_input_bwd + nfs_maxnorm_bwd,
// without the [V,H,W] intermediates and with two launches less per direction.
__constant__ float kInputMean[3] = {0.485f * 255.f, 0.456f * 255.f, 0.406f * 255.f};

__global__ void __launch_bounds__(256) maxnorm_input_kernel(const float* __restrict__ img, const float* __restrict__ gmax,
                                                            float* __restrict__ xo, int n) {
  const float m = gmax[blockIdx.y];
  const float* x = img + (int64_t)blockIdx.y * n;
  float* o = xo + (int64_t)blockIdx.y * n * 3;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const float v = __fmul_rn(x[i] / m, 255.f);        // (rounded like the stored intermediate of the two-step form)
    o[3 * (int64_t)i] = __fsub_rn(v, kInputMean[0]);
    o[3 * (int64_t)i + 1] = __fsub_rn(v, kInputMean[1]);
    o[3 * (int64_t)i + 2] = __fsub_rn(v, kInputMean[2]);
  }
}

// large groups, ONE launch: every one of the MN_NB blocks of a group first takes the maximum of the WHOLE group (n floats
// from L2: 160 KB at 200 x 200 -- the maximum does not depend on the order it is taken in, so all blocks hold the same
// value), then does its share of the pass.  Replaces -inf fill + atomic-max kernel + pass (three launches of ~5 us each on
// the latency floor) by one; same arithmetic per pixel, bit-identical x and gmax.
__global__ void __launch_bounds__(256) maxnorm_input_allmax_kernel(const float* __restrict__ img, float* __restrict__ gmax,
                                                                   float* __restrict__ xo, int n) {
  __shared__ float red[16];
  const float* x = img + (int64_t)blockIdx.y * n;
  float* o = xo + (int64_t)blockIdx.y * n * 3;
  float m = -INFINITY;
  const int n4 = n >> 2;
  if ((reinterpret_cast<uintptr_t>(x) & 15) == 0) {
    const float4* x4 = reinterpret_cast<const float4*>(x);
    for (int i = threadIdx.x; i < n4; i += 256) {
      const float4 v = x4[i];
      m = fmaxf(m, fmaxf(fmaxf(v.x, v.y), fmaxf(v.z, v.w)));
    }
    for (int i = 4 * n4 + threadIdx.x; i < n; i += 256) m = fmaxf(m, x[i]);
  } else {
    for (int i = threadIdx.x; i < n; i += 256) m = fmaxf(m, x[i]);
  }
  m = block_max(m, red);
  if (blockIdx.x == 0 && threadIdx.x == 0) gmax[blockIdx.y] = m;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const float v = __fmul_rn(x[i] / m, 255.f);        // (rounded like the stored intermediate of the two-step form)
    o[3 * (int64_t)i] = __fsub_rn(v, kInputMean[0]);
    o[3 * (int64_t)i + 1] = __fsub_rn(v, kInputMean[1]);
    o[3 * (int64_t)i + 2] = __fsub_rn(v, kInputMean[2]);
  }
}

// small groups (< 16384 pixels): one block per group does the maximum and the pass (no memset, no atomics)
__global__ void __launch_bounds__(1024) maxnorm_input_small_kernel(const float* __restrict__ img, float* __restrict__ gmax,
                                                                   float* __restrict__ xo, int n) {
  __shared__ float red[16];
  const float* x = img + (int64_t)blockIdx.x * n;
  float* o = xo + (int64_t)blockIdx.x * n * 3;
  float m = -INFINITY;
  for (int i = threadIdx.x; i < n; i += blockDim.x) m = fmaxf(m, x[i]);
  m = block_max(m, red);
  if (threadIdx.x == 0) gmax[blockIdx.x] = m;
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    const float v = __fmul_rn(x[i] / m, 255.f);        // (rounded like the stored intermediate of the two-step form)
    o[3 * (int64_t)i] = __fsub_rn(v, kInputMean[0]);
    o[3 * (int64_t)i + 1] = __fsub_rn(v, kInputMean[1]);
    o[3 * (int64_t)i + 2] = __fsub_rn(v, kInputMean[2]);
  }
}

__device__ __forceinline__ float input_grad_sum(const float* g, int64_t i) {
  return __fadd_rn(__fadd_rn(__fmul_rn(g[3 * i], 255.f), __fmul_rn(g[3 * i + 1], 255.f)), __fmul_rn(g[3 * i + 2], 255.f));
}

__global__ void __launch_bounds__(256) maxnorm_input_bwd_part_kernel(const float* __restrict__ img,
                                                                     const float* __restrict__ gmax,
                                                                     const float* __restrict__ g_x,
                                                                     float* __restrict__ part, int n) {
  __shared__ float red[16];
  const float* x = img + (int64_t)blockIdx.y * n;
  const float* gx = g_x + (int64_t)blockIdx.y * n * 3;
  const float m = gmax[blockIdx.y];
  float s = 0.f, ties = 0.f;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const float xi = x[i];
    s += input_grad_sum(gx, i) * xi;
    ties += (xi == m) ? 1.f : 0.f;
  }
  s = block_sum(s, red);
  ties = block_sum(ties, red);
  if (threadIdx.x == 0) {
    part[((int64_t)blockIdx.y * MN_NB + blockIdx.x) * 2] = s;
    part[((int64_t)blockIdx.y * MN_NB + blockIdx.x) * 2 + 1] = ties;
  }
}

__global__ void __launch_bounds__(256) maxnorm_input_bwd_apply_kernel(const float* __restrict__ img,
                                                                      const float* __restrict__ gmax,
                                                                      const float* __restrict__ g_x,
                                                                      const float* __restrict__ part,
                                                                      float* __restrict__ g_img, int n) {
  const float* x = img + (int64_t)blockIdx.y * n;
  const float* gx = g_x + (int64_t)blockIdx.y * n * 3;
  float* go = g_img + (int64_t)blockIdx.y * n;
  const float m = gmax[blockIdx.y];
  float s = 0.f, ties = 0.f;
#pragma unroll 8
  for (int k = 0; k < MN_NB; ++k) {
    s += part[((int64_t)blockIdx.y * MN_NB + k) * 2];
    ties += part[((int64_t)blockIdx.y * MN_NB + k) * 2 + 1];
  }
  const float corr = s / (m * m) / ties;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    float g = input_grad_sum(gx, i) / m;
    if (x[i] == m) g -= corr;
    go[i] = g;
  }
}

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

// per (view, depth segment, ray): the coefficients of dI/ds = A u - B for the rotate adjoint's COEF form, from the UOUT
// forward's per-segment sums (seg: [3][V][RR_SEG][HW] = segment ray sum, segment image sum, max |u|) and the image
// gradient g [V][HW].  E_s = exp(-tau P_s) (P_s: ray sums of the farther segments), F_s = sum_{s' < s} E_s' I_s',
// I = F_{RR_SEG}:  A_s = g E_s,  B_s = g tau (I - F_s).  max over everything of |A_s| umax_s + |B_s| bounds every
// sample gradient of the batch: the rotate adjoint's fixed-point scale.  Every block writes ITS maximum to
// bounds[blockIdx.x] and the consumer takes the maximum of them: no zero-initialised word, no atomics (a 4-byte
// memset node in front of an atomicMax did not reliably precede it in hipGraph replays).
__global__ void __launch_bounds__(256) render_ray_coef_kernel(const float* __restrict__ g_img,
                                                              const float* __restrict__ seg, float2* __restrict__ ab,
                                                              int V, int HW, float tau, float* __restrict__ bounds) {
  __shared__ float red[16];
  const int64_t total = (int64_t)V * HW, gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  float bound = 0.f;
  if (gid < total) {
    const int v = (int)(gid / HW), px = (int)(gid - (int64_t)v * HW);
    const int64_t plane = (int64_t)V * RR_SEG * HW, at0 = (int64_t)v * RR_SEG * HW + px;
    const float ntau = -tau * 1.44269504088896341f, g = g_img[gid];
    float S[RR_SEG], Is[RR_SEG], um[RR_SEG], E[RR_SEG], F[RR_SEG];
    float P = 0.f, Itot = 0.f;
#pragma unroll
    for (int s2 = 0; s2 < RR_SEG; ++s2) {          // same order of sums as the forward's own combination
      S[s2] = seg[at0 + (int64_t)s2 * HW];
      Is[s2] = seg[plane + at0 + (int64_t)s2 * HW];
      um[s2] = seg[2 * plane + at0 + (int64_t)s2 * HW];
      E[s2] = __builtin_amdgcn_exp2f(P * ntau);
      F[s2] = Itot;
      Itot = fmaf(E[s2], Is[s2], Itot);
      P += S[s2];
    }
#pragma unroll
    for (int s2 = 0; s2 < RR_SEG; ++s2) {
      const float A = g * E[s2], B = g * tau * (Itot - F[s2]);
      ab[at0 + (int64_t)s2 * HW] = make_float2(A, B);
      bound = fmaxf(bound, fmaf(fabsf(A), um[s2], fabsf(B)));
    }
  }
  bound = block_max(bound, red);
  if (threadIdx.x == 0) bounds[blockIdx.x] = bound > 0.f ? fminf(bound, 3.0e38f) : 0.f;   // (NaN -> 0: no finite scale)
}

}  // namespace nfs

using namespace nfs;

extern "C" {

int nfs_render_fwd(const float* d, float* img, float* raysum, int V, int D, int H, int W, float tau, int liquid,
                   nfs_stream_t stream) {
  NFS_REQUIRE(d && img, "nfs_render_fwd: null pointer");
  NFS_REQUIRE(V > 0 && D > 0 && H > 0 && W > 0, "nfs_render_fwd: non-positive dimension");
  NFS_REQUIRE(liquid >= 0 && liquid <= 3, "nfs_render_fwd: mode must be 0 (transmittance), 1 (liquid), 2 (max) or 3 (mean)");
  const int64_t n = (int64_t)V * H * W;
  hipLaunchKernelGGL(render_fwd_kernel, dim3(blocks_for(n, 256)), dim3(256), 0, as_stream(stream), d, img, raysum, V, D,
                     H * W, tau, liquid);
  return check_launch("nfs_render_fwd");
}

// Segmented render adjoint for small batches (one or two views per GPU): with V*H*W = 40 k rays a one-thread-per-ray
// march is latency-bound (80 us for a 32 MB volume).  Four waves share 64 rays, one depth segment each; a first walk
// gives every segment its ray sum S and R = sum s*exp(-tau (total - local prefix)), the block combines them into the
// segment's starting prefix / weighted sum (the transmittance factorises: exp(-tau (total - p0 - local)) =
// exp(tau p0) * exp(-tau (total - local))), a second walk writes the gradient.  The volume is read twice, so this
// form is only used while it stays cache-resident (<= 64 MB).
extern "C++" {
template <int NSEG>
__global__ void __launch_bounds__(64 * NSEG) render_bwd_seg_kernel(const float* d, const float* __restrict__ raysum,
                                                                   const float* __restrict__ g_img, float* g_d, int V,
                                                                   int D, int HW, float tau,
                                                                   unsigned* __restrict__ gmax_bits) {
  __shared__ float seg_S[NSEG][64], seg_R[NSEG][64], red[16];
  const int lane = threadIdx.x & 63, seg = threadIdx.x >> 6;
  const int64_t total_rays = (int64_t)V * HW;
  const int64_t gid_raw = (int64_t)blockIdx.x * 64 + lane;
  const bool live = gid_raw < total_rays;
  const int64_t gid = live ? gid_raw : total_rays - 1;
  const int v = (int)(gid / HW);
  const int px = (int)(gid - (int64_t)v * HW);
  const int64_t base = (int64_t)v * D * HW + px;
  const float total = raysum[gid], g = g_img[gid];
  const float ntau = -tau * 1.44269504088896341f;
  const int L = (D + NSEG - 1) / NSEG;
  const int zlo = min(seg * L, D), zhi = min(zlo + L, D);         // segment 0 starts at z = 0 (prefix order)
  float S = 0.f, R = 0.f;
#pragma unroll 4
  for (int z = zlo; z < zhi; ++z) {
    const float sv = d[base + (int64_t)z * HW];
    R = fmaf(sv, __builtin_amdgcn_exp2f((total - S) * ntau), R);
    S += sv;
  }
  seg_S[seg][lane] = S;
  seg_R[seg][lane] = R;
  __syncthreads();
  float p0 = 0.f, P = 0.f;                                        // prefix and weighted sum before this segment
  for (int j = 0; j < seg; ++j) {
    P = fmaf(__builtin_amdgcn_exp2f(-p0 * ntau), seg_R[j][lane], P);
    p0 += seg_S[j][lane];
  }
  float prefix = p0, amax = 0.f;
#pragma unroll 4
  for (int z = zlo; z < zhi; ++z) {
    const float sv = d[base + (int64_t)z * HW];
    const float T = __builtin_amdgcn_exp2f((total - prefix) * ntau);
    P = fmaf(sv, T, P);
    const float o = g * (T - tau * P);
    if (live) g_d[base + (int64_t)z * HW] = o;
    amax = fmaxf(amax, fabsf(o));
    prefix += sv;
  }
  if (gmax_bits) {
    amax = block_max(live ? amax : 0.f, red);
    if (threadIdx.x == 0 && amax > 0.f) atomicMax(gmax_bits, __float_as_uint(fminf(amax, 3.0e38f)));
  }
}
}  // extern "C++"

int nfs_render_bwd(const float* d, const float* raysum, const float* g_img, float* g_d, int V, int D, int H, int W,
                   float tau, int liquid, float* gmax_out, nfs_stream_t stream) {
  NFS_REQUIRE(d && raysum && g_img && g_d, "nfs_render_bwd: null pointer");
  NFS_REQUIRE(V > 0 && D > 0 && H > 0 && W > 0, "nfs_render_bwd: non-positive dimension");
  const int64_t n = (int64_t)V * H * W;
  if (gmax_out) zero_words(gmax_out, 1, as_stream(stream));               // (a kernel, not a memset node: common.h)
  static const bool no_seg = getenv("NFS_RB_NOSEG") != nullptr;   // timing comparisons only
  NFS_REQUIRE(liquid >= 0 && liquid <= 3, "nfs_render_bwd: mode must be 0 (transmittance), 1 (liquid), 2 (max) or 3 (mean)");
  if (!liquid && !no_seg && D >= 4 * RR_SEG && (int64_t)V * D * H * W <= ((int64_t)16 << 20)) {
    // segments per ray: 4, or 8 while the 64-ray blocks are few (200^2 rays: one view 32.5 -> 24.7 us, two views
    // 45.0 -> 41.2 us; 16 segments 25.4 / 39.7; tools/render_family_by_views.py, NFS_RB_SEG forces a count)
    static const int seg_env = getenv("NFS_RB_SEG") ? atoi(getenv("NFS_RB_SEG")) : 0;
    const int64_t blocks = blocks_for(n, 64);
    int nseg = blocks >= 2048 ? 4 : 8;
    if (seg_env == 4 || seg_env == 8 || seg_env == 16) nseg = seg_env;
    while (nseg > 4 && D < 4 * nseg) nseg /= 2;
    unsigned* gm = reinterpret_cast<unsigned*>(gmax_out);
    if (nseg == 16)
      hipLaunchKernelGGL(render_bwd_seg_kernel<16>, dim3(blocks), dim3(1024), 0, as_stream(stream), d, raysum, g_img, g_d,
                         V, D, H * W, tau, gm);
    else if (nseg == 8)
      hipLaunchKernelGGL(render_bwd_seg_kernel<8>, dim3(blocks), dim3(512), 0, as_stream(stream), d, raysum, g_img, g_d,
                         V, D, H * W, tau, gm);
    else
      hipLaunchKernelGGL(render_bwd_seg_kernel<4>, dim3(blocks), dim3(256), 0, as_stream(stream), d, raysum, g_img, g_d,
                         V, D, H * W, tau, gm);
    return check_launch("nfs_render_bwd(segmented)");
  }
  hipLaunchKernelGGL(render_bwd_kernel, dim3(blocks_for(n, 256)), dim3(256), 0, as_stream(stream), d, raysum, g_img, g_d,
                     V, D, H * W, tau, liquid, reinterpret_cast<unsigned*>(gmax_out));
  return check_launch("nfs_render_bwd");
}

int nfs_rotate_render_fwd(const float* d, const float* rot, float* img, float* raysum, float* d_rot, int V, int D,
                          int H, int W, float tau, int liquid, nfs_stream_t stream) {
  NFS_REQUIRE(d && rot && img, "nfs_rotate_render_fwd: null pointer");
  NFS_REQUIRE(V > 0 && D > 0 && H > 0 && W > 0, "nfs_rotate_render_fwd: non-positive dimension");
  NFS_REQUIRE(liquid == 0 || liquid == 1, "nfs_rotate_render_fwd: ray modes 2 / 3 go through nfs_rotate_fwd + nfs_render_fwd");
  const int64_t n = (int64_t)V * H * W;
  static const bool no_seg = getenv("NFS_RR_NOSEG") != nullptr;   // timing comparisons only
  static const bool no_reuse = getenv("NFS_RR_NOREUSE") != nullptr;   // timing comparisons only
  static const int tile_env = getenv("NFS_RR_TILE") ? atoi(getenv("NFS_RR_TILE")) : -1;
  if (W >= 2 && H >= 2 && D >= 4 * RR_SEG && (int64_t)D * H * W < (1ll << 31) && !no_seg) {
    // wave footprint: 16 x 4 pixel tiles when the image has room for them, else 64 consecutive pixels
    const int lxb = tile_env >= 0 ? tile_env : (W >= 16 && H >= 4 ? 4 : 0);
    static const int band = getenv("NFS_RR_BAND") ? atoi(getenv("NFS_RR_BAND")) : 1;   // 0: one view per XCD
    int64_t waves = blocks_for(n, 64);
    const int64_t tiles_v = (int64_t)((W + (1 << lxb) - 1) >> lxb) * ((H + (64 >> lxb) - 1) / (64 >> lxb));
    if (lxb) waves = (int64_t)V * tiles_v;
    if (lxb && band) waves = (int64_t)V * ((tiles_v + 7) / 8) * 8;
    const dim3 grid((waves + 7) / 8 * 8);
    const bool fits32 = (int64_t)V * D * H * W < (1ll << 30);      // byte offsets of the buffer addressing
    if (fits32 && !no_reuse)
      hipLaunchKernelGGL(rotate_render_fwd_seg_kernel<true>, grid, dim3(256), 0, as_stream(stream), d, rot, img,
                         raysum, d_rot, V, D, H, W, tau, liquid, lxb, lxb ? band : 0);
    else
      hipLaunchKernelGGL(rotate_render_fwd_seg_kernel<false>, grid, dim3(256), 0, as_stream(stream), d, rot, img,
                         raysum, d_rot, V, D, H, W, tau, liquid, lxb, lxb ? band : 0);
  } else
    hipLaunchKernelGGL(rotate_render_fwd_kernel, dim3(blocks_for(n, 256)), dim3(256), 0, as_stream(stream), d, rot,
                       img, raysum, d_rot, V, D, H, W, tau, liquid);
  return check_launch("nfs_rotate_render_fwd");
}

// does the segmented forward (and with it the u / coefficient form of the adjoint) take this shape?  nseg / seg_len: the
// depth segments of a ray, far end first (segment of plane z = (D - 1 - z) / seg_len)
int nfs_render_coef_layout(int V, int D, int H, int W, int* nseg, int* seg_len) {
  NFS_REQUIRE(V > 0 && D > 0 && H > 0 && W > 0, "nfs_render_coef_layout: non-positive dimension");
  if (nseg) *nseg = RR_SEG;
  if (seg_len) *seg_len = (D + RR_SEG - 1) / RR_SEG;
  const bool ok = W >= 2 && H >= 2 && D >= 4 * RR_SEG && (int64_t)D * H * W < (1ll << 31) &&
                  (int64_t)V * D * H * W < (1ll << 30);
  if (!ok) {
    set_error("nfs_render_coef_layout: shape outside the segmented forward (needs D >= 16, H, W >= 2, V*D*H*W < 2^30)");
    return NFS_EINVAL;
  }
  return NFS_OK;
}

int nfs_rotate_render_fwd_coef(const float* d, const float* rot, float* img, float* raysum, float* u_rot, float* seg,
                               int V, int D, int H, int W, float tau, nfs_stream_t stream) {
  NFS_REQUIRE(d && rot && img && u_rot && seg, "nfs_rotate_render_fwd_coef: null pointer");
  if (int e = nfs_render_coef_layout(V, D, H, W, nullptr, nullptr)) return e;
  const int lxb = W >= 16 && H >= 4 ? 4 : 0;
  int64_t waves = blocks_for((int64_t)V * H * W, 64);
  const int64_t tiles_v = (int64_t)((W + (1 << lxb) - 1) >> lxb) * ((H + (64 >> lxb) - 1) / (64 >> lxb));
  if (lxb) waves = (int64_t)V * ((tiles_v + 7) / 8) * 8;
  hipLaunchKernelGGL((rotate_render_fwd_seg_kernel<true, true>), dim3((waves + 7) / 8 * 8), dim3(256), 0,
                     as_stream(stream), d, rot, img, raysum, u_rot, V, D, H, W, tau, 0, lxb, lxb ? 1 : 0, seg);
  return check_launch("nfs_rotate_render_fwd_coef");
}

int nfs_render_ray_coef_bounds(int V, int H, int W) {
  if (V <= 0 || H <= 0 || W <= 0) return 0;
  return (int)blocks_for((int64_t)V * H * W, 256);
}

int nfs_render_ray_coef(const float* g_img, const float* seg, float* ab, float* bounds, int V, int H, int W, float tau,
                        nfs_stream_t stream) {
  NFS_REQUIRE(g_img && seg && ab && bounds, "nfs_render_ray_coef: null pointer");
  NFS_REQUIRE(V > 0 && H > 0 && W > 0, "nfs_render_ray_coef: non-positive dimension");
  const int64_t n = (int64_t)V * H * W;
  hipLaunchKernelGGL(render_ray_coef_kernel, dim3(blocks_for(n, 256)), dim3(256), 0, as_stream(stream), g_img, seg,
                     reinterpret_cast<float2*>(ab), V, H * W, tau, bounds);
  return check_launch("nfs_render_ray_coef");
}

int nfs_rotate_render_bwd(const float* d, const float* rot, const float* raysum, const float* g_img, float* g_d_acc,
                          int V, int D, int H, int W, float tau, int liquid, nfs_stream_t stream) {
  NFS_REQUIRE(d && rot && raysum && g_img && g_d_acc, "nfs_rotate_render_bwd: null pointer");
  NFS_REQUIRE(V > 0 && D > 0 && H > 0 && W > 0, "nfs_rotate_render_bwd: non-positive dimension");
  NFS_REQUIRE(liquid == 0 || liquid == 1, "nfs_rotate_render_bwd: ray modes 2 / 3 go through nfs_render_bwd + nfs_rotate_bwd");
  const int64_t n = (int64_t)V * H * W;
  hipLaunchKernelGGL(rotate_render_bwd_kernel, dim3(blocks_for(n, 256)), dim3(256), 0, as_stream(stream), d, rot, raysum,
                     g_img, g_d_acc, V, D, H, W, tau, liquid);
  return check_launch("nfs_rotate_render_bwd");
}

int nfs_maxnorm_fwd(const float* img, float* out, float* gmax, int G, int n, nfs_stream_t stream) {
  NFS_REQUIRE(img && out && gmax, "nfs_maxnorm_fwd: null pointer");
  NFS_REQUIRE(G > 0 && n > 0, "nfs_maxnorm_fwd: non-positive size");
  if (n >= 16384) {
    hipLaunchKernelGGL(maxnorm_init_kernel, dim3((G + 63) / 64), dim3(64), 0, as_stream(stream), gmax, G);
    hipLaunchKernelGGL(maxnorm_max_kernel, dim3(MN_NB, G), dim3(256), 0, as_stream(stream), img, gmax, n);
    hipLaunchKernelGGL(maxnorm_div_kernel, dim3(MN_NB, G), dim3(256), 0, as_stream(stream), img, gmax, out, n);
    return check_launch("nfs_maxnorm_fwd(multi-block)");
  }
  hipLaunchKernelGGL(maxnorm_fwd_kernel, dim3(G), dim3(1024), 0, as_stream(stream), img, out, gmax, n);
  return check_launch("nfs_maxnorm_fwd");
}

int nfs_maxnorm_input_fwd(const float* img, float* x, float* gmax, int G, int n, nfs_stream_t stream) {
  NFS_REQUIRE(img && x && gmax, "nfs_maxnorm_input_fwd: null pointer");
  NFS_REQUIRE(G > 0 && n > 0, "nfs_maxnorm_input_fwd: non-positive size");
  if (n < 16384) {
    hipLaunchKernelGGL(maxnorm_input_small_kernel, dim3(G), dim3(1024), 0, as_stream(stream), img, gmax, x, n);
    return check_launch("nfs_maxnorm_input_fwd");
  }
  static const bool three = [] { const char* e = getenv("NFS_MAXNORM_3K"); return e && atoi(e) != 0; }();
  if (!three && (int64_t)n <= (1 << 20)) {        // (beyond 1 M pixels per group the redundant maxima stop being free)
    hipLaunchKernelGGL(maxnorm_input_allmax_kernel, dim3(MN_NB, G), dim3(256), 0, as_stream(stream), img, gmax, x, n);
    return check_launch("nfs_maxnorm_input_fwd(one launch)");
  }
  hipLaunchKernelGGL(maxnorm_init_kernel, dim3((G + 63) / 64), dim3(64), 0, as_stream(stream), gmax, G);
  hipLaunchKernelGGL(maxnorm_max_kernel, dim3(MN_NB, G), dim3(256), 0, as_stream(stream), img, gmax, n);
  hipLaunchKernelGGL(maxnorm_input_kernel, dim3(MN_NB, G), dim3(256), 0, as_stream(stream), img, gmax, x, n);
  return check_launch("nfs_maxnorm_input_fwd");
}

int nfs_maxnorm_input_bwd(const float* img, const float* gmax, const float* g_x, float* g_img, int G, int n,
                          float* workspace, nfs_stream_t stream) {
  NFS_REQUIRE(img && gmax && g_x && g_img && workspace, "nfs_maxnorm_input_bwd: null pointer");
  NFS_REQUIRE(G > 0 && n > 0, "nfs_maxnorm_input_bwd: non-positive size");
  hipLaunchKernelGGL(maxnorm_input_bwd_part_kernel, dim3(MN_NB, G), dim3(256), 0, as_stream(stream), img, gmax, g_x,
                     workspace, n);
  hipLaunchKernelGGL(maxnorm_input_bwd_apply_kernel, dim3(MN_NB, G), dim3(256), 0, as_stream(stream), img, gmax, g_x,
                     workspace, g_img, n);
  return check_launch("nfs_maxnorm_input_bwd");
}

int nfs_maxnorm_bwd(const float* img, const float* gmax, const float* g_out, float* g_img, int G, int n,
                    float* workspace, nfs_stream_t stream) {
  NFS_REQUIRE(img && gmax && g_out && g_img, "nfs_maxnorm_bwd: null pointer");
  NFS_REQUIRE(G > 0 && n > 0, "nfs_maxnorm_bwd: non-positive size");
  if (workspace && n >= 16384) {   // two-phase, MN_NB blocks per group, partial sums through the workspace
    hipLaunchKernelGGL(maxnorm_bwd_part_kernel, dim3(MN_NB, G), dim3(256), 0, as_stream(stream), img, gmax, g_out,
                       workspace, n);
    hipLaunchKernelGGL(maxnorm_bwd_apply_kernel, dim3(MN_NB, G), dim3(256), 0, as_stream(stream), img, gmax, g_out,
                       workspace, g_img, n);
    return check_launch("nfs_maxnorm_bwd(multi-block)");
  }
  hipLaunchKernelGGL(maxnorm_bwd_kernel, dim3(G), dim3(1024), 0, as_stream(stream), img, gmax, g_out, g_img, n);
  return check_launch("nfs_maxnorm_bwd");
}

}  // extern "C"
