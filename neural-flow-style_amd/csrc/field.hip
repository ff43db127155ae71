// Field-side elementwise / small-stencil kernels, all HBM-bound:
//   A9  smoothing conv + clamp        (styler_3p.py:112-125)
//   A10 TF ApplyAdam                  (styler_3p.py:320-323)
//   A5  loss-net input                (styler_base.py:33-45, vgg.py:50-53)
//   A6  2x2 average pool              (vgg.py:93-104)
//   A12 total variation               (styler_base.py:211-213)
#include "common.h"

namespace nfs {

// ---- A9 ------------------------------------------------------------------------------
// z-marching separable stencil: a thread owns one (y,x) column of a z-chunk.  Per step it loads
// the 3x3 (y,x) neighbourhood of ONE new plane (9 loads, coalesced along x), reduces it to the
// 2-D filtered value and combines the last three of those along z -- 9 loads per output instead
// of 27 (the 2-D sums are reused by three consecutive z).  HBM traffic = one read + one write.
// Tiled z-march.  A block (8 waves) owns TY = 8 or 16 rows x txe (<= 64) columns of a z-chunk, one column (two rows
// of it at TY = 16) per thread.
// Per plane it stages the tile plus a one-cell halo in LDS (zero outside the volume = SAME padding; for the
// adjoint the staged value is g * (pre >= 0), the TF Maximum mask read from the sign bit of the forward output),
// each thread forms the 3x3 in-plane sum from LDS and combines the last three plane sums along z in registers.
// Global loads per output: (TY+2)/TY * (txe+2)/txe * (zc+2)/zc ~ 1.3-1.4 (x2 for the adjoint) instead of 9
// (18): the first version (one thread per column, 9 L1-cached loads per plane) was bound by vector-memory
// instruction issue at 0.8-1.0 TB/s whatever the chunk length.  Planes are double-buffered in LDS and the
// next plane's global loads are issued before the current plane is consumed: one barrier per plane.
constexpr int SM_TXMAX = 64, SM_THREADS = 512, SM_ZCHUNK = 25;
#ifndef NFS_SM_PF
#define NFS_SM_PF 2
#endif
constexpr int SM_PF = NFS_SM_PF;                  // planes of register look-ahead (even: the LDS buffer parity follows the ring slot)

// Loads and stores go through buffer descriptors based at the block's first plane (32-bit offsets relative to it, so
// volumes beyond 4 GB work): an element outside the volume or outside the thread's staging slots gets an offset beyond
// num_records -- the hardware returns zero for the load (= SAME padding) and drops the store.  No branch surrounds a
// load or a store, so the compiler counts them (s_waitcnt vmcnt(n), not 0) and SM_PF planes really stay in flight.
constexpr uint32_t SM_OOB = 0x80000000u;

// RY = rows per thread: a block covers 8 * RY rows.  RY = 2 (round 4) halves the barriers and the halo rows per voxel:
// 200^3 forward 21.1 -> 17.5 us back to back, ~27 -> ~24 us behind other kernels (tools/micro/field_variants.hip
// "S2").  Its two staging slots per thread hold (16 + 2) x (txe + 2) elements, so the column tiles stop at 54.
__device__ __forceinline__ float sm_tap(float wa, float wb, float a, float b, float c) {
  return __fmaf_rn(wa, c, __fmaf_rn(wb, b, wa * a));
}

template <int RY> struct SmTile {
  static constexpr int TY = 8 * RY, TXE_MAX = RY == 1 ? SM_TXMAX : 54;
};

template <bool BWD, int RY>
__global__ void __launch_bounds__(SM_THREADS) smooth3d_kernel(const float* __restrict__ in,
                                                              const float* __restrict__ act, float* __restrict__ out,
                                                              int D, int H, int W, float k, int txe, int ntx, int nty, int nz) {
  constexpr int TY = SmTile<RY>::TY, LW = SM_TXMAX + 2;
  constexpr int TB = (TY + 2) * LW;
  __shared__ float tile[2][TY + 2][LW];
  __shared__ float dump[TB + 1];                  // where the threads without a staging slot put their zeros
  const int t = threadIdx.x, tx = t & 63, ty = t >> 6;
  // consecutive workgroups go round-robin to the 8 XCDs: give each XCD a contiguous range of tiles, so that x / y
  // neighbours (which share halo lines, and 128-byte lines at 50-column tile edges) meet in one L2
  const unsigned per_xcd = gridDim.x / 8;
  const unsigned lb = (blockIdx.x % 8) * per_xcd + blockIdx.x / 8;
  if (lb >= (unsigned)(ntx * nty * nz)) return;
  const int bx = lb % ntx, by = (lb / ntx) % nty, bz = lb / (ntx * nty);
  const int x0 = bx * txe, y0 = by * TY, z0 = bz * SM_ZCHUNK, z1 = min(z0 + SM_ZCHUNK, D);
  const int x = x0 + tx;
  // 1-D weights [1,k,1]/(k+2); k <= 0 skips the conv (identity)
  const float inv = k > 0.f ? 1.f / (k + 2.f) : 1.f;
  const float wa = k > 0.f ? inv : 0.f, wb = k > 0.f ? k * inv : 1.f;
  // the planes this block touches: zb .. ze - 1 (its chunk plus one plane either side, inside the volume)
  const int zb = max(z0 - 1, 0), ze = min(z1 + 1, D);
  const uint32_t plane_b = (uint32_t)H * (uint32_t)W * 4u, recs = (uint32_t)(ze - zb) * plane_b;
  const int64_t base = (int64_t)zb * H * W;
  const __amdgpu_buffer_rsrc_t in_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(in + base), 0, recs, 0x00020000);
  [[maybe_unused]] const __amdgpu_buffer_rsrc_t act_rsrc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(BWD ? act + base : in + base), 0, recs, 0x00020000);
  const __amdgpu_buffer_rsrc_t out_rsrc = __builtin_amdgcn_make_buffer_rsrc(out + base, 0, recs, 0x00020000);
  // staging slots of this thread: elements t and t + 512 of the (TY+2) x (txe+2) halo'd tile
  const int cols = txe + 2, ne = (TY + 2) * cols;
  uint32_t so[2];
  float* sp[2];
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    const int e = t + SM_THREADS * s, r = e / cols, c = e - r * cols;
    const int yy = y0 - 1 + r, xx = x0 - 1 + c;
    so[s] = (e < ne && yy >= 0 && yy < H && xx >= 0 && xx < W) ? (uint32_t)(yy * W + xx) * 4u : SM_OOB;
    sp[s] = e < ne ? &tile[0][r][c] : dump;                     // (+ TB for the second buffer: dump[TB])
  }
  uint32_t oo[RY];
#pragma unroll
  for (int r = 0; r < RY; ++r) {
    const int y = y0 + ty + 8 * r;
    oo[r] = (tx < txe && x < W && y < H) ? (uint32_t)(y * W + x) * 4u : SM_OOB;
  }
  float v0, v1;
  // plane p_ of the volume -> (v0, v1); a plane outside the block's range reads as zeros
#define NFS_SM_GLOAD(p_)                                                                               \
  {                                                                                                    \
    const int pz_ = (p_);                                                                              \
    const bool in_ = pz_ >= zb && pz_ < ze;                                                            \
    const uint32_t so_ = in_ ? (uint32_t)(pz_ - zb) * plane_b : 0u;                                    \
    const uint32_t a0_ = in_ ? so[0] : SM_OOB, a1_ = in_ ? so[1] : SM_OOB;                             \
    v0 = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(in_rsrc, a0_, so_, 0));        \
    v1 = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(in_rsrc, a1_, so_, 0));        \
    if (BWD) {                                     /* g_out * (pre >= 0): the sign bit of the forward output */ \
      const uint32_t m0_ = __builtin_amdgcn_raw_buffer_load_b32(act_rsrc, a0_, so_, 0);                \
      const uint32_t m1_ = __builtin_amdgcn_raw_buffer_load_b32(act_rsrc, a1_, so_, 0);                \
      v0 = (m0_ >> 31) ? 0.f : v0;                                                                     \
      v1 = (m1_ >> 31) ? 0.f : v1;                                                                     \
    }                                                                                                  \
  }
  // plane z0 - 1 goes to LDS directly; planes z0 .. z0 + SM_PF - 1 wait in a register ring (measured at 200^3,
  // tools/variant_sweep.sh: forward 24.8 us with branches around the loads, 21.2 branch-free with SM_PF = 2, 23.2 with
  // 4 or 6 -- the stores of the same loop share the counter with the loads and the compiler keeps the waits near
  // vmcnt(2) whatever the ring depth; what is left is the per-plane barrier of eight waves)
  NFS_SM_GLOAD(z0 - 1)
  sp[0][0] = v0;
  sp[1][0] = v1;
  float rv0[SM_PF], rv1[SM_PF];
#pragma unroll
  for (int u = 0; u < SM_PF; ++u) {
    NFS_SM_GLOAD(z0 + u)
    rv0[u] = v0;
    rv1[u] = v1;
  }
  __syncthreads();
  float pm[RY], pc[RY];
#pragma unroll
  for (int r = 0; r < RY; ++r) pm[r] = pc[r] = 0.f;
  // iteration p consumes plane p (buffer (p - z0 + 1) & 1), stages plane p + 1 from the ring and fetches plane
  // p + 1 + SM_PF into the freed ring slot
  for (int pb = z0 - 1; pb <= z1; pb += SM_PF) {
#pragma unroll
    for (int u = 0; u < SM_PF; ++u) {
      const int p = pb + u;
      if (p > z1) break;
      const int b = u & 1;
      float pn[RY];
#pragma unroll
      for (int r = 0; r < RY; ++r) {
        const float* t0 = &tile[b][ty + 8 * r][tx];
        // (explicit fmas: the 8- and 16-row instances must round alike, whatever the compiler would contract)
        const float ra = sm_tap(wa, wb, t0[0], t0[1], t0[2]);
        const float rb = sm_tap(wa, wb, t0[LW], t0[LW + 1], t0[LW + 2]);
        const float rc = sm_tap(wa, wb, t0[2 * LW], t0[2 * LW + 1], t0[2 * LW + 2]);
        pn[r] = sm_tap(wa, wb, ra, rb, rc);
      }
      sp[0][(b ^ 1) * TB] = rv0[u];              // plane p + 1 (its buffer was last read in iteration p - 1)
      sp[1][(b ^ 1) * TB] = rv1[u];
      NFS_SM_GLOAD(p + 1 + SM_PF)
      rv0[u] = v0;
      rv1[u] = v1;
      const bool wr = p >= z0 + 1;               // (p - 1 is a plane of this chunk: inside [zb, ze))
#pragma unroll
      for (int r = 0; r < RY; ++r) {
        float o = sm_tap(wa, wb, pm[r], pc[r], pn[r]);    // output plane p - 1
        // forward: max(pre,0) with the sign bit carrying (pre < 0) for the TF Maximum gradient
        if (!BWD) o = (o >= 0.f) ? fabsf(o) : (o < 0.f ? -0.0f : o);
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, o), out_rsrc, wr ? oo[r] : SM_OOB,
                                              wr ? (uint32_t)(p - 1 - zb) * plane_b : 0u, 0);
        pm[r] = pc[r];
        pc[r] = pn[r];
      }
      __syncthreads();
    }
  }
#undef NFS_SM_GLOAD
}

// 16-row tiles once they still give every CU a block; 8-row tiles for small volumes (NFS_SM_ROWS=8 / 16 forces one)
template <bool BWD>
static void launch_smooth3d(const float* in, const float* act, float* out, int D, int H, int W, float k, hipStream_t st) {
  static const int forced = [] { const char* e = getenv("NFS_SM_ROWS"); return e ? atoi(e) : 0; }();
  const int nz = (D + SM_ZCHUNK - 1) / SM_ZCHUNK;
  auto go = [&](auto ry) {
    constexpr int RY = decltype(ry)::value;
    const int ntx = (W + SmTile<RY>::TXE_MAX - 1) / SmTile<RY>::TXE_MAX, txe = (W + ntx - 1) / ntx;   // balanced column tiles
    const int nty = (H + SmTile<RY>::TY - 1) / SmTile<RY>::TY;
    hipLaunchKernelGGL((smooth3d_kernel<BWD, RY>), dim3((ntx * nty * nz + 7) / 8 * 8), dim3(SM_THREADS), 0, st, in, act,
                       out, D, H, W, k, txe, ntx, nty, nz);
  };
  const int blocks16 = ((W + 53) / 54) * ((H + 15) / 16) * nz;
  if (forced == 16 || (forced != 8 && blocks16 >= 256)) go(std::integral_constant<int, 2>{});
  else go(std::integral_constant<int, 1>{});
}

// ---- A10 -----------------------------------------------------------------------------
__global__ void __launch_bounds__(256) adam_kernel(float* __restrict__ x, float* __restrict__ m, float* __restrict__ v,
                                                   const float* __restrict__ g, int64_t n, float lr_t, float b1,
                                                   float b2, float eps) {
  const int64_t i4 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (i4 >= n) return;
  if (i4 + 3 < n) {
    float4 X = *reinterpret_cast<float4*>(x + i4), M = *reinterpret_cast<float4*>(m + i4),
           Vv = *reinterpret_cast<float4*>(v + i4);
    const float4 G = *reinterpret_cast<const float4*>(g + i4);
    float* xp = &X.x; float* mp = &M.x; float* vp = &Vv.x; const float* gp = &G.x;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      mp[j] = b1 * mp[j] + (1.f - b1) * gp[j];
      vp[j] = b2 * vp[j] + (1.f - b2) * gp[j] * gp[j];
      xp[j] -= lr_t * mp[j] / (sqrtf(vp[j]) + eps);
    }
    *reinterpret_cast<float4*>(x + i4) = X;
    *reinterpret_cast<float4*>(m + i4) = M;
    *reinterpret_cast<float4*>(v + i4) = Vv;
  } else {
    for (int64_t i = i4; i < n; ++i) {
      const float gi = g[i];
      const float mi = b1 * m[i] + (1.f - b1) * gi;
      const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
      m[i] = mi; v[i] = vi;
      x[i] -= lr_t * mi / (sqrtf(vi) + eps);
    }
  }
}

__global__ void __launch_bounds__(256) fill_kernel(float* __restrict__ x, float value, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) x[i] = value;
}
__global__ void __launch_bounds__(256) axpy_kernel(float* __restrict__ y, const float* __restrict__ x, float a,
                                                   int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) y[i] += a * x[i];
}

// send buffer of the D-slab reduce-scatter (nfs_hip.h: nfs_slab_pack): blockIdx.y = destination plane, 16 bytes a lane
__global__ void __launch_bounds__(256) slab_pack_kernel(const float4* __restrict__ gpad, float4* __restrict__ pack, int D,
                                                        int64_t plane4, int cs) {
  const int dp = blockIdx.y, k = dp / (cs + 5), j = dp - k * (cs + 5);
  const int sp = j == cs + 4 ? D + 4 : k * cs + j;            // source plane of gpad; >= D + 4 here: past the volume
  const bool zero = j != cs + 4 && sp >= D + 4;
  const float4* __restrict__ src = gpad + (int64_t)sp * plane4;
  float4* __restrict__ dst = pack + (int64_t)dp * plane4;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < plane4; i += (int64_t)gridDim.x * blockDim.x)
    dst[i] = zero ? make_float4(0.f, 0.f, 0.f, 0.f) : src[i];
}

// ---- 2-D colour stylizer: the elementwise links of its chain (styler_2p.py:68-102, 259-262) -----------------------------
// clip(c, 0, 1) read through the frame's grid order; the adjoints of the two clips (tf.clip_by_value passes the gradient
// where min <= x <= max); the iterate update g_opt += nan_to_num(x) - g_opt with the next step's variable in the same pass
__global__ void __launch_bounds__(256) colour_clamp_gather_kernel(const float* __restrict__ var,
                                                                  const long long* __restrict__ order,
                                                                  float* __restrict__ out, int64_t n, int C) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int64_t row = i / C, src = order ? (int64_t)order[row] * C + (i - row * C) : i;
  out[i] = fminf(fmaxf(var[src], 0.f), 1.f);
}
__global__ void __launch_bounds__(256) clamp01_bwd_kernel(const float* __restrict__ g, const float* __restrict__ x,
                                                          float* __restrict__ out, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = (x[i] >= 0.f && x[i] <= 1.f) ? g[i] : 0.f;
}
__global__ void __launch_bounds__(256) colour_clamp_scatter_bwd_kernel(const float* __restrict__ g_cc,
                                                                       const long long* __restrict__ order,
                                                                       const float* __restrict__ var,
                                                                       float* __restrict__ g_var, int64_t n, int C) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int64_t row = i / C, dst = order ? (int64_t)order[row] * C + (i - row * C) : i;
  const float v = var[dst];
  g_var[dst] = (v >= 0.f && v <= 1.f) ? g_cc[i] : 0.f;
}
__global__ void __launch_bounds__(256) iterate_update_kernel(float* __restrict__ x, float* __restrict__ g_opt, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float v = x[i];
  v = v != v ? 0.f : fminf(fmaxf(v, -3.4028234663852886e38f), 3.4028234663852886e38f);      // np.nan_to_num
  const float g = g_opt[i];
  const float d = v - g;
  const float r = g + d;                     // (the reference's two roundings: g_tmp = new - old, g_opt += g_tmp)
  g_opt[i] = r;
  x[i] = r;
}

// ---- A5 ------------------------------------------------------------------------------
__constant__ float kVggMean[3] = {0.485f * 255.f, 0.456f * 255.f, 0.406f * 255.f};  // vgg.py:18-20

struct Lerp { int i0, i1; float w; };
__device__ __forceinline__ Lerp tf1_lerp(int dst, int n_in, int n_out) {
  // legacy tf.image.resize: src = dst * in/out (no half-pixel centres), i1 = min(i0+1, in-1)
  const float scale = (float)n_in / (float)n_out;
  const float s = (float)dst * scale;
  Lerp l;
  l.i0 = (int)floorf(s);
  l.i1 = min(l.i0 + 1, n_in - 1);
  l.w = s - (float)l.i0;
  return l;
}

__global__ void __launch_bounds__(256) loss_net_input_fwd_kernel(const float* __restrict__ img,
                                                                 float* __restrict__ d_img, float* __restrict__ xo,
                                                                 int B, int H, int W, int Cin, int H2, int W2) {
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= (int64_t)B * H2 * W2) return;
  const int x2 = (int)(gid % W2);
  const int y2 = (int)((gid / W2) % H2);
  const int b = (int)(gid / ((int64_t)W2 * H2));
  const float* im = img + (int64_t)b * H * W * Cin;
  float val[3];
  const bool resize = (H2 != H) || (W2 != W);
  for (int c = 0; c < Cin; ++c) {
    float r;
    if (!resize) {
      r = im[((int64_t)y2 * W + x2) * Cin + c];
    } else {
      const Lerp ly = tf1_lerp(y2, H, H2), lx = tf1_lerp(x2, W, W2);
      const float tl = im[((int64_t)ly.i0 * W + lx.i0) * Cin + c], tr = im[((int64_t)ly.i0 * W + lx.i1) * Cin + c];
      const float bl = im[((int64_t)ly.i1 * W + lx.i0) * Cin + c], br = im[((int64_t)ly.i1 * W + lx.i1) * Cin + c];
      const float top = tl + (tr - tl) * lx.w, bot = bl + (br - bl) * lx.w;
      r = top + (bot - top) * ly.w;
    }
    val[c] = r * 255.f;
  }
  if (Cin == 1) val[1] = val[2] = val[0];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    if (d_img) d_img[gid * 3 + c] = val[c];
    if (xo) xo[gid * 3 + c] = val[c] - kVggMean[c];
  }
}

__global__ void __launch_bounds__(256) loss_net_input_bwd_kernel(const float* __restrict__ g_x,
                                                                 float* __restrict__ g_img, int B, int H, int W,
                                                                 int Cin, int H2, int W2) {
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= (int64_t)B * H2 * W2) return;
  const int x2 = (int)(gid % W2);
  const int y2 = (int)((gid / W2) % H2);
  const int b = (int)(gid / ((int64_t)W2 * H2));
  float g[3] = {g_x[gid * 3] * 255.f, g_x[gid * 3 + 1] * 255.f, g_x[gid * 3 + 2] * 255.f};
  if (Cin == 1) g[0] = g[0] + g[1] + g[2];
  float* gi = g_img + (int64_t)b * H * W * Cin;
  const bool resize = (H2 != H) || (W2 != W);
  if (!resize) {
    for (int c = 0; c < Cin; ++c) gi[((int64_t)y2 * W + x2) * Cin + c] = g[c];
    return;
  }
  const Lerp ly = tf1_lerp(y2, H, H2), lx = tf1_lerp(x2, W, W2);
  for (int c = 0; c < Cin; ++c) {
    atomicAdd(gi + ((int64_t)ly.i0 * W + lx.i0) * Cin + c, g[c] * (1.f - ly.w) * (1.f - lx.w));
    atomicAdd(gi + ((int64_t)ly.i0 * W + lx.i1) * Cin + c, g[c] * (1.f - ly.w) * lx.w);
    atomicAdd(gi + ((int64_t)ly.i1 * W + lx.i0) * Cin + c, g[c] * ly.w * (1.f - lx.w));
    atomicAdd(gi + ((int64_t)ly.i1 * W + lx.i1) * Cin + c, g[c] * ly.w * lx.w);
  }
}

// ---- A6 pool -------------------------------------------------------------------------
// channels-last, 4 channels (one float4) per thread
__global__ void __launch_bounds__(256) avgpool2_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int B,
                                                           int H, int W, int C4) {
  const int Ho = H >> 1, Wo = W >> 1;
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= (int64_t)B * Ho * Wo * C4) return;
  const int c = (int)(gid % C4);
  const int j = (int)((gid / C4) % Wo);
  const int i = (int)((gid / ((int64_t)C4 * Wo)) % Ho);
  const int b = (int)(gid / ((int64_t)C4 * Wo * Ho));
  const float4* xp = reinterpret_cast<const float4*>(x) + (((int64_t)b * H + 2 * i) * W + 2 * j) * C4 + c;
  const float4 a = xp[0], bq = xp[C4], cq = xp[(int64_t)W * C4], dq = xp[(int64_t)W * C4 + C4];
  float4 r;
  r.x = (a.x + bq.x + cq.x + dq.x) * 0.25f;
  r.y = (a.y + bq.y + cq.y + dq.y) * 0.25f;
  r.z = (a.z + bq.z + cq.z + dq.z) * 0.25f;
  r.w = (a.w + bq.w + cq.w + dq.w) * 0.25f;
  reinterpret_cast<float4*>(y)[gid] = r;
}

__global__ void __launch_bounds__(256) avgpool2_bwd_kernel(const float* __restrict__ gy, const float* __restrict__ x,
                                                           const float* __restrict__ addend, float* __restrict__ gx,
                                                           int B, int H, int W, int C4) {
  const int Ho = H >> 1, Wo = W >> 1;
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= (int64_t)B * H * W * C4) return;
  const int c = (int)(gid % C4);
  const int w = (int)((gid / C4) % W);
  const int h = (int)((gid / ((int64_t)C4 * W)) % H);
  const int b = (int)(gid / ((int64_t)C4 * W * H));
  float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
  if ((h >> 1) < Ho && (w >> 1) < Wo) {
    const float4 t = reinterpret_cast<const float4*>(gy)[(((int64_t)b * Ho + (h >> 1)) * Wo + (w >> 1)) * C4 + c];
    g = make_float4(t.x * 0.25f, t.y * 0.25f, t.z * 0.25f, t.w * 0.25f);
  }
  if (x) {
    const float4 xv = reinterpret_cast<const float4*>(x)[gid];
    g.x = xv.x > 0.f ? g.x : 0.f; g.y = xv.y > 0.f ? g.y : 0.f;
    g.z = xv.z > 0.f ? g.z : 0.f; g.w = xv.w > 0.f ? g.w : 0.f;
  }
  if (addend) {
    const float4 a = reinterpret_cast<const float4*>(addend)[gid];
    g.x += a.x; g.y += a.y; g.z += a.z; g.w += a.w;
  }
  reinterpret_cast<float4*>(gx)[gid] = g;
}

// ---- A12 -----------------------------------------------------------------------------
__device__ __forceinline__ float sgn(float v) { return (v > 0.f) ? 1.f : (v < 0.f ? -1.f : 0.f); }

__global__ void __launch_bounds__(256) tv_kernel(const float* __restrict__ x, float* __restrict__ loss,
                                                 float* __restrict__ g, int B, int H, int W, int C, float scale) {
  __shared__ float red[16];
  const int64_t n = (int64_t)B * H * W * C;
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  float part = 0.f;
  if (gid < n) {
    const int w = (int)((gid / C) % W);
    const int h = (int)((gid / ((int64_t)C * W)) % H);
    const float v = x[gid];
    float gv = 0.f;
    if (h + 1 < H) { const float dlt = x[gid + (int64_t)W * C] - v; part += fabsf(dlt); gv -= sgn(dlt); }
    if (w + 1 < W) { const float dlt = x[gid + C] - v; part += fabsf(dlt); gv -= sgn(dlt); }
    if (h > 0) gv += sgn(v - x[gid - (int64_t)W * C]);
    if (w > 0) gv += sgn(v - x[gid - C]);
    if (g) g[gid] += scale * gv;
  }
  part = block_sum(part, red);
  if (threadIdx.x == 0 && part != 0.f) atomicAdd(loss, part * scale);
}

}  // namespace nfs

using namespace nfs;

extern "C" {

int nfs_smooth3d_relu_fwd(const float* d, float* out, int D, int H, int W, float k, nfs_stream_t stream) {
  NFS_REQUIRE(d && out, "nfs_smooth3d_relu_fwd: null pointer");
  NFS_REQUIRE(D > 0 && H > 0 && W > 0, "nfs_smooth3d_relu_fwd: non-positive dimension");
  NFS_REQUIRE((int64_t)H * W * 4 * (SM_ZCHUNK + 2) < ((int64_t)1 << 31), "nfs_smooth3d_relu_fwd: a z-chunk of planes must stay below 2 GB (32-bit buffer offsets)");
  launch_smooth3d<false>(d, nullptr, out, D, H, W, k, as_stream(stream));
  return check_launch("nfs_smooth3d_relu_fwd");
}

int nfs_smooth3d_relu_bwd(const float* out, const float* g_out, float* g_d, int D, int H, int W, float k,
                          nfs_stream_t stream) {
  NFS_REQUIRE(out && g_out && g_d, "nfs_smooth3d_relu_bwd: null pointer");
  NFS_REQUIRE(D > 0 && H > 0 && W > 0, "nfs_smooth3d_relu_bwd: non-positive dimension");
  NFS_REQUIRE((int64_t)H * W * 4 * (SM_ZCHUNK + 2) < ((int64_t)1 << 31), "nfs_smooth3d_relu_bwd: a z-chunk of planes must stay below 2 GB (32-bit buffer offsets)");
  launch_smooth3d<true>(g_out, out, g_d, D, H, W, k, as_stream(stream));
  return check_launch("nfs_smooth3d_relu_bwd");
}

int nfs_adam_tf_step(float* x, float* m, float* v, const float* g, int64_t n, float lr_t, float beta1, float beta2,
                     float eps, nfs_stream_t stream) {
  NFS_REQUIRE(x && m && v && g, "nfs_adam_tf_step: null pointer");
  NFS_REQUIRE(n > 0, "nfs_adam_tf_step: n must be positive");
  NFS_REQUIRE((((uintptr_t)x | (uintptr_t)m | (uintptr_t)v | (uintptr_t)g) & 15) == 0,
              "nfs_adam_tf_step: pointers must be 16-byte aligned");
  hipLaunchKernelGGL(adam_kernel, dim3(blocks_for((n + 3) / 4, 256)), dim3(256), 0, as_stream(stream), x, m, v, g, n,
                     lr_t, beta1, beta2, eps);
  return check_launch("nfs_adam_tf_step");
}

int nfs_fill(float* x, float value, int64_t n, nfs_stream_t stream) {
  NFS_REQUIRE(x && n > 0, "nfs_fill: bad argument");
  hipLaunchKernelGGL(fill_kernel, dim3(blocks_for(n, 256)), dim3(256), 0, as_stream(stream), x, value, n);
  return check_launch("nfs_fill");
}

int nfs_axpy(float* y, const float* x, float a, int64_t n, nfs_stream_t stream) {
  NFS_REQUIRE(x && y && n > 0, "nfs_axpy: bad argument");
  hipLaunchKernelGGL(axpy_kernel, dim3(blocks_for(n, 256)), dim3(256), 0, as_stream(stream), y, x, a, n);
  return check_launch("nfs_axpy");
}

int nfs_colour_clamp_gather(const float* var, const long long* order, float* out, int64_t N, int C, nfs_stream_t stream) {
  NFS_REQUIRE(var && out && N > 0 && C > 0, "nfs_colour_clamp_gather: bad argument");
  hipLaunchKernelGGL(colour_clamp_gather_kernel, dim3(blocks_for(N * C, 256)), dim3(256), 0, as_stream(stream), var, order,
                     out, N * C, C);
  return check_launch("nfs_colour_clamp_gather");
}

int nfs_clamp01_bwd(const float* g, const float* x, float* out, int64_t n, nfs_stream_t stream) {
  NFS_REQUIRE(g && x && out && n > 0, "nfs_clamp01_bwd: bad argument");
  hipLaunchKernelGGL(clamp01_bwd_kernel, dim3(blocks_for(n, 256)), dim3(256), 0, as_stream(stream), g, x, out, n);
  return check_launch("nfs_clamp01_bwd");
}

int nfs_colour_clamp_scatter_bwd(const float* g_cc, const long long* order, const float* var, float* g_var, int64_t N,
                                 int C, nfs_stream_t stream) {
  NFS_REQUIRE(g_cc && var && g_var && N > 0 && C > 0, "nfs_colour_clamp_scatter_bwd: bad argument");
  hipLaunchKernelGGL(colour_clamp_scatter_bwd_kernel, dim3(blocks_for(N * C, 256)), dim3(256), 0, as_stream(stream), g_cc,
                     order, var, g_var, N * C, C);
  return check_launch("nfs_colour_clamp_scatter_bwd");
}

int nfs_iterate_update(float* x, float* g_opt, int64_t n, nfs_stream_t stream) {
  NFS_REQUIRE(x && g_opt && n > 0, "nfs_iterate_update: bad argument");
  hipLaunchKernelGGL(iterate_update_kernel, dim3(blocks_for(n, 256)), dim3(256), 0, as_stream(stream), x, g_opt, n);
  return check_launch("nfs_iterate_update");
}

int nfs_slab_pack(const float* gpad, float* pack, int D, int64_t plane, int world, int cs, nfs_stream_t stream) {
  NFS_REQUIRE(gpad && pack, "nfs_slab_pack: null pointer");
  NFS_REQUIRE(D > 0 && plane > 0 && world > 0 && cs > 0, "nfs_slab_pack: non-positive dimension");
  NFS_REQUIRE(plane % 4 == 0 && ((uintptr_t)gpad & 15) == 0 && ((uintptr_t)pack & 15) == 0,
              "nfs_slab_pack: plane size must be a multiple of 4 floats and the buffers 16-byte aligned");
  NFS_REQUIRE((int64_t)world * (cs + 5) <= 65535, "nfs_slab_pack: more than 65535 packed planes");
  const int64_t plane4 = plane / 4;
  const unsigned bx = (unsigned)((plane4 + 255) / 256 < 64 ? (plane4 + 255) / 256 : 64);
  hipLaunchKernelGGL(slab_pack_kernel, dim3(bx, (unsigned)(world * (cs + 5))), dim3(256), 0, as_stream(stream),
                     reinterpret_cast<const float4*>(gpad), reinterpret_cast<float4*>(pack), D, plane4, cs);
  return check_launch("nfs_slab_pack");
}

int nfs_loss_net_input_fwd(const float* img, float* d_img, float* x, int B, int H, int W, int Cin, int H2, int W2,
                           nfs_stream_t stream) {
  NFS_REQUIRE(img && (d_img || x), "nfs_loss_net_input_fwd: null pointer");
  NFS_REQUIRE(B > 0 && H > 0 && W > 0 && H2 > 0 && W2 > 0, "nfs_loss_net_input_fwd: non-positive dimension");
  NFS_REQUIRE(Cin == 1 || Cin == 3, "nfs_loss_net_input_fwd: Cin must be 1 or 3");
  const int64_t n = (int64_t)B * H2 * W2;
  hipLaunchKernelGGL(loss_net_input_fwd_kernel, dim3(blocks_for(n, 256)), dim3(256), 0, as_stream(stream), img, d_img,
                     x, B, H, W, Cin, H2, W2);
  return check_launch("nfs_loss_net_input_fwd");
}

int nfs_loss_net_input_bwd(const float* g_x, float* g_img, int B, int H, int W, int Cin, int H2, int W2,
                           nfs_stream_t stream) {
  NFS_REQUIRE(g_x && g_img, "nfs_loss_net_input_bwd: null pointer");
  NFS_REQUIRE(B > 0 && H > 0 && W > 0 && H2 > 0 && W2 > 0, "nfs_loss_net_input_bwd: non-positive dimension");
  NFS_REQUIRE(Cin == 1 || Cin == 3, "nfs_loss_net_input_bwd: Cin must be 1 or 3");
  if (H2 != H || W2 != W) {
    zero_words(g_img, (long long)B * H * W * Cin, as_stream(stream));     // (a kernel, not a memset node: common.h)
  }
  const int64_t n = (int64_t)B * H2 * W2;
  hipLaunchKernelGGL(loss_net_input_bwd_kernel, dim3(blocks_for(n, 256)), dim3(256), 0, as_stream(stream), g_x, g_img,
                     B, H, W, Cin, H2, W2);
  return check_launch("nfs_loss_net_input_bwd");
}

int nfs_avgpool2_fwd(const float* x, float* y, int B, int H, int W, int C, nfs_stream_t stream) {
  NFS_REQUIRE(x && y, "nfs_avgpool2_fwd: null pointer");
  NFS_REQUIRE(B > 0 && H > 1 && W > 1 && C > 0 && C % 4 == 0, "nfs_avgpool2_fwd: need H,W >= 2 and C %% 4 == 0");
  const int64_t n = (int64_t)B * (H / 2) * (W / 2) * (C / 4);
  hipLaunchKernelGGL(avgpool2_fwd_kernel, dim3(blocks_for(n, 256)), dim3(256), 0, as_stream(stream), x, y, B, H, W,
                     C / 4);
  return check_launch("nfs_avgpool2_fwd");
}

int nfs_avgpool2_bwd(const float* gy, const float* x, const float* addend, float* gx, int B, int H, int W, int C,
                     nfs_stream_t stream) {
  NFS_REQUIRE(gy && gx, "nfs_avgpool2_bwd: null pointer");
  NFS_REQUIRE(B > 0 && H > 1 && W > 1 && C > 0 && C % 4 == 0, "nfs_avgpool2_bwd: need H,W >= 2 and C %% 4 == 0");
  const int64_t n = (int64_t)B * H * W * (C / 4);
  hipLaunchKernelGGL(avgpool2_bwd_kernel, dim3(blocks_for(n, 256)), dim3(256), 0, as_stream(stream), gy, x, addend, gx,
                     B, H, W, C / 4);
  return check_launch("nfs_avgpool2_bwd");
}

int nfs_tv_loss(const float* d_img, float* loss_acc, float* g_acc, int B, int H, int W, int C, float weight,
                nfs_stream_t stream) {
  NFS_REQUIRE(d_img && loss_acc, "nfs_tv_loss: null pointer");
  NFS_REQUIRE(B > 0 && H > 0 && W > 0 && C > 0, "nfs_tv_loss: non-positive dimension");
  const int64_t n = (int64_t)B * H * W * C;
  hipLaunchKernelGGL(tv_kernel, dim3(blocks_for(n, 256)), dim3(256), 0, as_stream(stream), d_img, loss_acc, g_acc, B, H,
                     W, C, weight / (float)B);
  return check_launch("nfs_tv_loss");
}

}  // extern "C"
