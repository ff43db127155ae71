// Micro-benchmark behind the round-4 field-work fusion: where do the 69 us of the scalar advect
// forward (160 MB algorithmic: 0.29 of HBM) go, and what does advect + smooth/clamp in ONE kernel cost?
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/micro/field_variants.hip -o tools/micro/field_variants
//   tools/micro/field_variants [G=200]
// Synthetic fields on the device (smooth blobs, sine velocity of +-2 cells); every variant is checked against variant A.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <vector>
#include <functional>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

struct __attribute__((packed, aligned(4))) F3u { float x, y, z; };
struct __attribute__((packed, aligned(4))) F2u { float x, y; };

__global__ void init_kernel(float* d, float* vel, int G) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= G * G * G) return;
  const int x = i % G, y = (i / G) % G, z = i / (G * G);
  const float fx = (float)x / G, fy = (float)y / G, fz = (float)z / G;
  float v = __expf(-40.f * ((fx - .5f) * (fx - .5f) + (fy - .45f) * (fy - .45f) + (fz - .55f) * (fz - .55f))) +
            0.6f * __expf(-90.f * ((fx - .3f) * (fx - .3f) + (fy - .6f) * (fy - .6f) + (fz - .4f) * (fz - .4f)));
  d[i] = v < 0.02f ? 0.f : fminf(v, 1.f);
  const float a = 4.f / (G - 1);      // 2 cells in normalised units
  vel[3 * i + 0] = a * __sinf(6.28f * (2 * fx + 3 * fy + fz));
  vel[3 * i + 1] = a * __sinf(6.28f * (3 * fx - fy + 2 * fz) + 1.f);
  vel[3 * i + 2] = a * __cosf(6.28f * (fx + 2 * fy - 3 * fz));
}

// ---- A..E: the product's advect1_kernel<0> with switches ---------------------------------------------------------------------
template <bool REMAP, int GATHER>   // GATHER 0: none (streams only), 1: four 8-byte loads, 2: eight 4-byte loads
__global__ void __launch_bounds__(256) advect_v4(const float* __restrict__ d, const float* vel, float* out, int D, int H, int W) {
  const int n = D * H * W;
  const int lane = threadIdx.x & 63;
  const unsigned per_xcd = gridDim.x / 8;
  const unsigned lb = !REMAP ? blockIdx.x : (blockIdx.x % 8) * per_xcd + blockIdx.x / 8;
  const int first = (lb * blockDim.x + (threadIdx.x - lane)) * 4 + lane;
  if (first - lane >= n) return;
  const F3u* v3 = reinterpret_cast<const F3u*>(vel);
  F3u vv[4];
  bool ok[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int idx = first + 64 * j;
    ok[j] = idx < n;
    vv[j] = v3[ok[j] ? idx : n - 1];
  }
  const float hz = 0.5f * (float)(D - 1), hy = 0.5f * (float)(H - 1), hx = 0.5f * (float)(W - 1);
  const float nz1 = (float)(D - 1), ny1 = (float)(H - 1), nx1 = (float)(W - 1);
  const unsigned uW = (unsigned)W, uHW = (unsigned)(H * W);
  const int f0 = min(first, n - 1);
  int w = f0 % W;
  const int t2 = f0 / W;
  int h = t2 % H, z = t2 / H;
  F2u p[4][4];
  float wz[4], wy[4], wx[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float xz = fmaf(-vv[j].x, hz, (float)z), xy = fmaf(-vv[j].y, hy, (float)h), xx = fmaf(-vv[j].z, hx, (float)w);
    const float cz = __builtin_amdgcn_fmed3f(xz, 0.f, nz1), cy = __builtin_amdgcn_fmed3f(xy, 0.f, ny1),
                cx = __builtin_amdgcn_fmed3f(xx, 0.f, nx1);
    const float bz = fminf(floorf(cz), nz1 - 1.f), by = fminf(floorf(cy), ny1 - 1.f), bx = fminf(floorf(cx), nx1 - 1.f);
    wz[j] = cz - bz; wy[j] = cy - by; wx[j] = cx - bx;
    const unsigned o = (unsigned)(int)bz * uHW + (unsigned)(int)by * uW + (unsigned)(int)bx;
    if (GATHER == 1) {
      p[j][0] = *reinterpret_cast<const F2u*>(d + o);
      p[j][1] = *reinterpret_cast<const F2u*>(d + o + uW);
      p[j][2] = *reinterpret_cast<const F2u*>(d + o + uHW);
      p[j][3] = *reinterpret_cast<const F2u*>(d + o + uHW + uW);
    } else if (GATHER == 2) {
      p[j][0] = F2u{d[o], d[o + 1]};
      p[j][1] = F2u{d[o + uW], d[o + uW + 1]};
      p[j][2] = F2u{d[o + uHW], d[o + uHW + 1]};
      p[j][3] = F2u{d[o + uHW + uW], d[o + uHW + uW + 1]};
    } else {
      const float q = (float)o;
      p[j][0] = F2u{q, q}; p[j][1] = F2u{q, wz[j]}; p[j][2] = F2u{wy[j], q}; p[j][3] = F2u{q, wx[j]};
    }
    w += 64;
    while (w >= W) { w -= W; if (++h == H) { h = 0; if (z < D - 1) ++z; } }
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float a00 = fmaf(wx[j], p[j][0].y - p[j][0].x, p[j][0].x), a01 = fmaf(wx[j], p[j][1].y - p[j][1].x, p[j][1].x);
    const float a10 = fmaf(wx[j], p[j][2].y - p[j][2].x, p[j][2].x), a11 = fmaf(wx[j], p[j][3].y - p[j][3].x, p[j][3].x);
    const float b0 = fmaf(wy[j], a01 - a00, a00), b1 = fmaf(wy[j], a11 - a10, a10);
    if (ok[j]) out[first + 64 * j] = fmaf(wz[j], b1 - b0, b0);
  }
}

// one voxel per lane, plain indexing
template <bool REMAP>
__global__ void __launch_bounds__(256) advect_v1(const float* __restrict__ d, const float* vel, float* out, int D, int H, int W) {
  const int n = D * H * W;
  const unsigned per_xcd = gridDim.x / 8;
  const unsigned lb = !REMAP ? blockIdx.x : (blockIdx.x % 8) * per_xcd + blockIdx.x / 8;
  const int idx = lb * 256 + threadIdx.x;
  if (idx >= n) return;
  const F3u v = reinterpret_cast<const F3u*>(vel)[idx];
  const int w = idx % W, t2 = idx / W, h = t2 % H, z = t2 / H;
  const float hz = 0.5f * (float)(D - 1), hy = 0.5f * (float)(H - 1), hx = 0.5f * (float)(W - 1);
  const float nz1 = (float)(D - 1), ny1 = (float)(H - 1), nx1 = (float)(W - 1);
  const float cz = __builtin_amdgcn_fmed3f(fmaf(-v.x, hz, (float)z), 0.f, nz1),
              cy = __builtin_amdgcn_fmed3f(fmaf(-v.y, hy, (float)h), 0.f, ny1),
              cx = __builtin_amdgcn_fmed3f(fmaf(-v.z, hx, (float)w), 0.f, nx1);
  const float bz = fminf(floorf(cz), nz1 - 1.f), by = fminf(floorf(cy), ny1 - 1.f), bx = fminf(floorf(cx), nx1 - 1.f);
  const float wz = cz - bz, wy = cy - by, wx = cx - bx;
  const unsigned uW = W, uHW = H * W, o = (unsigned)(int)bz * uHW + (unsigned)(int)by * uW + (unsigned)(int)bx;
  const F2u p0 = *reinterpret_cast<const F2u*>(d + o), p1 = *reinterpret_cast<const F2u*>(d + o + uW),
            p2 = *reinterpret_cast<const F2u*>(d + o + uHW), p3 = *reinterpret_cast<const F2u*>(d + o + uHW + uW);
  const float a00 = fmaf(wx, p0.y - p0.x, p0.x), a01 = fmaf(wx, p1.y - p1.x, p1.x);
  const float a10 = fmaf(wx, p2.y - p2.x, p2.x), a11 = fmaf(wx, p3.y - p3.x, p3.x);
  const float b0 = fmaf(wy, a01 - a00, a00), b1 = fmaf(wy, a11 - a10, a10);
  out[idx] = fmaf(wz, b1 - b0, b0);
}

// streaming baseline: read 12 B + write 4 B per voxel
__global__ void __launch_bounds__(256) stream_kernel(const float* vel, float* out, int n) {
  const int lane = threadIdx.x & 63;
  const int first = (blockIdx.x * blockDim.x + (threadIdx.x - lane)) * 4 + lane;
  const F3u* v3 = reinterpret_cast<const F3u*>(vel);
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int idx = first + 64 * j;
    if (idx < n) { const F3u v = v3[idx]; out[idx] = v.x + v.y + v.z; }
  }
}

// ---- reference smooth (3x3x3 separable [1,k,1]/(k+2), zero padding) + clamp, naive: for checking only -------------------
__global__ void smooth_ref_kernel(const float* in, float* out, int D, int H, int W, float k) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= D * H * W) return;
  const int x = i % W, y = (i / W) % H, z = i / (W * H);
  const float inv = 1.f / (k + 2.f), wgt[3] = {inv, k * inv, inv};
  float acc = 0.f;
  for (int dz = -1; dz <= 1; ++dz)
    for (int dy = -1; dy <= 1; ++dy)
      for (int dx = -1; dx <= 1; ++dx) {
        const int zz = z + dz, yy = y + dy, xx = x + dx;
        if (zz < 0 || zz >= D || yy < 0 || yy >= H || xx < 0 || xx >= W) continue;
        acc += wgt[dz + 1] * wgt[dy + 1] * wgt[dx + 1] * in[(zz * H + yy) * W + xx];
      }
  out[i] = fmaxf(acc, 0.f);
}

// ---- F: advect + smooth + clamp in one kernel, z-march ----------------------------------------------------------------------
// block = 512 threads = 8 waves; interior tile 16 rows x txe columns (thread (wv, tx) owns rows wv and wv + 8 of column tx),
// halo'd tile 18 x (txe + 2) <= 1024 elements: two staging slots per thread.  Per plane: the two staged elements are ADVECT
// SAMPLES (velocity load -> four 8-byte corner gathers -> trilinear), zero outside the volume.  Software pipeline over
// planes: velocity loads LV planes ahead of the gather issue, gathers LG planes ahead of their use.
constexpr uint32_t OOB = 0x80000000u;
template <int LV, int LG>
__global__ void __launch_bounds__(512) advect_smooth_kernel(const float* __restrict__ d, const float* __restrict__ vel,
                                                            float* __restrict__ out, int D, int H, int W, float k, int txe,
                                                            int ntx, int nty, int nz, int zchunk) {
  constexpr int TYI = 16, TXM = 64, TB = (TYI + 2) * (TXM + 2);
  __shared__ float tile[2][TYI + 2][TXM + 2];
  __shared__ float dump[TB + 2];
  const int t = threadIdx.x, tx = t & 63, wv = t >> 6;
  const unsigned per_xcd = gridDim.x / 8;
  const unsigned lb = (blockIdx.x % 8) * per_xcd + blockIdx.x / 8;
  if (lb >= (unsigned)(ntx * nty * nz)) return;
  const int bx = lb % ntx, by = (lb / ntx) % nty, bz = lb / (ntx * nty);
  const int x0 = bx * txe, y0 = by * TYI, z0 = bz * zchunk, z1 = min(z0 + zchunk, D);
  const float inv = k > 0.f ? 1.f / (k + 2.f) : 1.f;
  const float wa = k > 0.f ? inv : 0.f, wb = k > 0.f ? k * inv : 1.f;
  // staging slots: elements t and t + 512 of the halo'd tile, row-major with cols = txe + 2
  const int cols = txe + 2, ne = (TYI + 2) * cols;
  int sy[2], sx[2];
  bool sin_[2];
  float* sp[2];
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    const int e = t + 512 * s, r = e / cols, c = e - r * cols;
    sy[s] = y0 - 1 + r; sx[s] = x0 - 1 + c;
    sin_[s] = e < ne && sy[s] >= 0 && sy[s] < H && sx[s] >= 0 && sx[s] < W;
    sp[s] = e < ne ? &tile[0][r][c] : dump + s;
  }
  const float hz = 0.5f * (float)(D - 1), hy = 0.5f * (float)(H - 1), hx = 0.5f * (float)(W - 1);
  const float nz1 = (float)(D - 1), ny1 = (float)(H - 1), nx1 = (float)(W - 1);
  const unsigned uW = (unsigned)W, uHW = (unsigned)(H * W);
  const __amdgpu_buffer_rsrc_t v_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(vel), 0, (uint32_t)((size_t)D * H * W * 12), 0x00020000);
  const __amdgpu_buffer_rsrc_t d_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(d), 0, (uint32_t)((size_t)D * H * W * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t o_rsrc = __builtin_amdgcn_make_buffer_rsrc(out, 0, (uint32_t)((size_t)D * H * W * 4), 0x00020000);
  uint32_t voff[2];
#pragma unroll
  for (int s = 0; s < 2; ++s) voff[s] = sin_[s] ? (uint32_t)(sy[s] * W + sx[s]) * 12u : OOB;
  typedef float f3 __attribute__((ext_vector_type(3)));
  typedef float f2 __attribute__((ext_vector_type(2)));
  // rings
  f3 rv[LV][2];            // velocities of planes (issued LV iterations before the gathers)
  f2 rc[LG][2][4];         // corner pairs
  float rw[LG][2][3];      // weights
  auto load_vel = [&](int pz, f3* dst) {
    const bool in_ = pz >= 0 && pz < D;
#pragma unroll
    for (int s = 0; s < 2; ++s)
      dst[s] = __builtin_bit_cast(f3, __builtin_amdgcn_raw_buffer_load_b96(v_rsrc, in_ ? voff[s] : OOB, in_ ? (uint32_t)pz * uHW * 12u : 0u, 0));
  };
  auto gather = [&](int pz, const f3* v, f2 (*c)[4], float (*wgt)[3]) {
    const bool in_ = pz >= 0 && pz < D;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const float xz = fmaf(-v[s].x, hz, (float)pz), xy = fmaf(-v[s].y, hy, (float)sy[s]), xx = fmaf(-v[s].z, hx, (float)sx[s]);
      const float cz = __builtin_amdgcn_fmed3f(xz, 0.f, nz1), cy = __builtin_amdgcn_fmed3f(xy, 0.f, ny1),
                  cx = __builtin_amdgcn_fmed3f(xx, 0.f, nx1);
      const float bz_ = fminf(floorf(cz), nz1 - 1.f), by_ = fminf(floorf(cy), ny1 - 1.f), bx_ = fminf(floorf(cx), nx1 - 1.f);
      wgt[s][0] = cz - bz_; wgt[s][1] = cy - by_; wgt[s][2] = cx - bx_;
      const uint32_t o = (in_ && sin_[s]) ? ((unsigned)(int)bz_ * uHW + (unsigned)(int)by_ * uW + (unsigned)(int)bx_) * 4u : OOB;
      c[s][0] = __builtin_bit_cast(f2, __builtin_amdgcn_raw_buffer_load_b64(d_rsrc, o, 0, 0));
      c[s][1] = __builtin_bit_cast(f2, __builtin_amdgcn_raw_buffer_load_b64(d_rsrc, o, uW * 4u, 0));
      c[s][2] = __builtin_bit_cast(f2, __builtin_amdgcn_raw_buffer_load_b64(d_rsrc, o, uHW * 4u, 0));
      c[s][3] = __builtin_bit_cast(f2, __builtin_amdgcn_raw_buffer_load_b64(d_rsrc, o, (uHW + uW) * 4u, 0));
    }
  };
  auto interp = [&](const f2 (*c)[4], const float (*wgt)[3], int s) {
    const float a00 = fmaf(wgt[s][2], c[s][0].y - c[s][0].x, c[s][0].x), a01 = fmaf(wgt[s][2], c[s][1].y - c[s][1].x, c[s][1].x);
    const float a10 = fmaf(wgt[s][2], c[s][2].y - c[s][2].x, c[s][2].x), a11 = fmaf(wgt[s][2], c[s][3].y - c[s][3].x, c[s][3].x);
    const float b0 = fmaf(wgt[s][1], a01 - a00, a00), b1 = fmaf(wgt[s][1], a11 - a10, a10);
    return fmaf(wgt[s][0], b1 - b0, b0);       // (OOB corners read as zeros: the sample is zero)
  };
  // prologue: plane q is staged into LDS at iteration q - 1 (consumed at iteration q); first consumed plane is z0 - 1.
  // velocity of plane q is loaded LV + LG iterations before it is staged, its gathers LG iterations before.
  const int pf = z0 - 1;
#pragma unroll
  for (int u = 0; u < LV; ++u) load_vel(pf + LG + u, rv[u]);          // planes pf+LG .. pf+LG+LV-1 wait for their gathers
  {
    f3 v0[2];
#pragma unroll
    for (int u = 0; u < LG; ++u) {                                      // planes pf .. pf+LG-1: gathers now
      load_vel(pf + u, v0);
      gather(pf + u, v0, rc[u], rw[u]);
    }
  }
  // stage plane pf
  {
    sp[0][0] = interp(rc[0], rw[0], 0);
    sp[1][0] = interp(rc[0], rw[0], 1);
    // its ring slot is refilled with plane pf + LG
    gather(pf + LG, rv[0], rc[0], rw[0]);
    load_vel(pf + LG + LV, rv[0]);
  }
  __syncthreads();
  float pm[2] = {0.f, 0.f}, pc[2] = {0.f, 0.f};
  const int oy0 = y0 + wv, oy1 = y0 + wv + 8, ox = x0 + tx;
  const bool own0 = tx < txe && ox < W && oy0 < H, own1 = tx < txe && ox < W && oy1 < H;
  const uint32_t oo0 = own0 ? (uint32_t)(oy0 * W + ox) * 4u : OOB, oo1 = own1 ? (uint32_t)(oy1 * W + ox) * 4u : OOB;
  constexpr int LC = LV * LG * 2;    // unroll period: ring slots and the LDS buffer parity are compile-time
  for (int pb = pf; pb <= z1; pb += LC) {
#pragma unroll
    for (int u = 0; u < LC; ++u) {
      const int p = pb + u;
      if (p > z1) break;
      const int b = u & 1;
      float pn[2];
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        const float* t0 = &tile[b][wv + 8 * r][tx];
        const float ra = wa * t0[0] + wb * t0[1] + wa * t0[2];
        const float rb = wa * t0[TXM + 2] + wb * t0[TXM + 3] + wa * t0[TXM + 4];
        const float rcc = wa * t0[2 * (TXM + 2)] + wb * t0[2 * (TXM + 2) + 1] + wa * t0[2 * (TXM + 2) + 2];
        pn[r] = wa * ra + wb * rb + wa * rcc;
      }
      // stage plane p + 1 (ring slot (u + 1) % LG holds its corners), refill the slot with plane p + 1 + LG
      const int gs = (u + 1) % LG, vs = (u + 1) % LV;
      sp[0][(b ^ 1) * TB] = interp(rc[gs], rw[gs], 0);
      sp[1][(b ^ 1) * TB] = interp(rc[gs], rw[gs], 1);
      gather(p + 1 + LG, rv[vs], rc[gs], rw[gs]);
      load_vel(p + 1 + LG + LV, rv[vs]);
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        float o = wa * pm[r] + wb * pc[r] + wa * pn[r];
        o = (o >= 0.f) ? fabsf(o) : (o < 0.f ? -0.0f : o);
        const bool wr = p >= z0 + 1;
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, o), o_rsrc, wr ? (r ? oo1 : oo0) : OOB,
                                              wr ? (uint32_t)(p - 1) * uHW * 4u : 0u, 0);
        pm[r] = pc[r];
        pc[r] = pn[r];
      }
      __syncthreads();
    }
  }
}


// ---- S2: smooth + clamp alone in the fused kernel's geometry (16 rows x txe columns per block, two outputs and two staging
// slots per thread, PF planes of register look-ahead): does halving the barriers per voxel help the marching stencil? -------
template <int PF>
__global__ void __launch_bounds__(512) smooth2_kernel(const float* __restrict__ in, float* __restrict__ out, int D, int H,
                                                      int W, float k, int txe, int ntx, int nty, int nz, int zchunk) {
  constexpr int TYI = 16, TXM = 64, TB = (TYI + 2) * (TXM + 2);
  __shared__ float tile[2][TYI + 2][TXM + 2];
  __shared__ float dump[TB + 2];
  const int t = threadIdx.x, tx = t & 63, wv = t >> 6;
  const unsigned per_xcd = gridDim.x / 8;
  const unsigned lb = (blockIdx.x % 8) * per_xcd + blockIdx.x / 8;
  if (lb >= (unsigned)(ntx * nty * nz)) return;
  const int bx = lb % ntx, by = (lb / ntx) % nty, bz = lb / (ntx * nty);
  const int x0 = bx * txe, y0 = by * TYI, z0 = bz * zchunk, z1 = min(z0 + zchunk, D);
  const float inv = k > 0.f ? 1.f / (k + 2.f) : 1.f;
  const float wa = k > 0.f ? inv : 0.f, wb = k > 0.f ? k * inv : 1.f;
  const int cols = txe + 2, ne = (TYI + 2) * cols;
  const unsigned uHW = (unsigned)(H * W);
  uint32_t so[2];
  float* sp[2];
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    const int e = t + 512 * s, r = e / cols, c = e - r * cols;
    const int yy = y0 - 1 + r, xx = x0 - 1 + c;
    so[s] = (e < ne && yy >= 0 && yy < H && xx >= 0 && xx < W) ? (uint32_t)(yy * W + xx) * 4u : OOB;
    sp[s] = e < ne ? &tile[0][r][c] : dump + s;
  }
  const __amdgpu_buffer_rsrc_t i_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(in), 0, (uint32_t)((size_t)D * H * W * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t o_rsrc = __builtin_amdgcn_make_buffer_rsrc(out, 0, (uint32_t)((size_t)D * H * W * 4), 0x00020000);
  auto gload = [&](int pz, float* v) {
    const bool in_ = pz >= 0 && pz < D;
#pragma unroll
    for (int s = 0; s < 2; ++s)
      v[s] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(i_rsrc, in_ ? so[s] : OOB, in_ ? (uint32_t)pz * uHW * 4u : 0u, 0));
  };
  const int pf = z0 - 1;
  float rv[PF][2], v0[2];
  gload(pf, v0);
#pragma unroll
  for (int u = 0; u < PF; ++u) gload(pf + 1 + u, rv[u]);
  sp[0][0] = v0[0]; sp[1][0] = v0[1];
  __syncthreads();
  float pm[2] = {0.f, 0.f}, pc[2] = {0.f, 0.f};
  const int oy0 = y0 + wv, oy1 = y0 + wv + 8, ox = x0 + tx;
  const bool own0 = tx < txe && ox < W && oy0 < H, own1 = tx < txe && ox < W && oy1 < H;
  const uint32_t oo0 = own0 ? (uint32_t)(oy0 * W + ox) * 4u : OOB, oo1 = own1 ? (uint32_t)(oy1 * W + ox) * 4u : OOB;
  constexpr int LC = PF % 2 ? 2 * PF : PF;
  for (int pb = pf; pb <= z1; pb += LC) {
#pragma unroll
    for (int u = 0; u < LC; ++u) {
      const int p = pb + u;
      if (p > z1) break;
      const int b = u & 1;
      float pn[2];
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        const float* t0 = &tile[b][wv + 8 * r][tx];
        const float ra = wa * t0[0] + wb * t0[1] + wa * t0[2];
        const float rb = wa * t0[TXM + 2] + wb * t0[TXM + 3] + wa * t0[TXM + 4];
        const float rcc = wa * t0[2 * (TXM + 2)] + wb * t0[2 * (TXM + 2) + 1] + wa * t0[2 * (TXM + 2) + 2];
        pn[r] = wa * ra + wb * rb + wa * rcc;
      }
      const int rs = u % PF;
      sp[0][(b ^ 1) * TB] = rv[rs][0];
      sp[1][(b ^ 1) * TB] = rv[rs][1];
      gload(p + 1 + PF, rv[rs]);
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        float o = wa * pm[r] + wb * pc[r] + wa * pn[r];
        o = (o >= 0.f) ? fabsf(o) : (o < 0.f ? -0.0f : o);
        const bool wr = p >= z0 + 1;
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, o), o_rsrc, wr ? (r ? oo1 : oo0) : OOB,
                                              wr ? (uint32_t)(p - 1) * uHW * 4u : 0u, 0);
        pm[r] = pc[r];
        pc[r] = pn[r];
      }
      __syncthreads();
    }
  }
}

static float time_it(hipStream_t s, int reps, const std::function<void()>& f) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) f();
  CK(hipStreamSynchronize(s));
  CK(hipEventRecord(e0, s));
  for (int i = 0; i < reps; ++i) f();
  CK(hipEventRecord(e1, s));
  CK(hipEventSynchronize(e1));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  return 1e3f * ms / reps;
}

// cold: every rep is preceded by a 2 x 384 MB device copy (the step touches > 1 GB between two field kernels: nothing of
// theirs survives in the 256 MB Infinity Cache); HIP events around the kernel alone, summed
static float* g_pa = nullptr; static float* g_pb = nullptr;
static float time_cold(hipStream_t s, int reps, const std::function<void()>& f) {
  const size_t nb = (size_t)384 << 20;
  if (!g_pa) { CK(hipMalloc(&g_pa, nb)); CK(hipMalloc(&g_pb, nb)); CK(hipMemset(g_pa, 0, nb)); }
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  f();
  float tot = 0;
  for (int i = 0; i < reps; ++i) {
    CK(hipMemcpyAsync(g_pb, g_pa, nb, hipMemcpyDeviceToDevice, s));
    CK(hipEventRecord(e0, s));
    f();
    CK(hipEventRecord(e1, s));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    tot += ms;
  }
  return 1e3f * tot / reps;
}

static double max_diff(const float* a, const float* b, size_t n) {
  std::vector<float> ha(n), hb(n);
  CK(hipMemcpy(ha.data(), a, n * 4, hipMemcpyDeviceToHost));
  CK(hipMemcpy(hb.data(), b, n * 4, hipMemcpyDeviceToHost));
  double m = 0;
  for (size_t i = 0; i < n; ++i) m = fmax(m, fabs((double)ha[i] - hb[i]));
  return m;
}

int main(int argc, char** argv) {
  const int G = argc > 1 ? atoi(argv[1]) : 200;
  const int n = G * G * G;
  float *d, *vel, *o1, *o2, *o3, *m1, *m2;
  CK(hipMalloc(&d, (size_t)n * 4)); CK(hipMalloc(&vel, (size_t)n * 12));
  CK(hipMalloc(&o1, (size_t)n * 4)); CK(hipMalloc(&o2, (size_t)n * 4)); CK(hipMalloc(&o3, (size_t)n * 4));
  CK(hipMalloc(&m1, (size_t)n * 12)); CK(hipMalloc(&m2, (size_t)n * 12));       // moments: make the MALL state realistic
  hipStream_t s = 0;
  init_kernel<<<(n + 255) / 256, 256, 0, s>>>(d, vel, G);
  CK(hipDeviceSynchronize());
  const unsigned g4 = ((n + 1023) / 1024 + 7) / 8 * 8, g1 = ((n + 255) / 256 + 7) / 8 * 8;
  const int R = 50;
  auto report = [&](const char* name, float us, double mb, double diff) {
    printf("%-62s %8.1f us  %5.2f TB/s (%.0f MB)  maxdiff %.2e\n", name, us, mb / us, mb, diff);
  };
  const bool cold = argc > 2 && atoi(argv[2]) != 0;
  auto time_it = [&](hipStream_t st, int reps, const std::function<void()>& f) {
    return cold ? time_cold(st, 20, f) : ::time_it(st, reps, f);
  };
  printf("G = %d, %s\n", G, cold ? "COLD (polluting copy before every rep)" : "warm (back-to-back reps)");
  float us;
  us = time_it(s, R, [&] { advect_v4<true, 1><<<g4, 256, 0, s>>>(d, vel, o1, G, G, G); });
  report("A advect x4/lane, XCD remap, 8-byte gathers (product)", us, 160.0 * n / 8e6, 0);
  us = time_it(s, R, [&] { advect_v4<false, 1><<<g4, 256, 0, s>>>(d, vel, o2, G, G, G); });
  report("B  ... no remap", us, 160.0 * n / 8e6, max_diff(o1, o2, n));
  us = time_it(s, R, [&] { advect_v4<true, 0><<<g4, 256, 0, s>>>(d, vel, o2, G, G, G); });
  report("C  ... remap, NO gathers (streams only)", us, 128.0 * n / 8e6, -1);
  us = time_it(s, R, [&] { advect_v4<false, 0><<<g4, 256, 0, s>>>(d, vel, o2, G, G, G); });
  report("C' ... no remap, NO gathers", us, 128.0 * n / 8e6, -1);
  us = time_it(s, R, [&] { advect_v4<true, 2><<<g4, 256, 0, s>>>(d, vel, o2, G, G, G); });
  report("D  ... remap, 4-byte gathers x8", us, 160.0 * n / 8e6, max_diff(o1, o2, n));
  us = time_it(s, R, [&] { advect_v1<true><<<g1, 256, 0, s>>>(d, vel, o2, G, G, G); });
  report("E advect x1/lane, remap", us, 160.0 * n / 8e6, max_diff(o1, o2, n));
  us = time_it(s, R, [&] { advect_v1<false><<<g1, 256, 0, s>>>(d, vel, o2, G, G, G); });
  report("E' advect x1/lane, no remap", us, 160.0 * n / 8e6, max_diff(o1, o2, n));
  us = time_it(s, R, [&] { stream_kernel<<<(n + 1023) / 1024, 256, 0, s>>>(vel, o2, n); });
  report("S stream 12 B in + 4 B out per voxel", us, 128.0 * n / 8e6, -1);
  us = time_it(s, R, [&] { CK(hipMemcpyAsync(o2, o1, (size_t)n * 4, hipMemcpyDeviceToDevice, s)); });
  report("M hipMemcpy D2D 32 MB -> 32 MB", us, 64.0 * n / 8e6, -1);
  us = time_it(s, R, [&] { CK(hipMemcpyAsync(m1, vel, (size_t)n * 12, hipMemcpyDeviceToDevice, s)); });
  report("M hipMemcpy D2D 96 MB -> 96 MB", us, 192.0 * n / 8e6, -1);
  // in the step the forward follows the Adam kernel, which leaves 288 MB of vel / m / v behind: the same with that traffic between
  us = time_it(s, R, [&] {
    CK(hipMemcpyAsync(m1, m2, (size_t)n * 12, hipMemcpyDeviceToDevice, s));
    advect_v4<true, 1><<<g4, 256, 0, s>>>(d, vel, o1, G, G, G); });
  float us_c = time_it(s, R, [&] { CK(hipMemcpyAsync(m1, m2, (size_t)n * 12, hipMemcpyDeviceToDevice, s)); });
  printf("A after a 96 MB device copy: %.1f us (copy alone %.1f)\n", us - us_c, us_c);
  // fused advect + smooth
  smooth_ref_kernel<<<(n + 255) / 256, 256, 0, s>>>(o1, o3, G, G, G, 3.f);
  CK(hipDeviceSynchronize());
  for (int zc : {25, 13}) {
    const int txe = G <= 62 ? G : (G + ((G + 53) / 54) - 1) / ((G + 53) / 54);
    const int ntx = (G + txe - 1) / txe, nty = (G + 15) / 16, nz = (G + zc - 1) / zc;
    const unsigned grid = (ntx * nty * nz + 7) / 8 * 8;
    char name[128];
#define RUNS(PF_)                                                                                              \
    us = time_it(s, R, [&] { smooth2_kernel<PF_><<<grid, 512, 0, s>>>(o1, o2, G, G, G, 3.f, txe, ntx, nty, nz, zc); }); \
    snprintf(name, sizeof name, "S2 smooth alone, 16-row tiles, zc=%d PF=%d (%u blocks)", zc, PF_, grid);           \
    report(name, us, 64.0 * n / 8e6, max_diff(o3, o2, n));
    RUNS(1) RUNS(2) RUNS(4)
  }
  for (int zc : {25, 20, 13, 10}) {
    const int txe = G <= 62 ? G : (G + ((G + 53) / 54) - 1) / ((G + 53) / 54);
    const int ntx = (G + txe - 1) / txe, nty = (G + 15) / 16, nz = (G + zc - 1) / zc;
    const unsigned grid = (ntx * nty * nz + 7) / 8 * 8;
    char name[128];
#define RUN(LV_, LG_)                                                                                          \
    us = time_it(s, R, [&] { advect_smooth_kernel<LV_, LG_><<<grid, 512, 0, s>>>(d, vel, o2, G, G, G, 3.f, txe, ntx, nty, nz, zc); }); \
    snprintf(name, sizeof name, "F fused advect+smooth zc=%d txe=%d LV=%d LG=%d (%u blocks)", zc, txe, LV_, LG_, grid);   \
    report(name, us, 160.0 * n / 8e6, max_diff(o3, o2, n));
    RUN(1, 1) RUN(2, 1) RUN(2, 2) RUN(3, 2)
  }
  return 0;
}
