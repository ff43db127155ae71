// A6, the batched Winograd GEMMs of a deep layer when the launch has only a FEW ROWS (one or two views per GPU: conv4_x at
// 25 x 25 is 25 tiles, conv5_1 at 12 x 12 is 9; BASELINE configs[1] at 100^3: 25 / 9 / 4) -- "K7g".
//
// With a few dozen rows the product M_z = V_z U_z is no GEMM any more: the register-B kernel streams the transformed
// filters U = G g G^T -- 49 (36) floats per (ci, co) where the filter itself has 9: 51 MB for a 512 x 512 layer -- once
// per launch for 0.8 GFLOP, 18-23 us of a 28-us layer pass at ~2.5 TB/s.  Here the filter transform moves INTO the
// kernel: a wave keeps one ROW r of the component grid (z = R r + q, q = 0 .. R-1; R = 7 for F(5x5), 6 for F(4x4)),
// loads the 9 taps of its (k, n) operand elements from the direct-form pack (the [K/32][9][N][32] block at the head of
// every packed filter set: 9.4 MB instead of 51), applies row r of G down the filter columns (3 results) and all of G
// across them (R results: w5_g / wg4_g, the very functions the pack kernels call) and feeds the R values to R MFMAs as
// their B operands.  5.4 x (4 x) fewer filter bytes from HBM; the launch becomes MFMA-bound at its 16-row padding.
//
//   block (4 waves) = component row r  x  4 / MTB column tiles of 16  x  MTB row tiles of 16  x  K part
//   wave            = one row tile, one column tile: R accumulators of 16 x 16
//   A (V_z, z = R r + q): the block's R x 16 MTB rows x 16-deep k group staged through LDS (double buffer, one barrier per
//       group); rows beyond T come back from the buffer load as zeros
//   B: 9 buffer loads of float4 per lane and k group (the lane's four k of MFMA steps 0..3, as in the rb16 kernel),
//       prefetched one group ahead; G twice in registers (~30 VALU per element against R MFMAs of 32 cycles)
//   K parts: chosen by shape alone so that component rows x column tiles x row tiles x parts ~ the chip's 1024 SIMDs;
//       part p of a component goes to M + p Z T N and the output transforms sum the parts (NSPLIT 1 / 2 / 4)
//
// Same products as the three-kernel path up to float32 rounding (the filter transform rounds as the pack kernel does
// except for the single G row of the first stage, written per row here; the k sums run in a different grouping).
#include <atomic>
#include <mutex>

#include "common.h"
#include "winograd_gemm.h"
#include "winograd_math.h"

namespace nfs {

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float4 ft_ld4(__amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t soff) {
  return __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}

// row rp of G (the first stage of U = G g G^T runs down the filter columns with ONE row of G per block: three scalar
// coefficients; the second stage is the pack kernels' own w5_g / wg4_g).  The pack kernels write rows 1 and 2 as a
// factor times a sum, here they are three products: the same value up to one rounding.
__device__ __constant__ float FT_G5[7][3] = {{-0.5f, 0.f, 0.f},
                                             {-1.f / 3.f, -1.f / 3.f, -1.f / 3.f},
                                             {1.f / 9.f, -1.f / 9.f, 1.f / 9.f},
                                             {1.f / 36.f, 1.f / 18.f, 1.f / 9.f},
                                             {-1.f / 60.f, 1.f / 30.f, -1.f / 15.f},
                                             {32.f / 45.f, 16.f / 45.f, 8.f / 45.f},
                                             {0.f, 0.f, 1.f}};
__device__ __constant__ float FT_G4[6][3] = {{0.25f, 0.f, 0.f},
                                             {-1.f / 6.f, -1.f / 6.f, -1.f / 6.f},
                                             {-1.f / 6.f, 1.f / 6.f, -1.f / 6.f},
                                             {1.f / 24.f, 1.f / 12.f, 1.f / 6.f},
                                             {1.f / 24.f, -1.f / 12.f, 1.f / 6.f},
                                             {0.f, 0.f, 1.f}};

struct FtArgs {
  const float* V;     // [Z][T][K]
  const float* wp;    // direct-form pack [K/32][9][N][32]
  float* M;           // [ksplit][Z][T][N]
  int T, K, N, ksplit;
};

constexpr int FT_KG = 16;        // k per group (one MFMA quadruple)
constexpr int FT_LS = 20;        // LDS row stride in floats (16 + 4: the 16 fragment rows of a lane group hit 64 distinct banks)

template <int R, int MTB>
__global__ void __launch_bounds__(256) winograd_gemm_ft_kernel(FtArgs a) {
  constexpr int NTB = 4 / MTB;                 // column tiles per block
  constexpr int ROWS = 16 * MTB;               // staged rows per component
  constexpr int NF4 = R * ROWS * 4;            // float4 per staged group
  constexpr int AJ = (NF4 + 255) / 256;
  extern __shared__ __attribute__((aligned(16))) float ft_smem[];      // [2][R * ROWS * FT_LS]
  const int t = threadIdx.x, lane = t & 63;
  const int wid = __builtin_amdgcn_readfirstlane(t >> 6);
  // block -> (group = (K part, column block), component row): the R blocks of a group read the same filter taps, so
  // they sit on ONE XCD (ids congruent mod 8) and share them in its L2
  const int ncb = a.N / (16 * NTB), groups = ncb * a.ksplit;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int grp = (slot / R) * 8 + xcd, rp = slot % R;
  if (grp >= groups) return;
  const int split = grp / ncb, cb = grp - split * ncb;
  const int mt = wid % MTB, ntw = wid / MTB;
  const int n0 = (cb * NTB + ntw) * 16;
  const int kpart = a.K / a.ksplit, k_begin = split * kpart, ngroups = kpart / FT_KG;

  const __amdgpu_buffer_rsrc_t a_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(a.V + (int64_t)rp * R * a.T * a.K), 0, (uint32_t)((int64_t)R * a.T * a.K * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t b_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(a.wp), 0, (uint32_t)((int64_t)9 * a.K * a.N * 4), 0x00020000);

  // A staging: float4 #f = t + 256 j of the group: component q = f / (4 ROWS), row (f / 4) % ROWS, quarter f % 4
  uint32_t ao[AJ];
  int as_off[AJ];
#pragma unroll
  for (int j = 0; j < AJ; ++j) {
    const int f = t + 256 * j, q = f / (4 * ROWS), row = (f >> 2) % ROWS, qu = f & 3;
    const bool ok = f < NF4 && row < a.T;
    ao[j] = ok ? (uint32_t)((((int64_t)q * a.T + row) * a.K + 4 * qu) * 4) : 0x80000000u;     // (out of range: zeros)
    as_off[j] = f < NF4 ? (q * ROWS + row) * FT_LS + 4 * qu : -1;
  }
  // B: tap tp of the lane's (k = k16 + 4 (lane >> 4) + s, n = n0 + (lane & 15)):
  //    ((k16 / 32 * 9 + tp) * N + n) * 32 + k16 % 32 + 4 (lane >> 4)      (floats)
  const uint32_t bo = (uint32_t)(((n0 + (lane & 15)) * 32 + 4 * (lane >> 4)) * 4);
  const uint32_t tap_stride = (uint32_t)a.N * 32u * 4u;
#define FT_B_SOFF(k16) ((uint32_t)((k16) >> 5) * 9u * tap_stride + (uint32_t)((k16) & 16) * 4u)

  float4 av[AJ], gn[9];
#pragma unroll
  for (int j = 0; j < AJ; ++j) av[j] = ft_ld4(a_rsrc, ao[j], (uint32_t)k_begin * 4u);
  {
    const uint32_t so = FT_B_SOFF(k_begin);
#pragma unroll
    for (int tp = 0; tp < 9; ++tp) gn[tp] = ft_ld4(b_rsrc, bo, so + tp * tap_stride);
  }

  const float gr0 = R == 7 ? FT_G5[rp][0] : FT_G4[rp][0], gr1 = R == 7 ? FT_G5[rp][1] : FT_G4[rp][1],
              gr2 = R == 7 ? FT_G5[rp][2] : FT_G4[rp][2];
  f32x4 acc[R];
#pragma unroll
  for (int q = 0; q < R; ++q) acc[q] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int afrag = (16 * mt + (lane & 15)) * FT_LS + 4 * (lane >> 4);

#pragma unroll 1
  for (int g = 0; g < ngroups; ++g) {
    float* Ac = ft_smem + (g & 1) * (R * ROWS * FT_LS);
#pragma unroll
    for (int j = 0; j < AJ; ++j)
      if (as_off[j] >= 0) *reinterpret_cast<float4*>(Ac + as_off[j]) = av[j];
    float4 gc[9];
#pragma unroll
    for (int tp = 0; tp < 9; ++tp) gc[tp] = gn[tp];
    __syncthreads();                         // buffer (g & 1) visible; the other one was last read in iteration g - 1
    const int kn = k_begin + FT_KG * (g + 1 < ngroups ? g + 1 : g);      // (the last iteration re-fetches its own group)
#pragma unroll
    for (int j = 0; j < AJ; ++j) av[j] = ft_ld4(a_rsrc, ao[j], (uint32_t)kn * 4u);
    {
      const uint32_t so = FT_B_SOFF(kn);
#pragma unroll
      for (int tp = 0; tp < 9; ++tp) gn[tp] = ft_ld4(b_rsrc, bo, so + tp * tap_stride);
    }
    // U[R rp + q][k][n] for the lane's four k: G row rp down the columns, G across
    float u[4][R];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      float tc[3];
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const float g0 = s == 0 ? gc[c].x : s == 1 ? gc[c].y : s == 2 ? gc[c].z : gc[c].w;
        const float g1 = s == 0 ? gc[3 + c].x : s == 1 ? gc[3 + c].y : s == 2 ? gc[3 + c].z : gc[3 + c].w;
        const float g2 = s == 0 ? gc[6 + c].x : s == 1 ? gc[6 + c].y : s == 2 ? gc[6 + c].z : gc[6 + c].w;
        tc[c] = gr0 * g0 + gr1 * g1 + gr2 * g2;
      }
      if (R == 7) w5_g(tc[0], tc[1], tc[2], u[s]);
      else wg4_g(tc[0], tc[1], tc[2], u[s]);
    }
    float4 af[R];
#pragma unroll
    for (int q = 0; q < R; ++q) af[q] = *reinterpret_cast<const float4*>(Ac + q * ROWS * FT_LS + afrag);
    // step-major: the R accumulators take turns, so that no MFMA waits for the one before it
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int q = 0; q < R; ++q) {
        const float av_ = s == 0 ? af[q].x : s == 1 ? af[q].y : s == 2 ? af[q].z : af[q].w;
        asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[q]) : "v"(av_), "v"(u[s][q]));
      }
  }
  // (the MFMAs are opaque to the compiler's hazard recogniser: let the last ones retire before the stores read them)
  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");

  // epilogue: C layout of the 16x16x4 MFMA -- lane: column lane & 15, rows 4 (lane >> 4) + r
  const int Z = R * R;
  float* Mc = a.M + (((int64_t)split * Z + (int64_t)rp * R) * a.T) * a.N + n0 + (lane & 15);
#pragma unroll
  for (int q = 0; q < R; ++q)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = 16 * mt + 4 * (lane >> 4) + r;
      if (row < a.T) Mc[((int64_t)q * a.T + row) * a.N] = acc[q][r];
    }
}

// ---- host side ------------------------------------------------------------------------------------------------
// K parts of a few-row launch, by shape alone: 0 = no few-row form for this shape (more than 64 rows, operands beyond
// 32-bit offsets), else 1 / 2 / 4 parts of at least 32 k each so that the waves about fill the chip's 1024 SIMDs
int winograd_fewrow_parts_of_shape(int64_t T, int K, int N, int Z) {
  static const int tmax = [] { const char* e = getenv("NFS_WG_FEWROW_T"); return e ? atoi(e) : 64; }();
  if ((Z != 49 && Z != 36) || T < 1 || T > tmax || T > 64 || K % 32 || N % 64 || K < 64) return 0;
  if ((int64_t)Z * T * K * 4 >= ((int64_t)1 << 31) || (int64_t)9 * K * N * 4 >= ((int64_t)1 << 31)) return 0;
  const int R = Z == 49 ? 7 : 6, mtb = T <= 16 ? 1 : T <= 32 ? 2 : 4;
  const int waves = R * (N / 16) * mtb;
  static const int forced = [] { const char* e = getenv("NFS_WG_FEWROW_PARTS"); return e ? atoi(e) : 0; }();
  int s = 1;
  while (s < 4 && waves * s * 2 <= 1024 + 128 && K / (2 * s) >= 32 && (K / (2 * s)) % FT_KG == 0) s *= 2;
  if (forced == 1 || forced == 2 || forced == 4) s = forced;
  while (s > 1 && ((K / s) % FT_KG || K / s < FT_KG)) s >>= 1;
  return s;
}

// on / off: NFS_WG_FEWROW=0 at load time, nfs_conv3x3_fewrow() at run time (A/B tests); the workspace a convolution asks
// for (winograd_mparts) covers the few-row parts either way
static std::atomic<int> g_fewrow{-1};
static int fewrow_mode() {
  int m = g_fewrow.load();
  if (m < 0) {
    const char* e = getenv("NFS_WG_FEWROW");
    m = (e && atoi(e) == 0) ? 0 : 1;
    g_fewrow.store(m);
  }
  return m;
}
int winograd_fewrow_parts(int64_t T, int K, int N, int Z) {
  return fewrow_mode() ? winograd_fewrow_parts_of_shape(T, K, N, Z) : 0;
}

template <int R, int MTB>
static void launch_ft(const FtArgs& a, hipStream_t s) {
  const int ncb = a.N / (16 * (4 / MTB)), groups = ncb * a.ksplit;
  const int grid = (groups + 7) / 8 * R * 8;
  constexpr size_t lds = (size_t)2 * R * 16 * MTB * FT_LS * sizeof(float);
  static std::once_flag attr_once;
  if (lds > 65536) std::call_once(attr_once, [&] {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(winograd_gemm_ft_kernel<R, MTB>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); });
  hipLaunchKernelGGL((winograd_gemm_ft_kernel<R, MTB>), dim3(grid), dim3(256), lds, s, a);
}

// M [parts][Z][T][N] = V [Z][T][K] x (G g G^T); wp = the direct-form pack of g.  Returns the number of K parts written
// (the caller's output transform sums them), 0 when the launch is not taken.
int winograd_fewrow_gemm(const float* V, const float* wp, float* M, int64_t T, int K, int N, int Z, hipStream_t s) {
  const int parts = winograd_fewrow_parts(T, K, N, Z);
  if (parts == 0) return 0;
  FtArgs a{V, wp, M, (int)T, K, N, parts};
  const int mtb = T <= 16 ? 1 : T <= 32 ? 2 : 4;
  if (Z == 49) {
    if (mtb == 1) launch_ft<7, 1>(a, s); else if (mtb == 2) launch_ft<7, 2>(a, s); else launch_ft<7, 4>(a, s);
  } else {
    if (mtb == 1) launch_ft<6, 1>(a, s); else if (mtb == 2) launch_ft<6, 2>(a, s); else launch_ft<6, 4>(a, s);
  }
  return parts;
}

}  // namespace nfs

extern "C" int nfs_conv3x3_fewrow(int mode) {
  const int prev = nfs::fewrow_mode();
  if (mode == 0 || mode == 1) nfs::g_fewrow.store(mode);
  return prev;
}
