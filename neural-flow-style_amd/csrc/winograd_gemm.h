// The batched GEMM shared by the Winograd convolutions (winograd.hip, winograd5.hip) and the Gram gradient (gram.hip).
#pragma once
#include "common.h"

namespace nfs {

// Batched C_z = alpha_z * A_z B_z (optionally masked): the Winograd GEMMs (z = transform component) and the
// Gram gradient dF_b = 2 s_b F_b D_b (z = image) share this kernel.
struct WgGemmArgs {
  const float* V;    // A: [Z][T][K] row-major
  const float* U;    // B: element (k, n) of batch z at z*b_batch + (k/32)*b_chunk + n*b_row + k%32
  float* M;          // C: [Z][T][N] row-major
  int64_t T;
  int K, N;
  int64_t b_batch, b_chunk;
  int b_row;
  float alpha;               // C scale (1 for Winograd)
  const float* alpha_dev;    // optional per-batch scale (device)
  const float* mask;         // optional [Z][T][N]: C = mask > 0 ? C : 0
  int mt, nt, Z;             // tile counts (filled by launch_batched_gemm)
  const unsigned short* Ub = nullptr;   // B as three bf16 limb planes [Z][K/32][N][3][32] (split-limb kernel)
  const float* Uq = nullptr;            // B in MFMA fragment order [Z][N/32][K/8][64 lanes][4] (register-B kernel)
  const float* Uq16 = nullptr;          // ... for the 16x16x4 MFMA: [Z][N/16][K/16][64 lanes][4] (rb16 kernel)
  int symb = 0;                         // rb16: Uq16 is instead a plain SYMMETRIC [Z][K][N] matrix (the Gram gradient's D)
  unsigned long long* prof = nullptr;   // -DNFS_ABLATE builds: per-wave phase cycle sums (nfs_gemm_prof)
  int dbg = 0;               // NFS_GEMM_DBG timing ablations
};

// picks kernel family (by shape) and tile (measured per shape at first use) and launches; defined in winograd.hip
void winograd_launch_batched_gemm(const WgGemmArgs& a, int Z, int cus, hipStream_t s);
// filters U [Z][K/32][N][32] -> the 16x16x4 fragment order [Z][N/16][K/16][64][4] (total = Z*K*N elements)
void winograd_pack_frag16(const float* up, float* uq, int K, int N, int64_t total, hipStream_t s);

}  // namespace nfs
