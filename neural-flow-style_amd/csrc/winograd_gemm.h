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
  const float* Uq = nullptr;            // B in MFMA fragment order [Z][N/32][K/8][64 lanes][4] (register-B kernel)
  const float* Uq16 = nullptr;          // ... for the 16x16x4 MFMA: [Z][N/16][K/16][64 lanes][4] (rb16 / rb16s kernels)
  const void* Ub16 = nullptr;           // rb16s: the same pack as three bf16 limb planes [Z][N/16][K/32][3][64 lanes][8 bf16]
  int symb = 0;                         // rb16: Uq16 is instead a plain SYMMETRIC [Z][K][N] matrix (the Gram gradient's D)
  unsigned long long* prof = nullptr;   // -DNFS_ABLATE builds: per-wave phase cycle sums (nfs_gemm_prof)
  int dbg = 0;               // NFS_GEMM_DBG timing ablations
  int ksplit = 1;            // rb16: the K range in `ksplit` parts, part p of a tile written to M + p * Z*T*N (summed by the
                             // output transform): fills the chip when T is a few dozen rows (one view per GPU)
};

// picks kernel family (by shape) and tile (measured per shape at first use) and launches; defined in winograd.hip
// returns the number of K parts the result was written in (1 unless winograd_ksplit() says otherwise AND the 16-row
// register-B kernel took the launch): the caller's output transform sums M + p * Z*T*N over the parts
int winograd_launch_batched_gemm(const WgGemmArgs& a, int Z, int cus, hipStream_t s);
// K parts by shape alone (never by a measurement: the parts change the order of a tile's sums): 2 when a launch has at
// most 64 rows and K >= 512 (conv4_x / conv5_1 at one view: 392 blocks of 16 chunks -> 784 of 8), else 1
int winograd_ksplit(int64_t T, int K);
// filters U [Z][K/32][N][32] -> the 16x16x4 fragment order [Z][N/16][K/16][64][4] (total = Z*K*N elements)
void winograd_pack_frag16(const float* up, float* uq, int K, int N, int64_t total, hipStream_t s);
// Uq16 [Z][N/16][K/16][64][4] floats -> the limb planes [Z][N/16][K/32][3][64][8 bf16] (1.5 x the floats); Z K N / 8 threads
void winograd_pack_limbs16(const float* uq16, float* ub16, int K, int N, int Z, hipStream_t s);

}  // namespace nfs
