// F(4x4, 3x3) transform arithmetic shared by the three-kernel Winograd path (winograd.hip) and the single-kernel one
// (winograd_fused.hip): both must produce the same V and Y from the same operands.
#pragma once
#include "common.h"

namespace nfs {

// B^T (6x6) applied to a column / row of float2 channel pairs
__device__ __forceinline__ float2 wg_lin(float a, float2 x, float b, float2 y) {
#pragma clang fp contract(off)
  return make_float2(a * x.x + b * y.x, a * x.y + b * y.y);
}
// (no FMA contraction inside the transforms: which products get fused depends on the code around an inlined copy, and
// the kernels that share them -- one thread per tile or six waves per tile, float masks or the bit cache -- must round alike)
__device__ __forceinline__ void wg4_bt(const float2* d, float2* o) {
#pragma clang fp contract(off)
  // [4,0,-5,0,1,0] [0,-4,-4,1,1,0] [0,4,-4,-1,1,0] [0,-2,-1,2,1,0] [0,2,-1,-2,1,0] [0,4,0,-5,0,1]
  const float2 p = wg_lin(-4.f, d[2], 1.f, d[4]);     // d4 - 4 d2
  const float2 q = wg_lin(-4.f, d[1], 1.f, d[3]);     // d3 - 4 d1
  const float2 e = wg_lin(-1.f, d[2], 1.f, d[4]);     // d4 - d2
  const float2 f = wg_lin(-2.f, d[1], 2.f, d[3]);     // 2 (d3 - d1)
  o[0] = make_float2(4.f * d[0].x - 5.f * d[2].x + d[4].x, 4.f * d[0].y - 5.f * d[2].y + d[4].y);
  o[1] = make_float2(p.x + q.x, p.y + q.y);
  o[2] = make_float2(p.x - q.x, p.y - q.y);
  o[3] = make_float2(e.x + f.x, e.y + f.y);
  o[4] = make_float2(e.x - f.x, e.y - f.y);
  o[5] = make_float2(4.f * d[1].x - 5.f * d[3].x + d[5].x, 4.f * d[1].y - 5.f * d[3].y + d[5].y);
}

// A^T (4x6) = [1,1,1,1,1,0] [0,1,-1,2,-2,0] [0,1,1,4,4,0] [0,1,-1,8,-8,1]
__device__ __forceinline__ void wg4_at(const float2* m, float2* o) {
#pragma clang fp contract(off)
  const float2 s12 = make_float2(m[1].x + m[2].x, m[1].y + m[2].y), d12 = make_float2(m[1].x - m[2].x, m[1].y - m[2].y);
  const float2 s34 = make_float2(m[3].x + m[4].x, m[3].y + m[4].y), d34 = make_float2(m[3].x - m[4].x, m[3].y - m[4].y);
  o[0] = make_float2(m[0].x + s12.x + s34.x, m[0].y + s12.y + s34.y);
  o[1] = make_float2(d12.x + 2.f * d34.x, d12.y + 2.f * d34.y);
  o[2] = make_float2(s12.x + 4.f * s34.x, s12.y + 4.f * s34.y);
  o[3] = make_float2(d12.x + 8.f * d34.x + m[5].x, d12.y + 8.f * d34.y + m[5].y);
}

// the same B^T for one channel (winograd_fused.hip transforms one (tile, channel) per lane)
__device__ __forceinline__ void wg4_bt1(const float* d, float* o) {
  const float p = d[4] - 4.f * d[2];
  const float q = d[3] - 4.f * d[1];
  const float e = d[4] - d[2];
  const float f = 2.f * d[3] - 2.f * d[1];
  o[0] = 4.f * d[0] - 5.f * d[2] + d[4];
  o[1] = p + q;
  o[2] = p - q;
  o[3] = e + f;
  o[4] = e - f;
  o[5] = 4.f * d[1] - 5.f * d[3] + d[5];
}

}  // namespace nfs
