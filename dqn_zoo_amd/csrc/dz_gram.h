// Gram matrices of a wide linear layer's input on the f64 matrix pipe (see
// dz_fc1_onfly.h for why: the layer's weight gradient G = X^T D is never stored, and
//     |G|^2 = sum_{b,b'} (x_b . x_b') (d_b . d_b')          (rank <= 32)
// gives its contribution to the global gradient norm from two 32 x 32 Grams).
// For the noisy layer's sigma matrix of head h, G_sig = (X.eps_in_h)^T (D_h.eps_out_h):
// the same with scaled rows.  The three input Grams  X X^T, (X.eps_in_adv)(..)^T,
// (X.eps_in_val)(..)^T  (32 x 32, depth 3136) depend only on the torso's output.
// Products of two floats are exact in double and the accumulation is double, so the norm
// does not depend on how correlated the batch rows are.
//
// One workgroup = one variant x one 16 x 16 output tile x one depth chunk of 448; its
// four waves take a quarter of the chunk each (28 dependent 64-cycle MFMAs instead of
// 112: the f64 pipe issues one v_mfma_f64_16x16x4_f64 per 64 cycles per SIMD) and are
// summed through LDS in wave order.  Lane (i, kq) feeds row i with four consecutive k per
// 16-deep step (A and B rows are read the same way, so the order of k inside a step
// cancels).  The partial tiles stay in the instruction's own register layout
// [variant][chunk][tile][lane][4]: the consumer (GramDSide, same tiling) only ever
// multiplies tiles element by element.  Host launch: the loss kernel's (32 workgroups on
// a 256-CU device, 8 us of dependent arithmetic: the rest of the chip is idle and the L2
// unloaded; inside the bandwidth-saturated fc1 forward stream the same blocks' dependent
// loads queued behind the weight stream and stretched that launch from 16 to 24 us).
#pragma once
#include "dz_qnet_ops.h"

namespace {
typedef double dz_d4 __attribute__((ext_vector_type(4)));
constexpr int kGramChunks = 7;   // 3136 = 7 x 448
constexpr int kGramXBlocks = 3 * 4 * kGramChunks;
struct GramX {
  const float* x = nullptr;      // [M][K] (the online s_tm1 apply's rows); nullptr: off
  int M = 0, K = 0;
  const float* eps_in[2] = {nullptr, nullptr};
  double* part = nullptr;        // [3][kGramChunks][4][64][4]
};
__device__ __forceinline__ double dz_wave_sum_f64(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
// `red`: 3 x 64 dz_d4 of LDS.
__device__ __forceinline__ void dz_gram_x_block(const GramX& q, unsigned blk, dz_d4* red) {
  if (blk >= (unsigned)kGramXBlocks) return;
  const int tile = (int)blk & 3, vc = (int)blk >> 2;
  const int v = vc / kGramChunks, c = vc % kGramChunks;
  const int kc = q.K / kGramChunks, kw = kc / 4;          // per wave: 112 = 7 x 16
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, i = lane & 15, kq = lane >> 4;
  const int ra = (tile >> 1) * 16 + i, rb = (tile & 1) * 16 + i;
  const float ma = ra < q.M ? 1.f : 0.f, mb = rb < q.M ? 1.f : 0.f;
  const int k0 = c * kc + wave * kw + 4 * kq;
  const float* xa = q.x + (unsigned)(min(ra, q.M - 1) * q.K + k0);
  const float* xb = q.x + (unsigned)(min(rb, q.M - 1) * q.K + k0);
  const float* e = (v == 2 ? q.eps_in[1] : q.eps_in[0]) + k0;
  constexpr int NS = 7;
  float4 a[NS], b[NS], ee[NS];
#pragma unroll
  for (int s = 0; s < NS; ++s) {                          // everything in flight at once
    a[s] = *(const float4*)(xa + 16 * s); b[s] = *(const float4*)(xb + 16 * s);
    ee[s] = *(const float4*)(e + 16 * s);
  }
  dz_d4 acc[2] = {{0.0, 0.0, 0.0, 0.0}, {0.0, 0.0, 0.0, 0.0}};   // two chains
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    if (v == 0) ee[s] = dz_f4(1.f, 1.f, 1.f, 1.f);
    const float A[4] = {a[s].x, a[s].y, a[s].z, a[s].w}, Bv[4] = {b[s].x, b[s].y, b[s].z, b[s].w};
    const float E[4] = {ee[s].x, ee[s].y, ee[s].z, ee[s].w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const double da = (double)(A[j] * ma) * (double)E[j];
      const double db = (double)(Bv[j] * mb) * (double)E[j];
      acc[j & 1] = __builtin_amdgcn_mfma_f64_16x16x4f64(da, db, acc[j & 1], 0, 0, 0);
    }
  }
  dz_d4 t = acc[0] + acc[1];
  if (wave) red[(wave - 1) * 64 + lane] = t;
  __syncthreads();
  if (wave == 0) {
    t += red[lane]; t += red[64 + lane]; t += red[128 + lane];
    ((dz_d4*)q.part)[(vc * 4 + tile) * 64 + lane] = t;
  }
}
}  // namespace
