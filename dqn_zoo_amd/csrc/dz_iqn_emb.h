// The IQN tau embedding's forward pass (round 5):
//   head_in[row][c] = relu(sum_l cos[row][l] Wemb_g[l][c] + b_g[c]) * feat[feat_row(row)][c]
// (networks.py:277-285) for 6 144 rows x 3 136 columns x 64 deep at the reference sizes (64 / 64 / 64 taus): 2.5 GFLOP
// for 77 MB written (64 MB at round 5's 5 120 rows, where the numbers below were taken).  As a tile GEMM of the general skeleton (IqnLinOp, IQN_EPI_MIX) that is
// 3 920 workgroups of two stages behind masked loaders and two barriers each: 43 us.  Here a
// workgroup owns a 64-row tile -- its cosine rows go to LDS once, in the skeleton's KC fragment
// layout, and from there into each wave's registers for good -- and walks over a segment of
// 64-column tiles with the weight tiles double-buffered in LDS (the next tile's weights, and
// this tile's bias / feature factors, are requested before the tile's 32 MFMAs; one barrier per
// tile; no masks: whole tiles only).  Measured, whole step on one box: 454.5 (GEMM form) ->
// 449 us with 1-2 tiles per workgroup; 3, 6-7, 12 tiles per workgroup 453.9 / 451.5-453.2 /
// 456.8 -- walking further along the row buys nothing (the launch is bound by its 64 MB of
// 256-byte row segments, not by per-workgroup latency), the leaner tile body is the gain.
// The arithmetic order per output element is the skeleton's (chunks 0..3, k-slot permutation of
// dz_gemm.h; bias, ReLU, feature factor in the store).  Every other shape keeps the GEMM form
// (rows of every group % 64 == 0, latent == 64, N % 64 == 0 are required here).
#pragma once

#include "dz_iqn_ops.h"

namespace {

struct IqnEmbParams {
  const float* cos;          // [rows][64]
  int G;
  int row0[DZ_MAX_GROUPS];   // first row of the group
  int tiles[DZ_MAX_GROUPS];  // 64-row tiles of the group
  const float* params[DZ_MAX_GROUPS];
  long w_off, b_off;
  int ldw;                   // = N (3 136), multiple of 64
  int N;
  const float* feat;         // [*][N]
  int feat_row0[DZ_MAX_GROUPS];
  int samples[DZ_MAX_GROUPS];
  float* out;                // [rows][N]
  int segs;                  // column segments per row tile
};

constexpr int kEmbAChunk = 64 * 20 + 16;            // DzLdsTile<64, 4, KC>::CHUNK
constexpr int kEmbBChunk = 16 * 64;                 // DzLdsTile<64, 4, RC>::CHUNK
constexpr int kEmbLdsFloats = 4 * kEmbAChunk + 2 * 4 * kEmbBChunk;   // 53.5 KB

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3)))
void iqn_emb_fwd_kernel(IqnEmbParams p) {
  __shared__ __attribute__((aligned(16))) float lds[kEmbLdsFloats];
  float* As = lds;
  float* Bs = lds + 4 * kEmbAChunk;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1, half = lane >> 5, l31 = lane & 31;
  // block -> (group, row tile, column segment); static-index selects only (second loader rule)
  int rt = blockIdx.x / p.segs;
  const int seg = blockIdx.x % p.segs;
  int g = 0;
  if (rt >= p.tiles[0]) { rt -= p.tiles[0]; g = 1; if (rt >= p.tiles[1]) { rt -= p.tiles[1]; g = 2; } }
  const float* prm = dz_pick3(p.params, g);
  const int row0 = dz_pick3(p.row0, g) + rt * 64;      // first row of the tile (global row index)
  const int grow0 = rt * 64;                            // ... within its group
  const int samples = dz_pick3(p.samples, g), frow0 = dz_pick3(p.feat_row0, g);
  const int ntile = p.N / 64;
  const int t_begin = (int)((long)seg * ntile / p.segs), t_end = (int)((long)(seg + 1) * ntile / p.segs);
  if (t_begin >= t_end) return;
  const float* W = prm + p.w_off;
  const float* bias = prm + p.b_off;

  // weight tile t -> registers (4 float4 per thread), registers -> LDS buffer `buf`
  float4 rb[4];
  auto load_b = [&](int t) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int idx = tid + j * 256, kidx = idx >> 4, rq = idx & 15;
      rb[j] = dz_ld4(W + (long)kidx * p.ldw + t * 64 + 4 * rq);
    }
  };
  auto store_b = [&](int buf) {
    float* dst = Bs + buf * 4 * kEmbBChunk;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int idx = tid + j * 256, kidx = idx >> 4, rq = idx & 15;
      *(float4*)(dst + (kidx >> 4) * kEmbBChunk + (kidx & 15) * 64 + 4 * rq) = rb[j];
    }
  };
  load_b(t_begin);
  {  // the tile's cosine rows, once
    float4 ra[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int idx = tid + j * 256, row = idx >> 4, rem = idx & 15;
      ra[j] = dz_ld4(p.cos + (long)(row0 + row) * 64 + (rem >> 2) * 16 + 4 * (rem & 3));
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int idx = tid + j * 256, row = idx >> 4, rem = idx & 15;
      *(float4*)(As + (rem >> 2) * kEmbAChunk + row * 20 + 4 * (rem & 3)) = ra[j];
    }
  }
  store_b(0);
  __syncthreads();
  // this wave's A fragments never change: 4 chunks x 8 k-slots in registers
  float fa[4][8];
#pragma unroll
  for (int ch = 0; ch < 4; ++ch) {
    const float* src = As + ch * kEmbAChunk + (wm * 32 + l31) * 20 + half * 8;
    const float4 v0 = *(const float4*)src, v1 = *(const float4*)(src + 4);
    fa[ch][0] = v0.x; fa[ch][1] = v0.y; fa[ch][2] = v0.z; fa[ch][3] = v0.w;
    fa[ch][4] = v1.x; fa[ch][5] = v1.y; fa[ch][6] = v1.z; fa[ch][7] = v1.w;
  }
  // feature rows of this lane's 16 accumulator rows (they do not depend on the column tile)
  int frow[16];
#pragma unroll
  for (int r = 0; r < 16; ++r)
    frow[r] = frow0 + (grow0 + wm * 32 + dz_acc_row(r, lane)) / samples;

  for (int t = t_begin; t < t_end; ++t) {
    const int buf = (t - t_begin) & 1;
    if (t + 1 < t_end) load_b(t + 1);                 // in flight under this tile's MFMAs
    const int col = t * 64 + wn * 32 + l31;
    const float b = bias[col];
    float fm[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) fm[r] = p.feat[(long)frow[r] * p.N + col];
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    const float* Bt = Bs + buf * 4 * kEmbBChunk;
#pragma unroll
    for (int ch = 0; ch < 4; ++ch) {
      const float* src = Bt + ch * kEmbBChunk + (half * 8) * 64 + wn * 32 + l31;
      float fb[8];
#pragma unroll
      for (int s = 0; s < 8; ++s) fb[s] = src[s * 64];
#pragma unroll
      for (int s = 0; s < 8; ++s)
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[ch][s], fb[s], acc, 0, 0, 0);
    }
    if (t + 1 < t_end) store_b(buf ^ 1);              // (read last in iteration t-1: behind a barrier)
    float* o = p.out + (long)(row0 + wm * 32) * p.N + col;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float v = acc[r] + b;
      v = v > 0.f ? v : 0.f;
      o[(long)dz_acc_row(r, lane) * p.N] = v * fm[r];
    }
    __syncthreads();
  }
}

}  // namespace
