// conv1 forward with the weights resident in registers (experiment, round 2).
//
// conv1 has ONE column tile (32 output channels) and a 32 KB filter bank, yet the tile GEMM
// re-stages that bank through LDS in each of its 600 workgroups.  Here a workgroup keeps
// its quarter of the filter bank per wave in 32 VGPRs per lane for its whole life (wave
// wk owns kernel rows 2wk, 2wk+1: reduction chunks 4wk..4wk+3), walks 4-5 row tiles of 32
// output pixels, and only the uint8 input patch of a tile (a contiguous pixel range,
// copied flat into a double-buffered LDS slot while the previous tile computes) moves
// per tile.  The four K-quarter accumulators of a tile are exchanged through LDS and each
// wave finishes 4 of the 16 accumulator registers (+ bias, ReLU, store).
// grid = (WPG workgroups per group, G groups); tile t of group z runs on workgroup t % WPG.
#pragma once

#include "dz_qnet_ops.h"

namespace conv1p {
constexpr int H = 84, W = 84, KS = 8, S = 4, OH = 20, OW = 20, CO = 32, BM = 32;
constexpr int base_of(int m) {
  return (m / (OH * OW)) * H * W + ((m % (OH * OW)) / OW) * S * W + ((m % (OH * OW)) % OW) * S;
}
constexpr int max_span() {
  int mx = 0;
  for (int m0 = 0; m0 < OH * OW; ++m0) {
    const int sp = base_of(m0 + BM - 1) - base_of(m0) + (KS - 1) * W + KS;
    mx = sp > mx ? sp : mx;
  }
  return mx;
}
constexpr int MAXPX = max_span();
constexpr int PATCH = ((MAXPX + 3) / 4) * 4;      // dwords per patch slot
constexpr int NCP = (PATCH / 4 + 255) / 256;      // uint4 copy slots per thread
constexpr int RED = 4 * 16 * 64;                  // one exchange buffer
constexpr int SMEM = 2 * PATCH + 2 * RED;         // floats
}  // namespace conv1p

template <int WPG>
__global__ __launch_bounds__(256) void conv1_persist_kernel(ConvFwdParams p) {
  using namespace conv1p;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* patch = smem;                 // [2][PATCH] dwords
  float* red = smem + 2 * PATCH;       // [2][4][16][64]
  const int z = blockIdx.y, wg = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wk = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5, l31 = lane & 31;
  const uint8_t* __restrict__ in = (const uint8_t*)dz_pick3(p.in, z);
  const float* __restrict__ w = dz_pick3(p.w, z);
  const float* __restrict__ bias = dz_pick3(p.bias, z);
  const long img0 = dz_pick3(p.in_img_base, z);
  const int rows = p.B * OH * OW;
  const int tiles = (rows + BM - 1) / BM;

  // this wave's quarter of the filter bank: chunk c = 4wk + j, step s: W[16c + 8 half + s][l31]
  float wreg[4][8];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int s = 0; s < 8; ++s) wreg[j][s] = w[(long)(16 * (4 * wk + j) + 8 * half + s) * CO + l31];
  const float bcol = bias[l31];

  auto patch_src = [&](int t, int& n16) {
    const int m0 = t * BM, m_last = min(m0 + BM, rows) - 1;
    const int lo = base_of(m0);
    n16 = (base_of(m_last) + (KS - 1) * W + KS - lo + 3) / 4;
    return (const uint4*)(in + (img0 * H * W + lo) * 4);
  };
  uint4 cp[NCP];
  int n16 = 0;
  int t = wg;
  if (t < tiles) {
    const uint4* src = patch_src(t, n16);
#pragma unroll
    for (int j = 0; j < NCP; ++j) cp[j] = src[min(tid + j * 256, n16 - 1)];
  }
  int buf = 0;
  for (; t < tiles; t += WPG, buf ^= 1) {
    // current tile's patch: registers -> LDS slot `buf`
    {
      float* dst = patch + buf * PATCH;
#pragma unroll
      for (int j = 0; j < NCP; ++j) {
        const int idx = tid + j * 256;
        if (idx < n16) *(uint4*)(dst + 4 * idx) = cp[j];
      }
    }
    __syncthreads();
    // next tile's patch: global -> registers, in flight under this tile's MFMAs
    const int tn = t + WPG;
    int n16n = 0;
    if (tn < tiles) {
      const uint4* src = patch_src(tn, n16n);
#pragma unroll
      for (int j = 0; j < NCP; ++j) cp[j] = src[min(tid + j * 256, n16n - 1)];
    }
    const int m0 = t * BM;
    const int mrow = min(m0 + l31, rows - 1);
    const float* pa = patch + buf * PATCH + (base_of(mrow) - base_of(m0));
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      // chunk 4wk + j: kernel row ky = 2wk + j/2, pixels kx = 4 (j & 1) + 2 half + {0, 1}
      const uint2 raw = *(const uint2*)(pa + (2 * wk + (j >> 1)) * W + 4 * (j & 1) + 2 * half);
      const float4 v0 = dz_u8x4_to_unit(raw.x), v1 = dz_u8x4_to_unit(raw.y);
      const float a[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
#pragma unroll
      for (int s = 0; s < 8; ++s)
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], wreg[j][s], acc, 0, 0, 0);
    }
    // exchange: every wave publishes its partial tile, then finishes registers 4wk..4wk+3
    float* rb = red + buf * RED;
    {
      float* dst = rb + (wk * 16) * 64 + lane;
#pragma unroll
      for (int i = 0; i < 16; ++i) dst[i * 64] = acc[i];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r = 4 * wk + i;
      const float* src = rb + r * 64 + lane;
      float v = src[0];
      v += src[16 * 64]; v += src[32 * 64]; v += src[48 * 64];
      v += bcol;
      v = v > 0.f ? v : 0.f;
      const int ml = m0 + dz_acc_row(r, lane);
      if (ml < rows) p.out[((long)z * rows + ml) * CO + l31] = v;
    }
    n16 = n16n;
  }
}
