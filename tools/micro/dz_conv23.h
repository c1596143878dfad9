// conv2 -> conv3 of the DQN torso in ONE launch (experiment, round 2).
//
// One workgroup (8 waves) per (group, image, band): the image is cut into two
// horizontal bands of conv3 output rows ([0,4) and [4,7)); a band needs conv2 rows
// [0,6) / [4,9) (recomputing one shared row instead of exchanging it between
// workgroups) and act1 rows [0,14) / [8,20).  The act1 band is copied flat into LDS
// once (two planes by pixel parity, PITCH 36: conflict-free 16-byte fragment reads
// for the stride-2 layer), conv2's output band stays in LDS (PITCH 68) and is also
// written out for the backward pass (each conv2 row by exactly one band), conv3
// reads it from there.  Weights never touch LDS: lane (n, h) streams
// W[k0 + 8h + s][n0 + n] straight into its MFMA B register, 3 chunks of 8 in flight.
//   conv2: 4 tiles (2 x 32 rows, 2 x 32 cols) x 2 K-halves   = 8 waves, 128 MFMAs each
//   conv3: 2 tiles (32 rows, 2 x 32 cols)     x 4 K-quarters = 8 waves,  72 MFMAs each
// i.e. 12.8 k MFMA cycles per wave, 2 waves per SIMD: 10.7 us if the pipe never waits.
#pragma once

#include "dz_qnet_ops.h"

#ifndef DZ_C23_STAMP
#define DZ_C23_STAMP(i)
#endif

struct Conv23Params {
  const float* act1[DZ_MAX_GROUPS];   // [images][20][20][32]
  int img_base[DZ_MAX_GROUPS];
  const float* w2[DZ_MAX_GROUPS]; const float* b2[DZ_MAX_GROUPS];   // [512][64], [64]
  const float* w3[DZ_MAX_GROUPS]; const float* b3[DZ_MAX_GROUPS];   // [576][64], [64]
  float* act2;                        // [G*B][9][9][64]
  float* feat;                        // [G*B][7][7][64]
  int B, G;
  int act2_groups;                   // bit g: group g's conv2 output is written out
};

namespace conv23 {
constexpr int P1 = 36, PLANE1 = 140 * P1;       // act1 band: <= 280 pixels, 2 planes
constexpr int IN1 = 2 * PLANE1;                 // 10080 floats
constexpr int P2 = 68, A2 = 54 * P2;            // conv2 band: <= 54 pixels
constexpr int SCR = 8 * 1024;                   // K-split partial tiles
constexpr int SMEM = IN1 + A2 + SCR;            // 21944 floats = 87.8 KB
constexpr int DEPTH = 3;

// acc += A . W over NCH 16-deep chunks starting at chunk c0.  A(m, k) comes from the
// LDS-resident input through `addr(tap_y, tap_x)` (this lane's row), k = (ky, kx, c).
// chain_first issues the first DEPTH weight chunks (callable before the input is ready).
__device__ __forceinline__ void chain_first(const float* __restrict__ w, int col, int c0,
                                            int half, float (&wb)[DEPTH][8]) {
#pragma unroll
  for (int c = 0; c < DEPTH; ++c) {
    const float* src = w + (long)(16 * (c0 + c) + 8 * half) * 64 + col;
#pragma unroll
    for (int s = 0; s < 8; ++s) wb[c][s] = src[s * 64];
  }
}
template <int C, int KS, int NCH, class Addr>
__device__ __forceinline__ void chain_run(const float* __restrict__ w, int col, int c0,
                                          int half, Addr addr, float (&wb)[DEPTH][8],
                                          f32x16& acc) {
  constexpr int CO = 64;
  auto issue = [&](int c, float (&b)[8]) {
    const float* src = w + (long)(16 * (c0 + c) + 8 * half) * CO + col;
#pragma unroll
    for (int s = 0; s < 8; ++s) b[s] = src[s * CO];
  };
  struct Frag { float a[8]; };
  auto fetch = [&](int c, Frag& f) {
    const int k0 = 16 * (c0 + c);
    const int tap = k0 / C, cc = k0 % C;
    const float* src = addr(tap / KS, tap % KS) + cc + 8 * half;
    const float4 v0 = *(const float4*)src, v1 = *(const float4*)(src + 4);
    f.a[0] = v0.x; f.a[1] = v0.y; f.a[2] = v0.z; f.a[3] = v0.w;
    f.a[4] = v1.x; f.a[5] = v1.y; f.a[6] = v1.z; f.a[7] = v1.w;
  };
  Frag fr[2];
  fetch(0, fr[0]);
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    float (&b)[8] = wb[c % DEPTH];
    if (c + 1 < NCH) fetch(c + 1, fr[(c + 1) & 1]);
    const Frag& f = fr[c & 1];
#pragma unroll
    for (int s = 0; s < 8; ++s)
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a[s], b[s], acc, 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    if (c + DEPTH < NCH) issue(c + DEPTH, b);
    __builtin_amdgcn_sched_barrier(0);
  }
}
}  // namespace conv23

__global__ __launch_bounds__(512) void conv23_fused_kernel(Conv23Params p) {
  using namespace conv23;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* in1 = smem;
  float* a2 = smem + IN1;
  float* scr = smem + IN1 + A2;
  const int band = blockIdx.x, img = blockIdx.y, z = blockIdx.z;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5, l31 = lane & 31;
  const float* __restrict__ act1 = dz_pick3(p.act1, z);
  const float* __restrict__ w2 = dz_pick3(p.w2, z);
  const float* __restrict__ b2 = dz_pick3(p.b2, z);
  const float* __restrict__ w3 = dz_pick3(p.w3, z);
  const float* __restrict__ b3 = dz_pick3(p.b3, z);
  const int img_in = dz_pick3(p.img_base, z) + img;
  const int img_out = z * p.B + img;

  const int r3a = band ? 4 : 0, r3n = band ? 3 : 4;     // conv3 rows of this band
  const int r2a = band ? 4 : 0, r2n = band ? 5 : 6;     // conv2 rows it needs
  const int own_lo = band ? 5 : 0, own_hi = band ? 9 : 5;  // conv2 rows it writes out
  const int r1a = 2 * r2a, r1n = 2 * r2n + 2;            // act1 rows (k4 s2)
  const int npx2 = r2n * 9, npx3 = r3n * 7;

  DZ_C23_STAMP(0);
  // conv2 roles: tile = wave & 3 (row tile, column tile), K half = wave >> 2
  const int tile2 = wave & 3, mt2 = tile2 >> 1, nt2 = tile2 & 1, kh = wave >> 2;
  const int col2 = nt2 * 32 + l31;
  // conv3 roles: column tile = wave & 1, K quarter = wave >> 1
  const int nt3 = wave & 1, kq = wave >> 1;
  const int col3 = nt3 * 32 + l31;
  // ---- every load that does not depend on LDS contents, up front -------------------
  float4 cp[5];
  const int n4 = r1n * 20 * 8;
  {
    const float4* src = (const float4*)(act1 + ((long)img_in * 400 + r1a * 20) * 32);
#pragma unroll
    for (int j = 0; j < 5; ++j) cp[j] = src[min(tid + j * 512, n4 - 1)];
  }
  const float bias2 = b2[col2], bias3 = b3[col3];
  float wb[DEPTH][8];
  chain_first(w2, col2, 16 * kh, half, wb);
  __builtin_amdgcn_sched_barrier(0);
  // ---- act1 band -> LDS (flat copy, two planes by pixel parity) ----------------------
#pragma unroll
  for (int j = 0; j < 5; ++j) {
    const int idx = tid + j * 512;
    if (idx < n4) {
      const int px = idx >> 3, q = idx & 7;
      *(float4*)(in1 + (px >> 1) * P1 + (px & 1) * PLANE1 + 4 * q) = cp[j];
    }
  }
  __syncthreads();
  DZ_C23_STAMP(1);

  // ---- conv2 ---------------------------------------------------------------------------
  {
    const int mc = min(mt2 * 32 + l31, npx2 - 1);
    const int pb = (2 * (mc / 9)) * 20 + 2 * (mc % 9);   // even
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    auto addr = [&](int ky, int kx) {
      const int px = pb + ky * 20 + kx;
      return (const float*)(in1 + (px >> 1) * P1 + (px & 1) * PLANE1);
    };
    chain_run<32, 4, 16>(w2, col2, 16 * kh, half, addr, wb, acc);
    DZ_C23_STAMP(2);
    chain_first(w3, col3, 9 * kq, half, wb);   // conv3's first weights ride under the epilogue
    // both K halves exchange 8 accumulator registers and finish the other 8
    {
      float* dst = scr + (tile2 * 2 + kh) * 512 + lane;
#pragma unroll
      for (int i = 0; i < 8; ++i) dst[i * 64] = acc[(1 - kh) * 8 + i];
    }
    __syncthreads();
    {
      const float* src = scr + (tile2 * 2 + (1 - kh)) * 512 + lane;
      const bool to_global = (p.act2_groups >> z) & 1;
      float* gout = p.act2 + ((long)img_out * 81 + r2a * 9) * 64 + col2;
      const int glo = (own_lo - r2a) * 9, ghi = (own_hi - r2a) * 9;   // owned rows, band-local
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int r = kh * 8 + i;
        const int row = mt2 * 32 + dz_acc_row(r, lane);
        // K half 0 first, as a single-wave chain would add them
        float v = kh == 0 ? acc[r] + src[i * 64] : src[i * 64] + acc[r];
        v += bias2;
        v = v > 0.f ? v : 0.f;
        if (row < npx2) {
          a2[row * P2 + col2] = v;
          if (to_global && row >= glo && row < ghi) gout[(long)row * 64] = v;
        }
      }
    }
  }
  __syncthreads();
  DZ_C23_STAMP(3);

  // ---- conv3 ---------------------------------------------------------------------------
  {
    const int mc = min(l31, npx3 - 1);
    const int pb = (mc / 7) * 9 + mc % 7;
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    auto addr = [&](int ky, int kx) { return (const float*)(a2 + (pb + ky * 9 + kx) * P2); };
    chain_run<64, 3, 9>(w3, col3, 9 * kq, half, addr, wb, acc);
    DZ_C23_STAMP(4);
    // the four K quarters exchange through LDS; quarter q finishes registers 4q..4q+3
    {
      float* dst = scr + ((nt3 * 4 + kq) * 16) * 64 + lane;
#pragma unroll
      for (int i = 0; i < 16; ++i) dst[i * 64] = acc[i];
    }
    __syncthreads();
    {
      float* gout = p.feat + ((long)img_out * 49 + r3a * 7) * 64 + col3;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int r = kq * 4 + i;
        const float* src = scr + ((nt3 * 4) * 16 + r) * 64 + lane;
        float v = src[0];
        v += src[1024]; v += src[2048]; v += src[3072];
        v += bias3;
        v = v > 0.f ? v : 0.f;
        const int row = dz_acc_row(r, lane);
        if (row < npx3) gout[(long)row * 64] = v;
      }
    }
  }
  DZ_C23_STAMP(5);
}
