// Forward convolution with the input patch resident in LDS (round 2).
//
// The implicit-GEMM ConvFwdOp materialises im2col rows: every workgroup pulls
// BM x K floats of A through L2 -> registers -> LDS (conv3: 73 KB for a 32-row
// tile, 21.6 MB per launch for 2 MB of activations, because each input pixel is
// re-read once per filter tap that covers it), and the same again for its slice of
// the weights.  The per-workgroup trace and the load ablation in
// tools/micro/conv_micro.hip put 2-2.6 us of each ~11 us forward launch on those A
// loads and ~1 us on B, issued up front or not: the launch is bound by L2 -> CU
// bytes, not by latency.  Here
//   * the BM consecutive output pixels of a tile read a CONTIGUOUS range of input
//     pixels (NHWC images are back to back, and pixel -> first-input-pixel is
//     monotone), which is copied once, flat and fully coalesced, into LDS
//     (conv3: <= 25 KB instead of 73 KB; conv2 29 vs 65 KB; conv1 8 vs 16 KB of
//     bytes) and never rewritten: ONE barrier per workgroup instead of two per stage;
//   * the MFMA A fragment of (row m, tap, channel chunk) is read from
//     LDS[(pix(m) + tap offset) * PITCH + channel], PITCH = C + 4 words, the same
//     two 16-byte reads per 8 MFMAs as the KC tile layout;
//   * weights never touch LDS: lane (n, h) loads W[k0 + 8h + s][n0 + n] for step s
//     straight into its B operand register (128 contiguous bytes per half wave),
//     as a software pipeline of 16-deep chunks like dz_fc_stream.h.
// Tile shape, the split of the reduction over the WK waves, the order of the chunks
// within a wave and the epilogue sum are those of ConvFwdOp<same arguments>, so the
// outputs are bit-identical to the implicit-GEMM kernel (checked in the micro
// benchmark and by tests/test_rainbow_gpu.py::test_conv_patch_matches_gemm).
#pragma once

#include "dz_qnet_ops.h"

#ifndef DZ_PATCH_STAMP
#define DZ_PATCH_STAMP(i)
#endif

template <int IN_U8, int H, int W, int C, int KS, int S, int OH, int OW, int CO,
          int WM_, int WN_, int WK_, int KT_>
struct ConvPatchFwdOp {
  static constexpr int WM = WM_, WN = WN_, WK = WK_, KT = KT_, CPS = WK_ * KT_;
  static constexpr int BM = 32 * WM, BN = 32 * WN, BK = 16 * CPS;
  static constexpr int K = KS * KS * C;
  static constexpr int NST = K / BK;        // stages of the equivalent ConvFwdOp
  static constexpr int NCH = NST * KT;      // 16-deep chunks per wave
  static constexpr int DEPTH = NCH < 3 ? NCH : 3;  // weight chunks in flight per wave
  static_assert(K % BK == 0, "K must be a multiple of the stage depth");
  static_assert(IN_U8 ? (C == 4 && KS == 8 && S % 4 == 0) : (C % 16 == 0), "chunk within a tap");
  static_assert(CO % BN == 0, "column tiles are full");
  static_assert(WM * WN * WK == 4, "4 waves per workgroup");
  typedef ConvFwdParams Params;

  // first input pixel (flat index over images) of output pixel m
  static constexpr int base_of(int m) {
    return (m / (OH * OW)) * H * W + ((m % (OH * OW)) / OW) * S * W + ((m % (OH * OW)) % OW) * S;
  }
  static constexpr int max_span() {
    int mx = 0;
    for (int m0 = 0; m0 < OH * OW; ++m0) {
      const int sp = base_of(m0 + BM - 1) - base_of(m0) + (KS - 1) * W + KS;
      mx = sp > mx ? sp : mx;
    }
    return mx;
  }
  static constexpr int MAXPX = max_span();
  // LDS words per pixel.  Float patches are split into S planes by pixel index mod S
  // (slot(px) = (px / S) * PITCH + (px % S) * PLANE): the 32 rows of a fragment read
  // are S pixels apart, and (S * PITCH) words between lanes would put lanes i and
  // i + 8 on the same banks for S = 2; within a plane consecutive rows are PITCH =
  // C + 4 words apart, conflict-free for 16-byte reads like the KC tile pitch of 20
  static constexpr int PITCH = IN_U8 ? 1 : C + 4;
  static constexpr int PLANE = IN_U8 ? 0 : ((MAXPX + S - 1) / S) * PITCH;
  static constexpr int PATCH = IN_U8 ? ((MAXPX + 3) / 4) * 4 : S * PLANE;
  __device__ static int slot(int px) { return (px / S) * PITCH + (px % S) * PLANE; }
  static constexpr int RED = WK > 1 ? WK * WM * WN * 16 * 64 : 0;
  static constexpr int SMEM_ELEMS = PATCH > RED ? PATCH : RED;
  // 16-byte copy slots per thread
  static constexpr int NCP = ((IN_U8 ? (MAXPX + 3) / 4 : MAXPX * (C / 4)) + 255) / 256;

  static int tiles_per_group(int B) { return (B * OH * OW + BM - 1) / BM; }

  __device__ static void body(const Params& p, const dim3& bid, float* smem) {
    const int rows = p.B * OH * OW;
    const int tpg = (rows + BM - 1) / BM;
    const int z = bid.y / tpg;
    if (z >= p.G) return;
    const int m0 = (bid.y % tpg) * BM;
    const int n0 = bid.x * BN;
    const void* in = dz_pick3(p.in, z);
    const float* __restrict__ w = dz_pick3(p.w, z);
    const float* __restrict__ bias = dz_pick3(p.bias, z);
    const int img_base = dz_pick3(p.in_img_base, z);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // chunk addresses stay scalar
    const int wk = wave / (WM * WN), wm = (wave % (WM * WN)) / WN, wn = wave % WN;
    const int half = lane >> 5, l31 = lane & 31;

    DZ_PATCH_STAMP(0);
    const int m_last = min(m0 + BM, rows) - 1;
    const int lo = base_of(m0);                       // runtime use of the same formula
    const int npx = base_of(m_last) + (KS - 1) * W + KS - lo;

    // (1) the patch: a flat 16-byte copy of input pixels [lo, lo + npx)
    uint4 cp[NCP];
    const int n16 = IN_U8 ? (npx + 3) / 4 : npx * (C / 4);
    {
      const uint4* src = IN_U8
          ? (const uint4*)((const uint8_t*)in + ((long)img_base * H * W + lo) * 4)
          : (const uint4*)((const float*)in + ((long)img_base * H * W + lo) * C);
#pragma unroll
      for (int j = 0; j < NCP; ++j) cp[j] = src[min(tid + j * 256, n16 - 1)];
    }
    // (2) the epilogue's bias and the first weight chunks
    const int col = n0 + wn * 32 + l31;
    const float bcol = bias[col];
    // chunk c of this wave is chunk (c / KT) * CPS + wk * KT + c % KT of the reduction
    auto k_of = [&](int c) { return 16 * ((c / KT) * CPS + wk * KT + (c % KT)); };
    float wb[DEPTH][8];
    auto issue = [&](int c, float (&b)[8]) {
      const float* src = w + (long)(k_of(c) + 8 * half) * CO + col;
#pragma unroll
      for (int s = 0; s < 8; ++s) b[s] = src[s * CO];
    };
#pragma unroll
    for (int c = 0; c < DEPTH; ++c) issue(c, wb[c]);
    __builtin_amdgcn_sched_barrier(0);

    DZ_PATCH_STAMP(1);
    // (3) patch -> LDS
    {
#pragma unroll
      for (int j = 0; j < NCP; ++j) {
        const int idx = tid + j * 256;
        if (idx < n16) {
          if constexpr (IN_U8) {
            *(uint4*)(smem + 4 * idx) = cp[j];
          } else {
            const int px = idx / (C / 4), q = idx % (C / 4);
            *(uint4*)(smem + slot(px) + 4 * q) = cp[j];
          }
        }
      }
    }
    DZ_PATCH_STAMP(2);
    __syncthreads();
    DZ_PATCH_STAMP(3);

    // this lane's A row: output pixel m0 + wm*32 + l31 (clamped; surplus rows are
    // computed from a valid pixel and never stored)
    const int mrow = min(m0 + wm * 32 + l31, rows - 1);
    const int pb = base_of(mrow) - lo;
    struct Frag { float a[8]; };
    auto fetch = [&](int c, Frag& f) {
      const int k0 = k_of(c);
      if constexpr (IN_U8) {
        // 16 reduction indices = 4 pixels of one kernel row; half h takes pixels 2h, 2h+1
        const int ky = k0 / (KS * C), kx = (k0 % (KS * C)) / C;
        const uint2 raw = *(const uint2*)(smem + pb + ky * W + kx + 2 * half);
        const float4 v0 = dz_u8x4_to_unit(raw.x), v1 = dz_u8x4_to_unit(raw.y);
        f.a[0] = v0.x; f.a[1] = v0.y; f.a[2] = v0.z; f.a[3] = v0.w;
        f.a[4] = v1.x; f.a[5] = v1.y; f.a[6] = v1.z; f.a[7] = v1.w;
      } else {
        const int tap = k0 / C, cc = k0 % C;
        const float* src = smem + slot(pb + (tap / KS) * W + tap % KS) + cc + 8 * half;
        const float4 v0 = *(const float4*)src, v1 = *(const float4*)(src + 4);
        f.a[0] = v0.x; f.a[1] = v0.y; f.a[2] = v0.z; f.a[3] = v0.w;
        f.a[4] = v1.x; f.a[5] = v1.y; f.a[6] = v1.z; f.a[7] = v1.w;
      }
    };

    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
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

    DZ_PATCH_STAMP(4);
    // epilogue: the WK partial tiles are exchanged through LDS and every wave
    // finishes 16/WK accumulator registers (sum over k-groups 0..WK-1 in order)
    unsigned rmask = 0xffffu;
    if constexpr (WK > 1) {
      __syncthreads();
      constexpr int PER = WM * WN, RPW = 16 / WK;
      float* red = smem;
      {
        float* dst = red + ((wk * PER + wm * WN + wn) * 16) * 64 + lane;
#pragma unroll
        for (int i = 0; i < 16; ++i) dst[i * 64] = acc[i];
      }
      __syncthreads();
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        float v = 0.f;
        if (i / RPW == wk) {
          const float* src = red + ((wm * WN + wn) * 16 + i) * 64 + lane;
          v = src[0];
#pragma unroll
          for (int k2 = 1; k2 < WK; ++k2) v += src[(long)k2 * PER * 16 * 64];
        }
        acc[i] = v;
      }
      rmask = ((1u << RPW) - 1u) << (wk * RPW);
    }
    DZ_PATCH_STAMP(5);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int ml = m0 + wm * 32 + dz_acc_row(r, lane);
      if (((rmask >> r) & 1u) && ml < rows) {
        const float v = acc[r] + bcol;
        p.out[((long)z * rows + ml) * CO + col] = v > 0.f ? v : 0.f;
      }
    }
    DZ_PATCH_STAMP(6);
  }
};
