// Forward convolutions on the LDS-DMA mainloop (dz_dma_gemm.h's scheme with an implicit-GEMM A
// operand): out[img, oh, ow, :] = relu(sum_k patch(img, oh, ow, k) W[k, :] + b)
// (ref: dqn_zoo/networks.py:194-198), float32 NHWC activations, HWIO weights = [K][CO].
//
// What differs from the register-staged skeleton (dz_gemm.h, ConvFwdOp):
//   * BOTH operands go global memory -> LDS by `global_load_lds_dwordx4` (1 KB per wave
//     instruction, no VGPR round trip, no LDS-write phase), NBUF stage buffers with NBUF - 1
//     stages in flight from t = 0, ONE barrier per stage;
//   * the A operand is the im2col view addressed PER LANE: a kernel row of the patch is KS*C
//     contiguous floats in NHWC (512 B for conv2, 768 B for conv3), a stage is BK of them, so a
//     lane's 16 bytes never straddle a kernel row; the walk over (kernel row, offset) is a
//     wave-uniform offset added to the lane's pixel base;
//   * LDS layouts and their source-side swizzles are dz_dma_gemm.h's (A depth-contiguous with
//     the 16-byte units XOR-swizzled by the row, B output-contiguous), fragment reads
//     conflict-free, k-slot permutation of the skeleton.
// Whole tiles only (B * OH * OW a multiple of 32: every learner batch of 32); other shapes keep
// ConvFwdOp.  4 waves = SUBN column sub-tiles x WKD = 4 / SUBN depth groups; the depth groups'
// accumulators meet in LDS at the end (all waves finish 16 / WKD accumulator registers each).
#pragma once

#include "dz_dma_gemm.h"
#include "dz_qnet_ops.h"

namespace {

template <int H_, int W_, int C_, int KS_, int S_, int OH_, int OW_, int CO_, int SUBN_, int KT_, int NBUF_, int NI_ = 1>
struct ConvDmaCfg {
  static constexpr int H = H_, W = W_, C = C_, KS = KS_, S = S_, OH = OH_, OW = OW_, CO = CO_;
  static constexpr int SUBN = SUBN_, WKD = 4 / SUBN_, KT = KT_, NBUF = NBUF_, NI = NI_;   // NI accumulator chains per wave
  static constexpr int BM = 32, BN = 32 * SUBN * NI, BK = 16 * WKD * KT;
  static constexpr int K = KS * KS * C, ROWLEN = KS * C, SPR = ROWLEN / BK, NST = K / BK;
  static constexpr int A_FLOATS = BM * BK, B_FLOATS = BK * BN, STAGE = A_FLOATS + B_FLOATS;
  static constexpr int EXCH = 4 * NI * 16 * 64;   // the depth groups' exchange (floats)
  static constexpr int LDS_FLOATS = NBUF * STAGE > EXCH ? NBUF * STAGE : EXCH;
  static constexpr int A_PER_WAVE = A_FLOATS / 1024, B_PER_WAVE = B_FLOATS / 1024;
  static constexpr int PER_STAGE = A_PER_WAVE + B_PER_WAVE;
  static constexpr int UPR = BK / 4;          // 16-byte units per A row
  static constexpr int UW = BN / 4;           // 16-byte units per B row
  static_assert(SUBN == 1 || SUBN == 2, "one or two 32-column sub-tiles");
  static_assert(ROWLEN % BK == 0 && K % BK == 0, "a stage never straddles a kernel row");
  static_assert(A_FLOATS % 1024 == 0 && B_FLOATS % 1024 == 0, "whole DMA instructions per wave");
  static_assert(UPR == 8 || UPR % 16 == 0, "A swizzle: 8 or a multiple of 16 units per row");
  static_assert(CO % BN == 0, "whole column tiles");
  static_assert(NBUF >= 2 && NBUF <= 4, "stage buffers");
  __device__ static int swz(int row) { return UPR == 8 ? ((row >> 1) & 7) : (row & 15); }
  static int tiles_per_group(int B) { return B * OH * OW / BM; }
  static bool fits(int B) { return (B * OH * OW) % BM == 0; }
};

struct ConvDmaParams {
  const float* in[DZ_MAX_GROUPS];   // per group: [images][H][W][C]
  int in_img_base[DZ_MAX_GROUPS];
  const float* w[DZ_MAX_GROUPS];    // [K][CO]
  const float* bias[DZ_MAX_GROUPS];
  float* out;                       // [G*B][OH][OW][CO]
  int B, G;
  long long* dbg;                   // (DZ_GEMM_STAMPS builds) per-workgroup wall-clock stamps
};

#ifdef DZ_GEMM_STAMPS
#define DZ_CD_STAMP(i) do { if (dbgp && threadIdx.x == 0) dbgp[i] = wall_clock64(); } while (0)
#else
#define DZ_CD_STAMP(i) do {} while (0)
#endif

// one workgroup = one BM x BN tile: row tile `mt` (over all groups), column tile `nt`
template <class C>
__device__ __forceinline__ void dz_conv_dma_body(const ConvDmaParams& p, int mt, int nt, float* lds) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wk = wave / C::SUBN, sub = wave % C::SUBN;
  const int half = lane >> 5, l31 = lane & 31;
  const unsigned lds0 = (unsigned)(uintptr_t)lds;
#ifdef DZ_GEMM_STAMPS
  long long* dbgp = p.dbg ? p.dbg + (long)(nt + 4 * mt) * 8 : nullptr;   // (tools/conv_stamps.py: x + 4 y)
#endif
  DZ_CD_STAMP(0);

  const int tpg = p.B * C::OH * C::OW / C::BM;
  const int g = mt / tpg;
  const int m0 = (mt - g * tpg) * C::BM;          // first row within the group
  const int n0 = nt * C::BN;
  const float* in = dz_pick3(p.in, g);
  const float* wgt = dz_pick3(p.w, g);
  const float* bias = dz_pick3(p.bias, g);
  const int img_base = dz_pick3(p.in_img_base, g);

  // the bias of this lane's output column: the ONLY ordinary global load, issued before the
  // first DMA (vmcnt counts in order: the first stage wait covers it)
  float bcol[C::NI];
#pragma unroll
  for (int ni = 0; ni < C::NI; ++ni) bcol[ni] = bias[n0 + (sub * C::NI + ni) * 32 + l31];

  // ---- DMA sources ------------------------------------------------------------------
  const float* asrc[C::A_PER_WAVE]; unsigned adst[C::A_PER_WAVE];
  const float* bsrc[C::B_PER_WAVE]; unsigned bdst[C::B_PER_WAVE];
#pragma unroll
  for (int i = 0; i < C::A_PER_WAVE; ++i) {
    const int idx = wave * C::A_PER_WAVE + i;
    constexpr int RPI = 64 / C::UPR;                       // rows per instruction
    const int r = idx * RPI + lane / C::UPR;
    const int u = (lane % C::UPR) ^ C::swz(r);
    const int m = m0 + r;
    const int img = m / (C::OH * C::OW), pix = m - img * (C::OH * C::OW);
    const int oh = pix / C::OW, ow = pix - oh * C::OW;
    asrc[i] = in + (((long)(img_base + img) * C::H + oh * C::S) * C::W + ow * C::S) * C::C + 4 * u;
    adst[i] = 4u * ((unsigned)idx * 256u);
  }
#pragma unroll
  for (int i = 0; i < C::B_PER_WAVE; ++i) {
    const int idx = wave * C::B_PER_WAVE + i;
    constexpr int RPI = 64 / C::UW;
    const int k = idx * RPI + lane / C::UW;
    const int cu = C::UW >= 16 ? ((lane % C::UW) ^ (8 * ((k >> 3) & 1))) : (lane % C::UW);
    bsrc[i] = wgt + (long)k * C::CO + n0 + 4 * cu;
    bdst[i] = 4u * ((unsigned)C::A_FLOATS + (unsigned)idx * 256u);
  }
  auto issue = [&](int st) {
    const unsigned base = lds0 + 4u * (unsigned)((st % C::NBUF) * C::STAGE);
    const int aoff = (st / C::SPR) * (C::W * C::C) + (st % C::SPR) * C::BK;   // (wave-uniform)
#pragma unroll
    for (int i = 0; i < C::A_PER_WAVE; ++i) dz_glds16<0>(asrc[i] + aoff, base + adst[i]);
#pragma unroll
    for (int j = 0; j < C::B_PER_WAVE; ++j) { dz_glds16<0>(bsrc[j], base + bdst[j]); bsrc[j] += (long)C::BK * C::CO; }
  };

  f32x16 acc[C::NI];
#pragma unroll
  for (int ni = 0; ni < C::NI; ++ni)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[ni][i] = 0.f;

#pragma unroll
  for (int s = 0; s < C::NBUF - 1; ++s)
    if (s < C::NST) issue(s);
  DZ_CD_STAMP(1);
#pragma unroll 1
  for (int st = 0; st < C::NST; ++st) {
    // stage st has landed (for THIS wave) when at most the younger stages' instructions are out
    const int ahead = min(C::NBUF - 2, C::NST - 1 - st);
    if (ahead >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * C::PER_STAGE) : "memory");
    else if (ahead == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(C::PER_STAGE) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();   // ... for every wave; and every wave has finished reading stage st - 1
    if (st == 0) DZ_CD_STAMP(2);
    if (st + C::NBUF - 1 < C::NST) issue(st + C::NBUF - 1);
    const float* As = lds + (st % C::NBUF) * C::STAGE;
    const float* Bs = As + C::A_FLOATS;
#ifndef DZ_CONV_DMA_PF   // 1: chunk kt + 1's fragments are requested in front of chunk kt's MFMAs
#define DZ_CONV_DMA_PF 1
#endif
    auto frag = [&](int kt, float (&fa)[8], float (&fb)[C::NI][8]) {
      const int ch = wk * C::KT + kt;
      {
        const int fr = C::swz(l31), u0 = ch * 4 + half * 2;
        const float4 v0 = *(const float4*)(As + l31 * C::BK + 4 * (u0 ^ fr));
        const float4 v1 = *(const float4*)(As + l31 * C::BK + 4 * ((u0 + 1) ^ fr));
        fa[0] = v0.x; fa[1] = v0.y; fa[2] = v0.z; fa[3] = v0.w; fa[4] = v1.x; fa[5] = v1.y; fa[6] = v1.z; fa[7] = v1.w;
      }
#pragma unroll
      for (int ni = 0; ni < C::NI; ++ni) {
        const int b0 = sub * C::NI + ni;
        const int blk = C::UW >= 16 ? (b0 ^ half) : b0;
        const float* bl = Bs + (ch * 16 + half * 8) * C::BN + blk * 32 + l31;
#pragma unroll
        for (int s = 0; s < 8; ++s) fb[ni][s] = bl[s * C::BN];
      }
    };
    float fa[2][8], fb[2][C::NI][8];
    frag(0, fa[0], fb[0]);
#pragma unroll
    for (int kt = 0; kt < C::KT; ++kt) {
      if (DZ_CONV_DMA_PF && kt + 1 < C::KT) {
        frag(kt + 1, fa[(kt + 1) & 1], fb[(kt + 1) & 1]);
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int s = 0; s < 8; ++s)
#pragma unroll
        for (int ni = 0; ni < C::NI; ++ni)
          acc[ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[kt & 1][s], fb[kt & 1][ni][s], acc[ni], 0, 0, 0);
      if (!DZ_CONV_DMA_PF && kt + 1 < C::KT) frag(kt + 1, fa[(kt + 1) & 1], fb[(kt + 1) & 1]);
    }
  }
  DZ_CD_STAMP(3);
  // ---- the depth groups meet in LDS; every wave finishes 16 / WKD registers -------------
  __syncthreads();
#pragma unroll
  for (int ni = 0; ni < C::NI; ++ni) {
    float* dst = lds + (((wk * C::SUBN + sub) * C::NI + ni) * 16) * 64 + lane;
#pragma unroll
    for (int i = 0; i < 16; ++i) dst[i * 64] = acc[ni][i];
  }
  __syncthreads();
  constexpr int RPW = 16 / C::WKD;
  const long orow0 = (long)g * tpg * C::BM + m0;
  DZ_CD_STAMP(4);
#pragma unroll
  for (int ni = 0; ni < C::NI; ++ni) {
    float b = bcol[ni];
    asm volatile("" : "+v"(b));
    const int col = n0 + (sub * C::NI + ni) * 32 + l31;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      if (i / RPW == wk) {   // wave-uniform
        const float* src = lds + ((sub * C::NI + ni) * 16 + i) * 64 + lane;
        float v = src[0];
#pragma unroll
        for (int k2 = 1; k2 < C::WKD; ++k2) v += src[k2 * C::SUBN * C::NI * 16 * 64];
        v += b;
        p.out[(orow0 + dz_acc_row(i, lane)) * C::CO + col] = v > 0.f ? v : 0.f;
      }
    }
  }
  DZ_CD_STAMP(5);
}

// grid: 1-D, tiles in XCD-aware order (the column tiles of one row tile -- which share its
// im2col rows -- run on one XCD back to back)
template <class C, int OCC>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(OCC, OCC)))
void dz_conv_dma_fwd_kernel(ConvDmaParams p, dim3 g) {
  __shared__ __attribute__((aligned(1024))) float lds[C::LDS_FLOATS];
  dim3 bid;
  if (!dz_xcd_tile(blockIdx.x, g, bid)) return;
  dz_conv_dma_body<C>(p, (int)bid.y, (int)bid.x, lds);
}
template <class C, int OCC>
static inline int dz_launch_conv_dma_fwd(const ConvDmaParams& p, hipStream_t s) {
  const dim3 g(C::CO / C::BN, (unsigned)(p.G * C::tiles_per_group(p.B)), 1);
  hipLaunchKernelGGL((dz_conv_dma_fwd_kernel<C, OCC>), dim3(dz_xcd_blocks(g)), dim3(256), 0, s, p, g);
  DZ_LAUNCH_CHECK();
  return DZ_OK;
}

}  // namespace
