// IQN fc1 forward at the reference sizes (h1 = relu(head_in @ W1 + b1): 5 120 x 3 136 x 512,
// 16.4 GFLOP -- the launch of the package that the fp32 matrix pipe bounds) with both operands
// going global memory -> LDS WITHOUT touching a register: LDS-DMA (global_load_lds_dwordx4, 1 KB
// per wave-instruction), NBUF stage buffers, ONE barrier per stage.
//
// Two tile sets in one launch, 256 workgroups each (dz_iqn.hip: `two_sets`): the online rows as
// 64 x 64 tiles (MI = 1, NI = 2: two column blocks per wave), the 3 072 target rows of the two s_t
// applies as 96 x 64 tiles (MI = 3, NI = 1: three row blocks per wave): two workgroups per CU, 2-3
// independent MFMA chains per wave -- the regime in which the chunk loop runs at 0.93 of the
// pipe's rate (tools/micro/lds_mfma_micro.hip) -- and at two waves per SIMD the prefetch has to
// be DEEPER than the register-staged skeleton's one stage, which is what the LDS buffers give.
//
//   * 4 waves = (WMW x WNW) output sub-tiles x 2 depth halves (wk) of every stage; stage = 32 KT
//     deep (KT chunks of 16 per wave); the depth halves are added through LDS at the end (the
//     skeleton's WK = 2 epilogue: wk = 0 stores);
//   * A stage in LDS: [BM rows][32 KT floats], the 16-byte units of a row XOR-swizzled by the
//     row (fA below) so that a fragment read -- 16 consecutive rows, the same logical unit, one
//     ds_read_b128 each -- touches 16 distinct 16-byte slots.  The DMA writes LDS linearly
//     (base + lane * 16), so the swizzle goes on the SOURCE address;
//   * B stage in LDS: [32 KT rows][64 columns] (256 bytes = all 64 banks per row), the two
//     32-column halves of a row swapped when bit 3 of the row is set: MFMA step s reads row s
//     (lanes 0-31) and row s + 8 (lanes 32-63, k-slot 8 + s) of a column block -- in opposite
//     halves of the banks;
//   * k-slot permutation and chunk order as in dz_gemm.h (lane half h takes k = 8 h + s at step
//     s): each output element's MFMA sequence is the skeleton's WK = 2 sequence.
// Whole tiles only; inline-assembly DMA with hand-counted s_waitcnt: no ordinary global load is
// in flight between the first DMA and the last wait (the bias is loaded behind it).
#pragma once

#include "dz_fc1_dgrad.h"   // dz_glds16
#include "dz_iqn_ops.h"

namespace {

struct IqnFc1DmaSet {
  const float* x;            // [rows][ldx], first row of the set
  int ldx;
  int tiles;                 // row tiles of the set
  const float* w;            // [K][ldw]
  const float* bias;         // [N]
  int ldw, K, N;
  float* out; int ldo;       // [rows][ldo], first row of the set
};

template <int MI_, int NI_, int WMW_, int WNW_, int KT_, int NBUF_>
struct IqnFc1DmaCfg {
  static constexpr int MI = MI_, NI = NI_, WMW = WMW_, WNW = WNW_, KT = KT_, NBUF = NBUF_;
  static constexpr int BM = 32 * MI * WMW, BN = 32 * NI * WNW, BK = 32 * KT;
  static constexpr int UPR = BK / 4;                       // 16-byte units per A row
  static constexpr int A_FLOATS = BM * BK, B_FLOATS = BK * BN, STAGE = A_FLOATS + B_FLOATS;
  static constexpr int LDS_FLOATS = NBUF * STAGE;
  static constexpr int A_PER_WAVE = A_FLOATS / 256 / 4, B_PER_WAVE = B_FLOATS / 256 / 4;
  static constexpr int PER_STAGE = A_PER_WAVE + B_PER_WAVE;   // DMA instructions per wave and stage
  static constexpr int RPI = 256 / BK;                        // A rows per DMA instruction
  static_assert(WMW * WNW == 2 && BN == 64 && (KT == 1 || KT == 2), "two sub-tiles x two depth halves");
  static_assert(A_FLOATS % 1024 == 0 && B_FLOATS % 1024 == 0, "whole DMA instructions per wave");
  static_assert(2 * MI * NI * 16 * 64 <= LDS_FLOATS, "the depth halves' exchange fits in the stage buffers");
  __device__ static int fA(int row) { return KT == 1 ? ((row >> 1) & 7) : (row & 15); }
};

template <class C>
__device__ __forceinline__ void iqn_fc1_dma_body(const IqnFc1DmaSet& p, const dim3& bid, float* lds) {
  const int row0 = bid.y * C::BM, n0 = bid.x * C::BN;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // (wave-uniform: the DMA's LDS base is an SGPR)
  const int wk = wave >> 1, sub = wave & 1;
  const int wm = C::WMW == 2 ? sub : 0, wn = C::WNW == 2 ? sub : 0;
  const int half = lane >> 5, l31 = lane & 31;
  const unsigned lds0 = (unsigned)(uintptr_t)lds;

  // ---- this wave's DMA sources (stage 0) and LDS destinations --------------------------------
  const float* asrc[C::A_PER_WAVE];
  unsigned adst[C::A_PER_WAVE];
#pragma unroll
  for (int i = 0; i < C::A_PER_WAVE; ++i) {
    const int ia = wave * C::A_PER_WAVE + i;
    const int r = ia * C::RPI + lane / C::UPR;
    const int u = (lane % C::UPR) ^ C::fA(r);
    asrc[i] = p.x + (long)(row0 + r) * p.ldx + 4 * u;
    adst[i] = 4u * (unsigned)(ia * 256);
  }
  const float* bsrc[C::B_PER_WAVE];
  unsigned bdst[C::B_PER_WAVE];
#pragma unroll
  for (int j = 0; j < C::B_PER_WAVE; ++j) {
    const int ib = wave * C::B_PER_WAVE + j;
    const int k = 4 * ib + (lane >> 4);
    const int cu = (lane & 15) ^ (8 * ((k >> 3) & 1));
    bsrc[j] = p.w + (long)k * p.ldw + n0 + 4 * cu;
    bdst[j] = 4u * (unsigned)(C::A_FLOATS + ib * 256);
  }
  const long a_step = C::BK, b_step = (long)C::BK * p.ldw;
  auto issue = [&](int buf) {
    const unsigned base = lds0 + 4u * (unsigned)(buf * C::STAGE);
#pragma unroll
    for (int i = 0; i < C::A_PER_WAVE; ++i) { dz_glds16<0>(asrc[i], base + adst[i]); asrc[i] += a_step; }
#pragma unroll
    for (int j = 0; j < C::B_PER_WAVE; ++j) { dz_glds16<0>(bsrc[j], base + bdst[j]); bsrc[j] += b_step; }
  };

  f32x16 acc[C::MI][C::NI];
#pragma unroll
  for (int mi = 0; mi < C::MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < C::NI; ++ni)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[mi][ni][i] = 0.f;
  const int nst = p.K / C::BK;
  // NBUF - 1 stages are in flight ahead of the one being consumed
  issue(0);
  if (C::NBUF > 2 && nst > 1) issue(1);
  for (int st = 0; st < nst; ++st) {
    // stage st has landed (for THIS wave) when at most the younger stage's instructions are out
    if (C::NBUF > 2 && st + 1 < nst) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(C::PER_STAGE) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();   // ... for every wave; and every wave has finished reading stage st - 1
    if (st + C::NBUF - 1 < nst) issue((st + C::NBUF - 1) % C::NBUF);   // into the buffer stage st - 1 was read from
    const float* As = lds + (st % C::NBUF) * C::STAGE;
    const float* Bs = As + C::A_FLOATS;
#pragma unroll
    for (int kt = 0; kt < C::KT; ++kt) {
      const int ch = wk * C::KT + kt;
      const int u0 = ch * 4 + half * 2;
      float fa[C::MI][8], fb[C::NI][8];
#pragma unroll
      for (int mi = 0; mi < C::MI; ++mi) {
        const int arow = (wm * C::MI + mi) * 32 + l31, fr = C::fA(arow);
        const float4 v0 = *(const float4*)(As + arow * C::BK + 4 * (u0 ^ fr));
        const float4 v1 = *(const float4*)(As + arow * C::BK + 4 * ((u0 + 1) ^ fr));
        fa[mi][0] = v0.x; fa[mi][1] = v0.y; fa[mi][2] = v0.z; fa[mi][3] = v0.w;
        fa[mi][4] = v1.x; fa[mi][5] = v1.y; fa[mi][6] = v1.z; fa[mi][7] = v1.w;
      }
#pragma unroll
      for (int ni = 0; ni < C::NI; ++ni) {
        const int nb = wn * C::NI + ni;
        const float* bl = Bs + (ch * 16 + half * 8) * 64 + ((nb ^ half) * 32 + l31);
#pragma unroll
        for (int s = 0; s < 8; ++s) fb[ni][s] = bl[s * 64];
      }
#pragma unroll
      for (int s = 0; s < 8; ++s)
#pragma unroll
        for (int mi = 0; mi < C::MI; ++mi)
#pragma unroll
          for (int ni = 0; ni < C::NI; ++ni)
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[mi][s], fb[ni][s], acc[mi][ni], 0, 0, 0);
    }
  }
  // ---- the two depth halves through LDS (skeleton's WK = 2 epilogue), bias, ReLU, store --------
  __syncthreads();
  float* red = lds + sub * (C::MI * C::NI * 16 * 64);
  if (wk == 1) {
#pragma unroll
    for (int mi = 0; mi < C::MI; ++mi)
#pragma unroll
      for (int ni = 0; ni < C::NI; ++ni)
#pragma unroll
        for (int i = 0; i < 16; ++i) red[((mi * C::NI + ni) * 16 + i) * 64 + lane] = acc[mi][ni][i];
  }
  __syncthreads();
  if (wk == 1) return;
#pragma unroll
  for (int mi = 0; mi < C::MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < C::NI; ++ni) {
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[mi][ni][i] += red[((mi * C::NI + ni) * 16 + i) * 64 + lane];
      const int col = n0 + (wn * C::NI + ni) * 32 + l31;
      const float b = p.bias[col];
      float* o = p.out + (long)(row0 + (wm * C::MI + mi) * 32) * p.ldo + col;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float v = acc[mi][ni][r] + b;
        o[(long)dz_acc_row(r, lane) * p.ldo] = v > 0.f ? v : 0.f;
      }
    }
}

template <class CA, class CB, int OCC>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(OCC, OCC)))
void iqn_fc1_fwd_dma2_kernel(IqnFc1DmaSet pa, dim3 ga, IqnFc1DmaSet pb, dim3 gb) {
  constexpr int SM = CA::LDS_FLOATS > CB::LDS_FLOATS ? CA::LDS_FLOATS : CB::LDS_FLOATS;
  __shared__ __attribute__((aligned(1024))) float lds[SM];
  const unsigned na = 8 * ga.x * ((ga.y * ga.z + 7) / 8);
  dim3 bid;
  if (blockIdx.x < na) {
    if (dz_xcd_tile(blockIdx.x, ga, bid)) iqn_fc1_dma_body<CA>(pa, bid, lds);
  } else {
    if (dz_xcd_tile(blockIdx.x - na, gb, bid)) iqn_fc1_dma_body<CB>(pb, bid, lds);
  }
}

}  // namespace
