// fp32-MFMA tile GEMM with BOTH operands going global memory -> LDS by LDS-DMA
// (global_load_lds_dwordx4: 1 KB per wave-instruction, no register staging, no LDS-write phase),
// NBUF stage buffers, ONE barrier per stage.  The mainloop of the large IQN contractions
// (dz_iqn.hip): few resident waves with 2-3 independent MFMA chains each -- the regime in which
// the chunk loop runs at 0.93 of the matrix pipe's rate (tools/micro/lds_mfma_micro.hip) and in
// which the register-staged skeleton's one-stage prefetch (dz_gemm.h) is exposed.
//
//   C[i][j] = sum_k A(i, k) B(k, j),   tile = (32 MI WMW) x (32 NI WNW), 4 waves =
//   (WMW x WNW = 2 sub-tiles) x (2 depth halves of every stage); stage = 32 KT deep; the depth
//   halves are added through LDS at the end (dz_gemm.h's WK = 2 epilogue: wk = 0 stores).
//
// Each operand is one of two memory shapes, and has an LDS stage layout whose fragment reads are
// bank-conflict free.  The DMA writes LDS linearly (base + lane * 16), so every swizzle goes on
// the SOURCE address:
//   * DEPTH-contiguous (`KC`: X[i][k], a row = an output row/column, e.g. activations as A, W^T
//     as B): stage = [32 x blocks rows][32 KT floats], the 16-byte units of a row XOR-swizzled by
//     the row (swz below): a fragment read -- 16 consecutive rows, the same logical unit, one
//     ds_read_b128 each -- touches 16 distinct 16-byte slots;
//   * OUTPUT-contiguous (`RC`: X[k][i], a row = a depth index, e.g. weights as B, activations as
//     A of a weight gradient): stage = [32 KT rows][32 x blocks floats] (blocks even), the 32-float
//     blocks of a row swapped in pairs when bit 3 of the row is set: MFMA step s reads row s
//     (lanes 0-31) and row s + 8 (lanes 32-63, k-slot 8 + s) of a block -- in opposite halves of
//     the banks.
// k-slot permutation and chunk order as in dz_gemm.h (lane half h takes k = 8 h + s at step s):
// each output element's MFMA sequence is the skeleton's WK = 2 sequence.
// Whole tiles only.  Inline-assembly DMA with hand-counted s_waitcnt (dz_glds16, dz_fc1_dgrad.h):
// no ordinary global load may be in flight between the first DMA and the last wait -- epilogue
// operands are loaded behind it.
#pragma once

#include "dz_fc1_dgrad.h"   // dz_glds16

namespace {

struct DzDmaOperands {
  const float* a; long lda;   // KC: a[(i0 + i) * lda + k];  RC: a[k * lda + i0 + i]  (a at the set's origin)
  const float* b; long ldb;   // KC: b[(j0 + j) * ldb + k];  RC: b[k * ldb + j0 + j]
  int K;                      // depth, a multiple of 32 KT
};

template <int MI_, int NI_, int WMW_, int WNW_, int KT_, int NBUF_, bool A_KC_, bool B_KC_>
struct DzDmaCfg {
  static constexpr int MI = MI_, NI = NI_, WMW = WMW_, WNW = WNW_, KT = KT_, NBUF = NBUF_;
  static constexpr bool A_KC = A_KC_, B_KC = B_KC_;
  static constexpr int BM = 32 * MI * WMW, BN = 32 * NI * WNW, BK = 32 * KT;
  static constexpr int A_FLOATS = BM * BK, B_FLOATS = BK * BN, STAGE = A_FLOATS + B_FLOATS;
  static constexpr int LDS_FLOATS = NBUF * STAGE;
  static constexpr int A_PER_WAVE = A_FLOATS / 256 / 4, B_PER_WAVE = B_FLOATS / 256 / 4;
  static constexpr int PER_STAGE = A_PER_WAVE + B_PER_WAVE;   // DMA instructions per wave and stage
  static_assert(WMW * WNW == 2 && (KT == 1 || KT == 2), "two sub-tiles x two depth halves");
  static_assert(A_FLOATS % 1024 == 0 && B_FLOATS % 1024 == 0, "whole DMA instructions per wave");
  static_assert(A_KC || BM % 64 == 0, "RC operands swap 32-float blocks in pairs");
  static_assert(B_KC || BN % 64 == 0, "RC operands swap 32-float blocks in pairs");
  static_assert(2 * MI * NI * 16 * 64 <= LDS_FLOATS, "the depth halves' exchange fits in the stage buffers");
  __device__ static int swz(int row) { return KT == 1 ? ((row >> 1) & 7) : (row & 15); }
};

// this wave's share of one operand's DMA instructions: source pointers (stage 0), LDS offsets
template <class C, bool KC, int EXT /* rows (KC) or floats per row (RC) */, int PER_WAVE>
__device__ __forceinline__ void dz_dma_sources(const float* x, long ld, int origin, int wave, int lane,
                                               unsigned lds_off_floats, const float* (&src)[PER_WAVE],
                                               unsigned (&dst)[PER_WAVE], long& step) {
#pragma unroll
  for (int i = 0; i < PER_WAVE; ++i) {
    const int idx = wave * PER_WAVE + i;
    if constexpr (KC) {
      constexpr int UPR = C::BK / 4, RPI = 64 / UPR;          // units per row, rows per instruction
      const int r = idx * RPI + lane / UPR;
      const int u = (lane % UPR) ^ C::swz(r);
      src[i] = x + (long)(origin + r) * ld + 4 * u;
    } else {
      constexpr int UW = EXT / 4, RPI = 64 / UW;              // units per row, rows per instruction
      const int k = idx * RPI + lane / UW;
      const int cu = (lane % UW) ^ (8 * ((k >> 3) & 1));
      src[i] = x + (long)k * ld + origin + 4 * cu;
    }
    dst[i] = 4u * (lds_off_floats + (unsigned)idx * 256u);
  }
  step = KC ? (long)C::BK : (long)C::BK * ld;
}

// one 32-wide block's fragment of chunk `ch` (8 k-slot values of this lane's row / column)
template <class C, bool KC, int EXT>
__device__ __forceinline__ void dz_dma_fragment(const float* stage, int blk, int ch, int half, int l31,
                                                float (&f)[8]) {
  if constexpr (KC) {
    const int row = blk * 32 + l31, fr = C::swz(row), u0 = ch * 4 + half * 2;
    const float4 v0 = *(const float4*)(stage + row * C::BK + 4 * (u0 ^ fr));
    const float4 v1 = *(const float4*)(stage + row * C::BK + 4 * ((u0 + 1) ^ fr));
    f[0] = v0.x; f[1] = v0.y; f[2] = v0.z; f[3] = v0.w; f[4] = v1.x; f[5] = v1.y; f[6] = v1.z; f[7] = v1.w;
  } else {
    const float* bl = stage + (ch * 16 + half * 8) * EXT + ((blk ^ half) * 32 + l31);
#pragma unroll
    for (int s = 0; s < 8; ++s) f[s] = bl[s * EXT];
  }
}

// Epi::store(ep, i_blk0, j_blk0, lane, acc): one finished 32x32 block whose first row / column
// (within the problem) are i_blk0 / j_blk0.
template <class C, class Epi>
__device__ __forceinline__ void dz_dma_gemm_body(const DzDmaOperands& p, const typename Epi::Params& ep,
                                                 int i0, int j0, float* lds) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // (wave-uniform: the DMA's LDS base is an SGPR)
  const int wk = wave >> 1, sub = wave & 1;
  const int wm = C::WMW == 2 ? sub : 0, wn = C::WNW == 2 ? sub : 0;
  const int half = lane >> 5, l31 = lane & 31;
  const unsigned lds0 = (unsigned)(uintptr_t)lds;

  const float* asrc[C::A_PER_WAVE]; unsigned adst[C::A_PER_WAVE]; long a_step;
  const float* bsrc[C::B_PER_WAVE]; unsigned bdst[C::B_PER_WAVE]; long b_step;
  dz_dma_sources<C, C::A_KC, C::BM, C::A_PER_WAVE>(p.a, p.lda, i0, wave, lane, 0u, asrc, adst, a_step);
  dz_dma_sources<C, C::B_KC, C::BN, C::B_PER_WAVE>(p.b, p.ldb, j0, wave, lane, (unsigned)C::A_FLOATS, bsrc, bdst, b_step);
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
      float fa[C::MI][8], fb[C::NI][8];
#pragma unroll
      for (int mi = 0; mi < C::MI; ++mi) dz_dma_fragment<C, C::A_KC, C::BM>(As, wm * C::MI + mi, ch, half, l31, fa[mi]);
#pragma unroll
      for (int ni = 0; ni < C::NI; ++ni) dz_dma_fragment<C, C::B_KC, C::BN>(Bs, wn * C::NI + ni, ch, half, l31, fb[ni]);
#pragma unroll
      for (int s = 0; s < 8; ++s)
#pragma unroll
        for (int mi = 0; mi < C::MI; ++mi)
#pragma unroll
          for (int ni = 0; ni < C::NI; ++ni)
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[mi][s], fb[ni][s], acc[mi][ni], 0, 0, 0);
    }
  }
  // ---- the two depth halves through LDS (dz_gemm.h's WK = 2 epilogue), then the Op's store -------
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
      Epi::store(ep, i0 + (wm * C::MI + mi) * 32, j0 + (wn * C::NI + ni) * 32, lane, acc[mi][ni]);
    }
}

// Two tile sets in one launch, tiles of each in XCD-aware order (dz_xcd_tile: grid x = column
// tiles, y = row tiles).
template <class CA, class EA, class CB, class EB, int OCC>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(OCC, OCC)))
void dz_dma_gemm2_kernel(DzDmaOperands pa, typename EA::Params ea, dim3 ga,
                         DzDmaOperands pb, typename EB::Params eb, dim3 gb) {
  constexpr int SM = CA::LDS_FLOATS > CB::LDS_FLOATS ? CA::LDS_FLOATS : CB::LDS_FLOATS;
  __shared__ __attribute__((aligned(1024))) float lds[SM];
  const unsigned na = 8 * ga.x * ((ga.y * ga.z + 7) / 8);
  dim3 bid;
  if (blockIdx.x < na) {
    if (dz_xcd_tile(blockIdx.x, ga, bid)) dz_dma_gemm_body<CA, EA>(pa, ea, bid.y * CA::BM, bid.x * CA::BN, lds);
  } else {
    if (dz_xcd_tile(blockIdx.x - na, gb, bid)) dz_dma_gemm_body<CB, EB>(pb, eb, bid.y * CB::BM, bid.x * CB::BN, lds);
  }
}
template <class CA, class EA, class CB, class EB, int OCC>
static inline int dz_launch_dma_gemm2(const DzDmaOperands& pa, const typename EA::Params& ea, dim3 ga,
                                      const DzDmaOperands& pb, const typename EB::Params& eb, dim3 gb,
                                      hipStream_t s) {
  hipLaunchKernelGGL((dz_dma_gemm2_kernel<CA, EA, CB, EB, OCC>), dim3(dz_xcd_blocks(ga) + dz_xcd_blocks(gb)),
                     dim3(256), 0, s, pa, ea, ga, pb, eb, gb);
  DZ_LAUNCH_CHECK();
  return DZ_OK;
}

}  // namespace
