// The LDS-DMA tile GEMM as an Op skeleton (the counterpart of dz_gemm.h's register-staged
// skeleton): both operands go global memory -> LDS by `global_load_lds_dwordx4`, NBUF stage
// buffers with NBUF - 1 stages in flight from the first instruction, ONE barrier per stage,
// fragment reads bank-conflict free through source-side swizzles (layouts: dz_dma_gemm.h).
// An Op supplies, per 16-byte DMA granule, the ADDRESS it comes from -- so implicit-GEMM views
// (im2col gathers, transposed-convolution gathers with out-of-range taps reading a page of
// zeros, split-K ranges with ragged ends) need no register staging either.
//
//   4 waves = SUBM x SUBN sub-tiles x WKD depth groups (product 4); a wave owns MI x NI
//   accumulators (32 x 32 each); stage depth BK = 16 WKD KT; tile = (32 MI SUBM) x (32 NI SUBN).
//
// Op interface (all static, __device__):
//   constants   MI, NI, SUBM, SUBN, WKD, KT, NBUF, A_KC, B_KC
//   struct Params;  struct Tile { int nst; ... };  bool tile(p, bid, Tile&)
//   const float* a_src(p, t, st, r, u)   the granule of stage st at
//        KC operand: tile row r, 16-byte unit u of the stage's depth (u < BK / 4)
//        RC operand: depth row r of the stage (r < BK), unit u of the tile's rows (u < BM / 4)
//   const float* b_src(p, t, st, r, u)   (the same for B; "rows" are tile columns)
//   struct Pre;  Pre prefetch(p, t, bi, bj, lane, rmask)   ordinary loads, issued BEFORE the first
//        DMA (vmcnt counts in order: the first stage wait covers them); bi / bj = this wave's
//        first 32-block of the tile
//   void store(p, t, bi, bj, lane, acc, rmask, pre)     one finished 32 x 32 block (block
//        coordinates within the tile); only accumulator registers r with bit r of rmask set
//   optional  A_U8 = 1 (with A_KC = false): the A operand is uint8 (conv1's frames): a depth row is
//        BM BYTES, LDS holds the raw bytes [BK][BM], the fragment read converts (`x / 255` exactly,
//        dz_div255) -- a_src returns the address of 16 BYTES (unit u < BM / 16).
// Every address an Op returns must be 16-byte aligned and readable (masked granules point at
// dz_page_zero / dz_page_one / dz_page_u8one).
#pragma once

#include "dz_dma_gemm.h"
#include "dz_qnet_ops.h"   // dz_div255

namespace {

template <class Op, class = void> struct DzAU8 { static constexpr int v = 0; };
template <class Op> struct DzAU8<Op, decltype((void)Op::A_U8)> { static constexpr int v = Op::A_U8; };

template <class Op>
struct DzDmaOpShape {
  static constexpr bool A_U8 = DzAU8<Op>::v != 0;
  static constexpr int MI = Op::MI, NI = Op::NI, SUBM = Op::SUBM, SUBN = Op::SUBN, WKD = Op::WKD;
  static constexpr int KT = Op::KT, NBUF = Op::NBUF;
  static constexpr int BM = 32 * MI * SUBM, BN = 32 * NI * SUBN, BK = 16 * WKD * KT;
  static constexpr int A_FLOATS = A_U8 ? BM * BK / 4 : BM * BK, B_FLOATS = BK * BN, STAGE = A_FLOATS + B_FLOATS;
  static constexpr int A_PER_WAVE = A_FLOATS / 1024, B_PER_WAVE = B_FLOATS / 1024;
  static constexpr int PER_STAGE = A_PER_WAVE + B_PER_WAVE;
  static constexpr bool SPLIT = WKD > 1 && MI == 1 && NI == 1;   // distributed epilogue
  static constexpr int EXCH = WKD == 1 ? 0 : (SPLIT ? WKD : WKD - 1) * SUBM * SUBN * MI * NI * 1024;
  static constexpr int LDS_FLOATS = NBUF * STAGE > EXCH ? NBUF * STAGE : EXCH;
  static constexpr int UPR = BK / 4;   // 16-byte units per row of a depth-contiguous operand
  static_assert(SUBM * SUBN * WKD == 4, "4 waves");
  static_assert(A_FLOATS % 1024 == 0 && B_FLOATS % 1024 == 0, "whole DMA instructions per wave");
  static_assert(UPR == 8 || UPR % 16 == 0, "KC swizzle: 8 or a multiple of 16 units per row");
  static_assert(NBUF >= 2 && NBUF <= 4, "stage buffers");
  __device__ static int swz(int row) { return UPR == 8 ? ((row >> 1) & 7) : (row & 15); }
};

// the granule (r, u) that lane `lane` of DMA instruction `idx` of an operand fetches
template <class S, bool KC, int EXT>
__device__ __forceinline__ void dz_dmaop_granule(int idx, int lane, int& r, int& u) {
  if constexpr (KC) {
    constexpr int RPI = 64 / S::UPR;
    r = idx * RPI + lane / S::UPR;
    u = (lane % S::UPR) ^ S::swz(r);
  } else {
    constexpr int UW = EXT / 4, RPI = 64 / UW;
    r = idx * RPI + lane / UW;
    u = UW >= 16 ? ((lane % UW) ^ (8 * ((r >> 3) & 1))) : (lane % UW);
  }
}

template <class S, bool KC, int EXT>
__device__ __forceinline__ void dz_dmaop_fragment(const float* stage, int blk, int ch, int half, int l31,
                                                  float (&f)[8]) {
  if constexpr (KC) {
    const int row = blk * 32 + l31, fr = S::swz(row), u0 = ch * 4 + half * 2;
    const float4 v0 = *(const float4*)(stage + row * S::BK + 4 * (u0 ^ fr));
    const float4 v1 = *(const float4*)(stage + row * S::BK + 4 * ((u0 + 1) ^ fr));
    f[0] = v0.x; f[1] = v0.y; f[2] = v0.z; f[3] = v0.w; f[4] = v1.x; f[5] = v1.y; f[6] = v1.z; f[7] = v1.w;
  } else {
    const float* bl = stage + (ch * 16 + half * 8) * EXT + ((EXT >= 64 ? (blk ^ half) : blk) * 32 + l31);
#pragma unroll
    for (int s = 0; s < 8; ++s) f[s] = bl[s * EXT];
  }
}

template <class Op>
__device__ __forceinline__ void dz_dmaop_body(const typename Op::Params& p, const dim3& bid, float* lds) {
  using S = DzDmaOpShape<Op>;
  typename Op::Tile t;
  if (!Op::tile(p, bid, t)) return;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wk = wave / (S::SUBM * S::SUBN), sub = wave % (S::SUBM * S::SUBN);
  const int wm = sub / S::SUBN, wn = sub % S::SUBN;
  const int half = lane >> 5, l31 = lane & 31;
  const unsigned lds0 = (unsigned)(uintptr_t)lds;
  constexpr unsigned kRpw = S::SPLIT ? 16 / S::WKD : 16;
  const unsigned my_rmask = S::SPLIT ? (((1u << kRpw) - 1u) << (wk * kRpw)) : 0xffffu;

  const typename Op::Pre pre = Op::prefetch(p, t, wm * S::MI, wn * S::NI, lane, my_rmask);

  auto issue = [&](int st) {
    const unsigned base = lds0 + 4u * (unsigned)((st % S::NBUF) * S::STAGE);
#pragma unroll
    for (int i = 0; i < S::A_PER_WAVE; ++i) {
      const int idx = wave * S::A_PER_WAVE + i;
      int r, u;
      if constexpr (S::A_U8) { constexpr int UW = S::BM / 16; r = idx * (64 / UW) + lane / UW; u = lane % UW; }
      else dz_dmaop_granule<S, Op::A_KC, S::BM>(idx, lane, r, u);
      dz_glds16<0>(Op::a_src(p, t, st, r, u), base + 4u * ((unsigned)idx * 256u));
    }
#pragma unroll
    for (int i = 0; i < S::B_PER_WAVE; ++i) {
      const int idx = wave * S::B_PER_WAVE + i;
      int r, u;
      dz_dmaop_granule<S, Op::B_KC, S::BN>(idx, lane, r, u);
      dz_glds16<0>(Op::b_src(p, t, st, r, u), base + 4u * ((unsigned)S::A_FLOATS + (unsigned)idx * 256u));
    }
  };

  f32x16 acc[S::MI][S::NI];
#pragma unroll
  for (int mi = 0; mi < S::MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < S::NI; ++ni)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[mi][ni][i] = 0.f;

  const int nst = t.nst;
#pragma unroll
  for (int s = 0; s < S::NBUF - 1; ++s)
    if (s < nst) issue(s);
#pragma unroll 1
  for (int st = 0; st < nst; ++st) {
    const int ahead = min(S::NBUF - 2, nst - 1 - st);
    if (ahead >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * S::PER_STAGE) : "memory");
    else if (ahead == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(S::PER_STAGE) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (st + S::NBUF - 1 < nst) issue(st + S::NBUF - 1);
    const float* As = lds + (st % S::NBUF) * S::STAGE;
    const float* Bs = As + S::A_FLOATS;
#pragma unroll
    for (int kt = 0; kt < S::KT; ++kt) {
      const int ch = wk * S::KT + kt;
      float fa[S::MI][8], fb[S::NI][8];
#pragma unroll
      for (int mi = 0; mi < S::MI; ++mi) {
        if constexpr (S::A_U8) {
          const unsigned char* ab = (const unsigned char*)As + (ch * 16 + half * 8) * S::BM + (wm * S::MI + mi) * 32 + l31;
#pragma unroll
          for (int s = 0; s < 8; ++s) fa[mi][s] = dz_div255((float)ab[s * S::BM]);
        } else {
          dz_dmaop_fragment<S, Op::A_KC, S::BM>(As, wm * S::MI + mi, ch, half, l31, fa[mi]);
        }
      }
#pragma unroll
      for (int ni = 0; ni < S::NI; ++ni) dz_dmaop_fragment<S, Op::B_KC, S::BN>(Bs, wn * S::NI + ni, ch, half, l31, fb[ni]);
#pragma unroll
      for (int s = 0; s < 8; ++s)
#pragma unroll
        for (int mi = 0; mi < S::MI; ++mi)
#pragma unroll
          for (int ni = 0; ni < S::NI; ++ni)
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[mi][s], fb[ni][s], acc[mi][ni], 0, 0, 0);
    }
  }
  if (nst == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (the prefetch, if any)

  // ---- depth groups meet in LDS ----------------------------------------------------------
  if constexpr (S::WKD == 1) {
#pragma unroll
    for (int mi = 0; mi < S::MI; ++mi)
#pragma unroll
      for (int ni = 0; ni < S::NI; ++ni)
        Op::store(p, t, wm * S::MI + mi, wn * S::NI + ni, lane, acc[mi][ni], 0xffffu, pre);
  } else if constexpr (S::SPLIT) {
    __syncthreads();
    {
      float* dst = lds + ((wk * S::SUBM * S::SUBN + sub) * 16) * 64 + lane;
#pragma unroll
      for (int i = 0; i < 16; ++i) dst[i * 64] = acc[0][0][i];
    }
    __syncthreads();
    f32x16 out;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      out[i] = 0.f;
      if (i / (int)kRpw == wk) {   // wave-uniform
        const float* src = lds + (sub * 16 + i) * 64 + lane;
        float v = src[0];
#pragma unroll
        for (int k2 = 1; k2 < S::WKD; ++k2) v += src[k2 * S::SUBM * S::SUBN * 1024];
        out[i] = v;
      }
    }
    Op::store(p, t, wm, wn, lane, out, my_rmask, pre);
  } else {
    __syncthreads();
    constexpr int PER = S::SUBM * S::SUBN * S::MI * S::NI;
    if (wk > 0) {
#pragma unroll
      for (int mi = 0; mi < S::MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < S::NI; ++ni) {
          float* dst = lds + (((wk - 1) * PER + (sub * S::MI + mi) * S::NI + ni) * 16) * 64 + lane;
#pragma unroll
          for (int i = 0; i < 16; ++i) dst[i * 64] = acc[mi][ni][i];
        }
    }
    __syncthreads();
    if (wk > 0) return;
#pragma unroll
    for (int mi = 0; mi < S::MI; ++mi)
#pragma unroll
      for (int ni = 0; ni < S::NI; ++ni) {
#pragma unroll
        for (int k2 = 1; k2 < S::WKD; ++k2) {
          const float* src = lds + (((k2 - 1) * PER + (sub * S::MI + mi) * S::NI + ni) * 16) * 64 + lane;
#pragma unroll
          for (int i = 0; i < 16; ++i) acc[mi][ni][i] += src[i * 64];
        }
        Op::store(p, t, wm * S::MI + mi, wn * S::NI + ni, lane, acc[mi][ni], 0xffffu, pre);
      }
  }
}

template <class A, class B> struct DzDmaMax {
  static constexpr int v = DzDmaOpShape<A>::LDS_FLOATS > DzDmaOpShape<B>::LDS_FLOATS ? DzDmaOpShape<A>::LDS_FLOATS
                                                                                       : DzDmaOpShape<B>::LDS_FLOATS;
};

// One Op; two Ops (horizontal fusion: blocks [0, na) run OpA, the rest OpB); ... plus a side job
// behind them (dz_gemm.h's Side interface: run(sp, block, smem, bytes)).
template <class Op, int OCC>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(OCC, OCC)))
void dz_dmaop_kernel(typename Op::Params p, dim3 g) {
  __shared__ __attribute__((aligned(1024))) float lds[DzDmaOpShape<Op>::LDS_FLOATS];
  dz_dmaop_body<Op>(p, dz_unflatten(blockIdx.x, g), lds);
}
template <class Op, int OCC>
static inline int dz_launch_dmaop(const typename Op::Params& p, dim3 g, hipStream_t s) {
  hipLaunchKernelGGL((dz_dmaop_kernel<Op, OCC>), dim3(dz_count(g)), dim3(256), 0, s, p, g);
  DZ_LAUNCH_CHECK();
  return DZ_OK;
}
template <class OpA, class OpB, int OCC>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(OCC, OCC)))
void dz_dmaop2_kernel(typename OpA::Params pa, dim3 ga, typename OpB::Params pb, dim3 gb) {
  __shared__ __attribute__((aligned(1024))) float lds[DzDmaMax<OpA, OpB>::v];
  const unsigned na = ga.x * ga.y * ga.z;
  if (blockIdx.x < na) dz_dmaop_body<OpA>(pa, dz_unflatten(blockIdx.x, ga), lds);
  else dz_dmaop_body<OpB>(pb, dz_unflatten(blockIdx.x - na, gb), lds);
}
template <class OpA, class OpB, int OCC>
static inline int dz_launch_dmaop2(const typename OpA::Params& pa, dim3 ga, const typename OpB::Params& pb,
                                   dim3 gb, hipStream_t s) {
  hipLaunchKernelGGL((dz_dmaop2_kernel<OpA, OpB, OCC>), dim3(dz_count(ga) + dz_count(gb)), dim3(256), 0, s,
                     pa, ga, pb, gb);
  DZ_LAUNCH_CHECK();
  return DZ_OK;
}
template <class OpA, class OpB, class Side, int OCC>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(OCC, OCC)))
void dz_dmaop2_side_kernel(typename OpA::Params pa, dim3 ga, typename OpB::Params pb, dim3 gb,
                           typename Side::Params sp) {
  __shared__ __attribute__((aligned(1024))) float lds[DzDmaMax<OpA, OpB>::v];
  const unsigned na = ga.x * ga.y * ga.z, nb = gb.x * gb.y * gb.z;
  if (blockIdx.x < na) dz_dmaop_body<OpA>(pa, dz_unflatten(blockIdx.x, ga), lds);
  else if (blockIdx.x < na + nb) dz_dmaop_body<OpB>(pb, dz_unflatten(blockIdx.x - na, gb), lds);
  else Side::run(sp, blockIdx.x - na - nb, lds, (int)sizeof(lds));
}
template <class OpA, class OpB, class Side, int OCC>
static inline int dz_launch_dmaop2_side(const typename OpA::Params& pa, dim3 ga, const typename OpB::Params& pb,
                                        dim3 gb, const typename Side::Params& sp, unsigned side_blocks,
                                        hipStream_t s) {
  hipLaunchKernelGGL((dz_dmaop2_side_kernel<OpA, OpB, Side, OCC>),
                     dim3(dz_count(ga) + dz_count(gb) + side_blocks), dim3(256), 0, s, pa, ga, pb, gb, sp);
  DZ_LAUNCH_CHECK();
  return DZ_OK;
}

}  // namespace
