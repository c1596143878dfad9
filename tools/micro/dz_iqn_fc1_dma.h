// IQN fc1 forward (h1 = relu(head_in @ W1 + b1): 5 120 x 3 136 x 512 at the reference sizes,
// 16.4 GFLOP -- the one launch of the package that the fp32 matrix pipe bounds) with both
// operands going global memory -> LDS WITHOUT touching a register: LDS-DMA
// (global_load_lds_dwordx4, 1 KB per wave-instruction), three stage buffers, ONE barrier per
// stage.  The register-staged skeleton (dz_gemm.h) cannot keep its prefetch in flight here:
// hipcc's allocator parks freshly loaded registers around the MFMA block, and the copy waits
// for the load in front of the MFMAs (EXPERIMENTS.md, round 5).
//
//   * tile = 64 rows x 32 columns, 4 waves = 2 row halves (wm) x 2 depth halves (wk) of every
//     stage; stage = 32 KT deep (KT chunks of 16 per wave); the two depth halves are added
//     through LDS at the end (wave wk = 0 stores), exactly as the skeleton's WK = 2 form;
//   * A stage in LDS: [64 rows][32 KT floats], the 16-byte units of a row XOR-swizzled by the
//     row (f(row) below) so that the fragment read -- 16 consecutive rows, the same logical
//     unit, ONE ds_read_b128 each -- touches 16 distinct 16-byte slots.  The DMA writes LDS
//     linearly (base + lane * 16), so the swizzle goes on the SOURCE address;
//   * B stage in LDS: per 16-deep chunk eight 256-byte lines [row s | row s + 8] x 32 columns:
//     MFMA step s reads line s -- lanes 0-31 its first half, lanes 32-63 (k-slot 8 + s) its
//     second: all 64 banks, no conflict;
//   * k-slot permutation and chunk order as in dz_gemm.h (lane half h takes k = 8 h + s at step
//     s), so each output element's MFMA sequence is the skeleton's WK = 2 sequence.
// Whole tiles only (rows of every group % 64 == 0, K % (32 KT) == 0, N % 32 == 0); the masked
// GEMM form stays for every other shape.  Inline-assembly DMA with hand-counted s_waitcnt: no
// ordinary global load is in flight between the first DMA and the last wait (the bias is
// loaded behind it).
#pragma once

#include "dz_fc1_dgrad.h"   // dz_glds16
#include "dz_iqn_ops.h"

namespace {

struct IqnFc1DmaParams {
  const float* x;            // [rows][ldx]
  int ldx;
  int G;
  int row0[DZ_MAX_GROUPS];
  int tiles[DZ_MAX_GROUPS];  // 64-row tiles of the group
  const float* params[DZ_MAX_GROUPS];
  long w_off, b_off;
  int ldw, K, N;
  float* out; int ldo;
};

template <int KT, int NBUF_>
struct IqnFc1Dma {
  static constexpr int BK = 32 * KT;                 // floats of depth per stage
  static constexpr int UPR = BK / 4;                 // 16-byte units per A row
  static constexpr int A_FLOATS = 64 * BK;           // per stage
  static constexpr int B_FLOATS = BK * 32;
  static constexpr int STAGE = A_FLOATS + B_FLOATS;
  static constexpr int NBUF = NBUF_;
  static constexpr int LDS_FLOATS = NBUF * STAGE;
  static constexpr int A_INSTR = A_FLOATS / 256 / 4;   // DMA instructions per wave and stage (A)
  static constexpr int B_INSTR = B_FLOATS / 256 / 4;   //                                     (B)
  static constexpr int PER_STAGE = A_INSTR + B_INSTR;
  static_assert(KT == 1 || KT == 2, "32- or 64-deep stages");
  static_assert(2 * 16 * 64 <= LDS_FLOATS, "the depth halves' exchange fits in the stage buffers");
  __device__ static int f(int row) { return KT == 1 ? ((row >> 1) & 7) : (row & 15); }
};

template <int KT, int NBUF, int OCC>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(OCC, OCC)))
void iqn_fc1_fwd_dma_kernel(IqnFc1DmaParams p, dim3 g) {
  using C = IqnFc1Dma<KT, NBUF>;
  __shared__ __attribute__((aligned(1024))) float lds[C::LDS_FLOATS];
  dim3 bid;
  if (!dz_xcd_tile(blockIdx.x, g, bid)) return;
  const int grp = bid.z, rt = bid.y;
  if (grp >= p.G || rt >= dz_pick3(p.tiles, grp)) return;
  const float* prm = dz_pick3(p.params, grp);
  const int row0 = dz_pick3(p.row0, grp) + rt * 64, n0 = bid.x * 32;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // (wave-uniform: the DMA's LDS base is an SGPR)
  const int wm = wave & 1, wk = wave >> 1, half = lane >> 5, l31 = lane & 31;
  const unsigned lds0 = (unsigned)(uintptr_t)lds;     // LDS byte address of the buffer

  // ---- this wave's DMA sources (stage 0) and LDS destinations --------------------------------
  // A: instruction i of wave w covers LDS rows [R0, R0 + 256 / BK) of the stage, R0 = (w *
  // A_INSTR + i) * (256 / BK); lane L -> row R0 + L / UPR, physical unit L % UPR
  const float* asrc[C::A_INSTR];
  unsigned adst[C::A_INSTR];
#pragma unroll
  for (int i = 0; i < C::A_INSTR; ++i) {
    const int r = (wave * C::A_INSTR + i) * (256 / C::BK) + lane / C::UPR;
    const int u = (lane % C::UPR) ^ C::f(r);
    asrc[i] = p.x + (long)(row0 + r) * p.ldx + 4 * u;
    adst[i] = 4u * (unsigned)((wave * C::A_INSTR + i) * 256);
  }
  // B: instruction j of wave w is half (j & 1) of chunk (w * B_INSTR + j) / 2 ... one chunk =
  // two instructions; lane L -> line 4 (idx & 1) + L / 16, half (L >> 3) & 1, columns 4 (L & 7)
  const float* bsrc[C::B_INSTR];
  unsigned bdst[C::B_INSTR];
#pragma unroll
  for (int j = 0; j < C::B_INSTR; ++j) {
    const int idx = wave * C::B_INSTR + j, ch = idx >> 1;
    const int line = 4 * (idx & 1) + (lane >> 4);
    const int k = ch * 16 + line + 8 * ((lane >> 3) & 1);
    bsrc[j] = prm + p.w_off + (long)k * p.ldw + n0 + 4 * (lane & 7);
    bdst[j] = 4u * (unsigned)(C::A_FLOATS + idx * 256);
  }
  const long a_step = C::BK, b_step = (long)C::BK * p.ldw;
  auto issue = [&](int buf) {
    const unsigned base = lds0 + 4u * (unsigned)(buf * C::STAGE);
#pragma unroll
    for (int i = 0; i < C::A_INSTR; ++i) { dz_glds16<0>(asrc[i], base + adst[i]); asrc[i] += a_step; }
#pragma unroll
    for (int j = 0; j < C::B_INSTR; ++j) { dz_glds16<0>(bsrc[j], base + bdst[j]); bsrc[j] += b_step; }
  };

  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  const int nst = p.K / C::BK;
  // NBUF - 1 stages are in flight ahead of the one being consumed
  issue(0);
  if (C::NBUF > 2 && nst > 1) issue(1);
  const int arow = wm * 32 + l31;
  const int fr = C::f(arow);
  for (int st = 0; st < nst; ++st) {
    // stage st has landed (for THIS wave) when at most the younger stage's instructions are out
    if (C::NBUF > 2 && st + 1 < nst) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(C::PER_STAGE) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();   // ... for every wave; and every wave has finished reading stage st - 1
    if (st + C::NBUF - 1 < nst) issue((st + C::NBUF - 1) % C::NBUF);   // into the buffer stage st - 1 was read from
    const float* As = lds + (st % C::NBUF) * C::STAGE;
    const float* Bs = As + C::A_FLOATS;
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) {
      const int ch = wk * KT + kt;
      const int u0 = ch * 4 + half * 2;
      const float4 v0 = *(const float4*)(As + arow * C::BK + 4 * (u0 ^ fr));
      const float4 v1 = *(const float4*)(As + arow * C::BK + 4 * ((u0 + 1) ^ fr));
      const float fa[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
      const float* bsrc_l = Bs + ch * 512 + half * 32 + l31;
      float fb[8];
#pragma unroll
      for (int s = 0; s < 8; ++s) fb[s] = bsrc_l[s * 64];
#pragma unroll
      for (int s = 0; s < 8; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[s], fb[s], acc, 0, 0, 0);
    }
  }
  // ---- the two depth halves through LDS (skeleton's WK = 2 epilogue), bias, ReLU, store --------
  __syncthreads();
  float* red = lds;   // [wm][16][64]
  if (wk == 1) {
#pragma unroll
    for (int i = 0; i < 16; ++i) red[(wm * 16 + i) * 64 + lane] = acc[i];
  }
  __syncthreads();
  if (wk == 1) return;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] += red[(wm * 16 + i) * 64 + lane];
  const int col = n0 + l31;
  const float b = prm[p.b_off + col];
  float* o = p.out + (long)(row0 + wm * 32) * p.ldo + col;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const float v = acc[r] + b;
    o[(long)dz_acc_row(r, lane) * p.ldo] = v > 0.f ? v : 0.f;
  }
}

}  // namespace
