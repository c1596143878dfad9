// Rainbow's head chain as ONE multi-role launch (round 5).
// (ref: networks.py:239-258 the two noisy layers of the dueling head on top of the torso features,
//  rainbow/agent.py:97-109 the categorical double-Q loss, :112-118 its gradient.)
//
// Between fc1's weight stream and fc1's input gradient the learner step used to enqueue four
// launches with almost no arithmetic -- fold of fc1's 32 split-K slabs + bias + ReLU (4.8 us), the
// second noisy layer as a split-K GEMM (5.7), the loss kernel (9.0), the second layer's backward
// (6.8): 26.2 us, of which three kernel boundaries (2.4 us each) and three first trips to memory
// for data the previous kernel had just written (1.5-2 us each) are more than half.  Here the four
// are workgroup ROLES of one launch that hand their results to each other through seams whose data
// is its own flag (dz_seam.h), exactly as the actor's decision kernel does (dz_act_one.h):
//
//   A  fold     (4 G B blocks)   h1[r][c] = relu(sum of fc1's 32 slabs + b_mu + b_sig eps_out): one row
//                                x 256 columns per workgroup, fc_epilogue_kernel's summation order
//   B  fc2      (12 G (T0+T1))   one (64-column tile, apply, 128-deep K stage) per workgroup: W_eff
//                                = W_mu + W_sig (eps_in (x) eps_out) goes lane-wise from memory straight
//                                into MFMA B-operand registers BEFORE h1 exists; then h1 (A operand,
//                                registers as well: no LDS tile) as a seam; v_mfma_f32_32x32x2_f32 in
//                                the GEMM skeleton's k-slot order; partial slabs as a seam
//   C  loss     (B blocks)       rainbow_head_loss_block<1, 1>: folds the four fc2 slabs of its
//                                sample's three rows, dueling / softmax / double-Q argmax / Cramer
//                                projection / cross-entropy; dlogits as a seam
//   D1 dh1      (128 blocks)     row-owning stream over both heads' second-layer matrices
//                                (dz_row_dgrad.h, SEAM): weights of the first rows requested at launch
//   D2 dW2      (gw blocks)      fc2 weight gradient, operands lane-wise into registers, FcWgradOp's
//                                epilogue (gradient + sigma gradient + norm slots)
//   G  Gram     (84 blocks)      the wide layer's input Grams (dz_gram.h): independent, last in the grid
//
// Every role's arithmetic is the arithmetic of the kernel it replaces, in the same order on the
// same instructions: the results are BIT-IDENTICAL to the four launches (up to the sign of a zero:
// seams carry -0.0f for 0), tests/test_head_chain_gpu.py.
//
// The seam buffers are the step's ordinary workspace regions (ws_h1, the first four slabs of
// ws_fc2_part, ws_dout2); they are returned to all-zero bits by side blocks of the step's FIRST
// launch (conv1 forward: SeamClear), three launches before this one -- so a step that was
// abandoned half-way cannot leave stale "already written" words behind.
//
// LIVENESS: dependencies point from lower to higher block ids only (A < B < C < D), so the launch
// makes progress under the in-order dispatch of CDNA hardware whatever else occupies the chip
// (dz_act_one.h has the argument); every spin is bounded, a timeout sets the sticky word
// ws_scalars[DZ_SC_CHAIN_FAIL], turns the step's losses (and with them the priorities the replay
// validates) into NaN, and raises DZ_ST_CHAIN_TIMEOUT in the replay's status word if the step
// carries one.
#pragma once

#include "dz_qnet_kernels.h"
#include "dz_row_dgrad.h"
#include "dz_seam.h"

namespace {

// (tools builds, -DDZ_HC_STAMPS) per-workgroup wall-clock stamps in the idle dfeat slab buffer
#ifdef DZ_HC_STAMPS
#define HC_STAMP(p, i) do { if ((p).dbg && threadIdx.x == 0) (p).dbg[blockIdx.x * 8 + (i)] = (long long)wall_clock64(); } while (0)
#else
#define HC_STAMP(p, i) do {} while (0)
#endif

#ifndef DZ_HC_NAP_D
#define DZ_HC_NAP_D 24    // the backward roles wait ~15 us for the loss role: poll every ~0.7 us
#endif
#ifndef DZ_HC_NAP_C
#define DZ_HC_NAP_C 8     // the loss role waits ~9 us for the fc2 role
#endif
constexpr int kHcStages = 4;            // fc2's K = 512 as four 128-deep stages (FcFwdOp<1,2,2,4,2>)
#ifndef DZ_HC_FOLD_COLS
#define DZ_HC_FOLD_COLS 1024
#endif
constexpr int kHcFoldCols = DZ_HC_FOLD_COLS;   // role A: columns per workgroup

struct HeadChain {
  // ---- role A: fc1 epilogue --------------------------------------------------------------------
  const float* fc1_part;                // [32][rows][1024]
  int rows, B, G;                       // rows = G * B
  const float* prm[3]; const float* nz[3];
  long fc1_mu_b, fc1_sig_b; int n_fc1_out;
  float* h1;                            // [rows][1024]   SEAM
  // ---- role B: noisy fc2 -----------------------------------------------------------------------
  FcHead fc2h[2];
  int tiles0, tiles1;                   // 64-column tiles of the advantage / value head
  float* fc2_part;                      // [kHcStages][rows][ld2]   SEAM
  int ld2;
  // ---- role C ------------------------------------------------------------------------------------
  HeadLossArgs loss;
  // ---- role D ------------------------------------------------------------------------------------
  RowDgrad rd;                          // dy = loss.dout2 (SEAM), mask = h1 (SEAM)
  FcWgradParams wg; dim3 gw;
  GramX gram;
  // ---- failure -----------------------------------------------------------------------------------
  unsigned* fail;                       // ws_scalars[DZ_SC_CHAIN_FAIL], sticky
  uint32_t* status;                     // the replay's pinned status word (nullable)
  int limit;
  unsigned nA, nB, nC, nD1, nD2;        // role sizes (blocks)
  long long* dbg = nullptr;
};

__device__ __forceinline__ void hc_fail(const HeadChain& p) {
  if (threadIdx.x == 0) {
    __hip_atomic_store(p.fail, 1u, DZ_ACT_RLX);
    if (p.status)
      __hip_atomic_fetch_or(p.status, (uint32_t)DZ_ST_CHAIN_TIMEOUT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  if ((int)threadIdx.x < p.B) p.loss.losses[threadIdx.x] = __builtin_nanf("");
}

// ---- A: h1 = relu(fold of fc1's slabs + bias) ----------------------------------------------------
// fc_epilogue_kernel's arithmetic: wave w sums slabs w, w+4, .. in that order from 0.f, the four
// wave sums combine as (r0 + r1) + (r2 + r3), then + b_mu, then + b_sig * eps_out, ReLU.
__device__ __forceinline__ void hc_fold_block(const HeadChain& p, unsigned u, float* lds) {
  constexpr int NC = 1024 / kHcFoldCols, NP = kHcFoldCols / 256;   // passes of 256 columns
  static_assert(kHcFoldCols % 256 == 0 && 1024 % kHcFoldCols == 0, "fold tile");
  const int r = (int)u / NC, cq = (int)u % NC;
  const int tid = threadIdx.x, l = tid & 63, w = tid >> 6;
  const int g = r / p.B;
  HC_STAMP(p, 0);
  const float* prm = dz_pick3(p.prm, g);
  const float* nz = dz_pick3(p.nz, g);
  const int col0 = cq * kHcFoldCols + 4 * l;           // + 256 pass: wave 0's lanes finish four columns each
  const float* src = p.fc1_part + (long)r * 1024 + col0;
  const long stride = (long)p.rows * 1024;
  constexpr int SPW = kFc1Splits / 4;   // slabs per wave
  float4 x[NP][SPW];
#pragma unroll
  for (int ps = 0; ps < NP; ++ps)
#pragma unroll
    for (int j = 0; j < SPW; ++j) x[ps][j] = dz_ld4(src + 256 * ps + (long)(w + 4 * j) * stride);
  float4 bm[NP], bs[NP], be[NP];
#pragma unroll
  for (int ps = 0; ps < NP; ++ps) {
    bm[ps] = dz_ld4(prm + p.fc1_mu_b + col0 + 256 * ps); bs[ps] = dz_ld4(prm + p.fc1_sig_b + col0 + 256 * ps);
    be[ps] = dz_ld4(nz + p.n_fc1_out + col0 + 256 * ps);
  }
#pragma unroll
  for (int ps = 0; ps < NP; ++ps) {
    float4 v = dz_f4zero();
#pragma unroll
    for (int j = 0; j < SPW; ++j) { v.x += x[ps][j].x; v.y += x[ps][j].y; v.z += x[ps][j].z; v.w += x[ps][j].w; }
    *(float4*)(lds + w * kHcFoldCols + 256 * ps + 4 * l) = v;
  }
  __syncthreads();
  HC_STAMP(p, 1);
  // the four waves finish the passes round-robin (NP = 1: wave 0 alone)
  const __amdgpu_buffer_rsrc_t hr = act_rsrc(p.h1);
#pragma unroll
  for (int ps = 0; ps < NP; ++ps) {
    if ((ps & 3) != w) continue;   // (wave-uniform)
    const float* q0 = lds + 256 * ps + 4 * l;
    const float4 r0 = *(const float4*)q0, r1 = *(const float4*)(q0 + kHcFoldCols);
    const float4 r2 = *(const float4*)(q0 + 2 * kHcFoldCols), r3 = *(const float4*)(q0 + 3 * kHcFoldCols);
    float4 o;
    o.x = (r0.x + r1.x) + (r2.x + r3.x); o.y = (r0.y + r1.y) + (r2.y + r3.y);
    o.z = (r0.z + r1.z) + (r2.z + r3.z); o.w = (r0.w + r1.w) + (r2.w + r3.w);
    o.x += bm[ps].x; o.y += bm[ps].y; o.z += bm[ps].z; o.w += bm[ps].w;
    o.x += bs[ps].x * be[ps].x; o.y += bs[ps].y * be[ps].y; o.z += bs[ps].z * be[ps].z; o.w += bs[ps].w * be[ps].w;
    o.x = o.x > 0.f ? o.x : -0.f; o.y = o.y > 0.f ? o.y : -0.f;
    o.z = o.z > 0.f ? o.z : -0.f; o.w = o.w > 0.f ? o.w : -0.f;
    act_store4(hr, (unsigned)(r * 1024 + col0 + 256 * ps) * 4u, o);
  }
  HC_STAMP(p, 2);
}

// ---- B: one (column tile, apply, K stage) of the noisy second layer ------------------------------
// The arithmetic of dz_gemm_body<FcFwdOp<1, 2, 2, 4, 2>> for the tile (bid.x = column tile, split =
// stage): waves (wn, wk); wave (wn, wk) chains 32 MFMAs over chunks 4 wk .. 4 wk + 3 of the stage
// (lane half h takes k = 16 chunk + 8 h + s at step s), wk = 1 adds into wk = 0 through LDS.
__device__ __forceinline__ void hc_fc2_block(const HeadChain& p, unsigned u, float* lds) {
  const int st = (int)u % kHcStages;
  const int gt = (int)u / kHcStages;
  const int nt = p.tiles0 + p.tiles1;
  const int ct = gt % nt, g = gt / nt;
  const int hsel = ct >= p.tiles0 ? 1 : 0;
  const FcHead hd = dz_pick_head(p.fc2h, hsel);
  const int n0 = (ct - (hsel ? p.tiles0 : 0)) * 64;
  const float* prm = dz_pick3(p.prm, g);
  const float* nz = dz_pick3(p.nz, g);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wn = wave & 1, wk = wave >> 1, half = lane >> 5, l31 = lane & 31;
  HC_STAMP(p, 0);
  // B fragments: W_eff[k][ncol], everything requested before the first wait
  const int c = wn * 32 + l31;
  const int ncol = min(n0 + 4 * (c >> 2), hd.ldw - 4) + (c & 3);   // FcFwdOp::load_b's clamp
  const int kb = st * 128 + wk * 64 + half * 8;                    // + 16 kt + s
  float fb[4][8], sg[4][8], ei[4][8];
  {
    const float* wm = prm + hd.w_mu + (long)kb * hd.ldw + ncol;
    const float* ws = prm + hd.w_sig + (long)kb * hd.ldw + ncol;
    const float* ep = nz + hd.eps_in + kb;
#pragma unroll
    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        fb[kt][s] = wm[(long)(16 * kt + s) * hd.ldw];
        sg[kt][s] = ws[(long)(16 * kt + s) * hd.ldw];
        ei[kt][s] = ep[16 * kt + s];
      }
  }
  const float eo = nz[hd.eps_out + ncol];
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int kt = 0; kt < 4; ++kt)
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      fb[kt][s] = __builtin_fmaf(sg[kt][s], ei[kt][s] * eo, fb[kt][s]);   // networks.py:168-176
      asm volatile("" : "+v"(fb[kt][s]));   // formed HERE, before the wait for h1
    }
  HC_STAMP(p, 1);
  // A fragments: h1[g B + row][x_off + k], rows beyond the batch are zero
  const int row = min(l31, p.B - 1);
  const float rmask = l31 < p.B ? 1.f : 0.f;
  const __amdgpu_buffer_rsrc_t h1r = act_rsrc(p.h1);
  const unsigned xo = (unsigned)((g * p.B + row) * 1024 + hd.x_off + kb) * 4u;
  float fa[4][8];
  {
    // cheap rounds first (ONE load per thread): eight words of every row of the tile, one per
    // 16-byte store of the fold workgroup that produces the row -- so that the payload round
    // below (8 x 16 bytes per thread) normally runs once
    if (act_watch_each(p.h1 + (long)(g * p.B + min(tid >> 3, p.B - 1)) * 1024 + hd.x_off + st * 128 +
                           16 * (tid & 7) + 15, p.fail, p.limit)) { hc_fail(p); return; }
    HC_STAMP(p, 2);
    int round = 0;
    bool miss, give_up;
    do {
      // (all loads first, then the checks, branch-free: a short-circuit `||` behind each load is a
      // basic block with its own wait -- 16 serial round trips, 2.5 us per round)
#pragma unroll
      for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          const float4 t = act_load4(h1r, xo + (unsigned)(16 * kt + 4 * s2) * 4u);
          fa[kt][4 * s2] = t.x; fa[kt][4 * s2 + 1] = t.y; fa[kt][4 * s2 + 2] = t.z; fa[kt][4 * s2 + 3] = t.w;
        }
      unsigned all = 1u;
#pragma unroll
      for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int s = 0; s < 8; ++s) all &= __builtin_bit_cast(unsigned, fa[kt][s]) != 0u ? 1u : 0u;
      miss = all == 0u;
    } while (act_again(miss, round++, p.fail, &give_up, p.limit));
    if (give_up) { hc_fail(p); return; }
#ifdef DZ_HC_STAMPS
    if (p.dbg && threadIdx.x == 0) p.dbg[blockIdx.x * 8 + 5] = round;
#endif
  }
  HC_STAMP(p, 3);
  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
  for (int kt = 0; kt < 4; ++kt)
#pragma unroll
    for (int s = 0; s < 8; ++s)
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[kt][s] * rmask, fb[kt][s], acc, 0, 0, 0);
  // wk = 1 -> LDS -> wk = 0 (dz_gemm_body's WK > 1 epilogue without SPLIT_STORE)
  if (wk == 1) {
    float* dst = lds + (wn * 16) * 64 + lane;
#pragma unroll
    for (int i = 0; i < 16; ++i) dst[i * 64] = acc[i];
  }
  __syncthreads();
  // The tile goes out as 16-byte write-through stores (a dword `sc1` store costs ~6x per byte):
  // transposed through a second LDS block, two float4 per thread.  Columns of the head's pitch
  // beyond N are stored as (marked) zeros -- the one-launch-per-stage form leaves them at the
  // workspace's zero; the sum downstream is the same.
  float* T = lds + 2 * 16 * 64;          // [32][68]
  if (wk == 0) {
    const float* src = lds + (wn * 16) * 64 + lane;
    float* dst = T + wn * 32 + l31;
#pragma unroll
    for (int i = 0; i < 16; ++i) dst[dz_acc_row(i, lane) * 68] = acc[i] + src[i * 64];
  }
  __syncthreads();
  const __amdgpu_buffer_rsrc_t pr = act_rsrc(p.fc2_part);
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int idx = tid + 256 * j, m = idx >> 4, c4 = idx & 15;
    const int col = n0 + 4 * c4;
    if (m < p.B && col < hd.ldw) {
      float4 v = *(const float4*)(T + m * 68 + 4 * c4);
      v.x = col < hd.N ? v.x : 0.f; v.y = col + 1 < hd.N ? v.y : 0.f;
      v.z = col + 2 < hd.N ? v.z : 0.f; v.w = col + 3 < hd.N ? v.w : 0.f;
      act_store4(pr, (unsigned)(((st * p.rows + g * p.B + m) * p.ld2) + hd.out_off + col) * 4u, act_mark4(v));
    }
  }
  HC_STAMP(p, 4);
}

// ---- D2: fc2's weight gradient, one 64 x 64 tile of one head ---------------------------------------
// The arithmetic of dz_gemm_body<FcWgradOp<2, 2, 1, 2>> (one stage: the batch is the reduction): wave
// (wm, wn) chains 16 MFMAs over batch rows m = 16 kt + 8 h + s; epilogue = FcWgradOp::store.
__device__ __forceinline__ void hc_wgrad_block(const HeadChain& p, unsigned u) {
  using Op = FcWgradOp<2, 2, 1, 2>;
  Op::Tile t;
  if (!Op::tile(p.wg, dz_unflatten(u, p.gw), t)) return;
  const FcHead& hd = t.hd;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1, half = lane >> 5, l31 = lane & 31;
  const int M = p.wg.M;
  const int kx = t.m0 + wm * 32 + l31;                       // weight row (K = 512: no clamp needed)
  const int c = wn * 32 + l31, col = t.n0 + c;
  const int ncol = min(t.n0 + 4 * (c >> 2), hd.ldw - 4) + (c & 3);   // FcWgradOp::load_b's clamp
  const bool real = col < hd.N;                              // (then ncol == col)
  const float* xp = p.wg.x + hd.x_off + kx;
  const float* dp = p.wg.dy + hd.out_off + ncol;
  float fa[2][8], fb[2][8];
  HC_STAMP(p, 0);
  {
    // one word per producer first: every sample's loss workgroup stores its value-head
    // dlogits last
    if (act_watch_each<DZ_HC_NAP_D>(tid < 32 ? p.wg.dy + (long)min(tid, M - 1) * p.wg.ldy + p.loss.val_off + p.loss.K - 1
                                : nullptr, p.fail, p.limit)) { hc_fail(p); return; }
    int round = 0;
    bool miss, give_up;
    do {
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int s = 0; s < 8; ++s) {
          const int mc = min(16 * kt + 8 * half + s, M - 1);
          fa[kt][s] = act_load(xp + (long)mc * p.wg.ldx);
          fb[kt][s] = act_load(dp + (long)mc * p.wg.ldy);
        }
      unsigned all = 1u;
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int s = 0; s < 8; ++s) {
          all &= __builtin_bit_cast(unsigned, fa[kt][s]) != 0u ? 1u : 0u;
          all &= (!real || __builtin_bit_cast(unsigned, fb[kt][s]) != 0u) ? 1u : 0u;
        }
      miss = all == 0u;
    } while (act_again(miss, round++, p.fail, &give_up, p.limit));
    if (give_up) { hc_fail(p); return; }
  }
  HC_STAMP(p, 1);
#pragma unroll
  for (int kt = 0; kt < 2; ++kt)
#pragma unroll
    for (int s = 0; s < 8; ++s) {   // batch rows beyond M contribute zeros (FcWgradOp's loaders)
      const bool in = 16 * kt + 8 * half + s < M;
      fa[kt][s] = in ? fa[kt][s] : 0.f; fb[kt][s] = in ? fb[kt][s] : 0.f;
    }
  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
  for (int kt = 0; kt < 2; ++kt)
#pragma unroll
    for (int s = 0; s < 8; ++s)
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[kt][s], fb[kt][s], acc, 0, 0, 0);
  Op::store(p.wg, t, wm, wn, lane, acc);
  HC_STAMP(p, 2);
}

constexpr int kHcLdsFloats = kRdLdsFloats;   // the row-owning stream's butterfly block is the largest
static_assert(kHcLdsFloats >= 4 * kHcFoldCols && kHcLdsFloats >= 2 * 16 * 64 + 32 * 68, "fold / wk exchange + tile");

template <int NJ0>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2)))
void rainbow_head_chain_kernel(HeadChain p) {
  __shared__ __attribute__((aligned(32))) float lds[kHcLdsFloats];
  unsigned b = blockIdx.x;
  if (b < p.nA) { hc_fold_block(p, b, lds); return; }
  b -= p.nA;
  if (b < p.nB) { hc_fc2_block(p, b, lds); return; }
  b -= p.nB;
  if (b < p.nC) {
    HeadSeam seam; seam.fail = p.fail; seam.limit = p.limit;
    seam.stages = kHcStages; seam.tiles0 = p.tiles0; seam.tiles = p.tiles0 + p.tiles1; seam.groups = p.G;
#ifdef DZ_HC_STAMPS
    seam.dbg = p.dbg ? p.dbg + blockIdx.x * 8 : nullptr;
#endif
    HC_STAMP(p, 0);
    rainbow_head_loss_block<1, 1>(p.loss, (int)b, lds, seam);
    HC_STAMP(p, 3);
    return;
  }
  b -= p.nC;
  if (b < p.nD1) {
    HC_STAMP(p, 0);
    row_dgrad_block<NJ0, 1, false, (NJ0 >= 4 ? 1 : 4), true, true>(p.rd, b, lds);   // (five jobs: two rows in flight keep the launch at 256 VGPRs without scratch)
    HC_STAMP(p, 3);
    return;
  }
  b -= p.nD1;
  if (b < p.nD2) { hc_wgrad_block(p, b); return; }
  b -= p.nD2;
  HC_STAMP(p, 0);
  dz_gram_x_block(p.gram, b, (dz_d4*)lds);
  HC_STAMP(p, 1);
}

}  // namespace
