// Input gradient of the wide noisy layer (fc1: 3136 x [adv1 | val1] = 3136 x 1024, batch
// <= 32) on the matrix pipe:
//     dX[b][k] = relu'(x[b][k]) * sum_n dY[b][n] * W_eff[k][n],
//     W_eff[k][n] = Wmu[k][n] + Wsig[k][n] * (eps_in_h(n)[k] * eps_out[n])     (networks.py:168-176)
// The reduction index n is the contiguous one in memory, so W cannot be an MFMA operand
// straight from global memory (a lane would have to own a ROW): dz_row_dgrad.h therefore
// kept the layer on the vector ALU -- 64 packed multiply-adds per row and wave plus ~100
// instructions of cross-lane reduction, 4.1 us of VALU time per SIMD, the bound of that
// kernel whatever its launch shape (round-4 sweep: 224...640 workgroups all 12.3-13.9 us).
// Here the weights make the transposing trip through LDS, and they make it without
// touching a register: LDS-DMA (global_load_lds_dwordx4, 1 KB per wave-instruction).
//
//   * workgroup = 16 weight rows k0..k0+15 (K / 16 = 196 workgroups: one round on 256 CUs),
//     wave w = the 256 columns [256 w, 256 w + 256) of all 16 rows, i.e. 16 chunks of
//     16 rows x 16 columns for each of Wmu, Wsig (32 KB per wave, wave-private: no barrier
//     between the copy and its use);
//   * one DMA instruction = one chunk: lane L fetches the 16 bytes  W[k0 + (L >> 2)]
//     [16 t + 4 u .. +3],  u = (L & 3) ^ (L >> 4): the 16-byte units of a row are stored
//     XOR-swizzled by the row's upper two bits, so that the fragment read below -- lane
//     (j, q) reads unit q of row j as ONE ds_read_b128 -- touches 16 distinct 16-byte slots
//     per 16-lane group (LDS-DMA writes base + lane * 16 linearly: the swizzle goes on the
//     SOURCE address, cdna_hip_programming.md 5.4 rule 21);
//   * v_mfma_f32_16x16x4_f32, rows i = batch (two tiles of 16), columns j = weight rows,
//     depth = 4 columns per instruction: k-slot q of step (t, e) is column 16 t + 4 q + e,
//     so a lane's A operands are the float4s it loaded from dY ([b][16 t + 4 q ..+3], 32
//     coalesced 16-byte loads per lane, 128 VGPRs) and its B operands the four components of
//     the unit it read from LDS, turned into W_eff by one multiply and one FMA (the same
//     fma(sig, eps_in * eps_out, mu) as the forward kernel);
//   * 128 MFMAs of 32 cycles per wave (1.7 us), then the four waves' 16 x 16 x 2 partial
//     tiles are added through LDS in wave order, masked with relu'(x) and stored.
// Side blocks (Gram norms of dY, priority write-back) ride in front exactly as in
// fc1_dgrad_rows_kernel.  ref: rainbow/agent.py:112-118 (jax.grad through the network).
#pragma once
#include "dz_qnet_ops.h"
#include "dz_glds.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct Fc1DgradMfma {
  const float* params; const float* noise;
  long w_mu, w_sig; int ldw;          // fused [K][ldw] matrices, 1024 columns used
  int eps_in[2];                      // per head (columns < 512 / >= 512), indexed by row k
  int eps_out;                        // indexed by column 0..1023
  const float* dy; int ldy;           // dY [M][ldy] (1024 columns)
  const float* mask;                  // x [M][ldo] (post-ReLU): relu'(x) = x > 0
  float* out; int ldo;                // dX [M][ldo]
  int M, K;                           // batch rows (<= 32), weight rows (multiple of 16)
};
constexpr int kDmWaveFloats = 2 * 16 * 256 + 256;        // mu chunks, sigma chunks, eps_out
constexpr int kDmLdsFloats = 4 * kDmWaveFloats;          // 135 KB: one workgroup per CU

// the 8 DMA instructions of super-chunk `sc`: one instruction = 4 rows x 64 columns (256
// contiguous bytes per row); lane L = (row r = L >> 4 of the group, position L & 15) fetches
// unit (L & 15) ^ row: the 16-byte units of a row are stored XOR-swizzled by the row index
// (source-side swizzle: the DMA writes LDS linearly)
__device__ __forceinline__ void dm_issue(const Fc1DgradMfma& q, int k0, int col0, int lane, unsigned b_w,
                                         int sc) {
  const int r = lane >> 4, pos = lane & 15;
#pragma unroll
  for (int mat = 0; mat < 2; ++mat) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int row = 4 * g + r;
      const float* src = q.params + (mat ? q.w_sig : q.w_mu) + (long)(k0 + row) * q.ldw + col0 +
                         64 * sc + 4 * (pos ^ row);
      dz_glds16<0>(src, b_w + 4u * (unsigned)(sc * 2048 + mat * 1024 + g * 256));
    }
  }
}

template <int SC>
__device__ __forceinline__ void dm_super_chunk(const Fc1DgradMfma& q, int k0, int col0, int lane,
                                               unsigned b_w, const float* l_w, const float* l_eo, int j,
                                               int kq, float ein, const float4 (&D0)[16],
                                               const float4 (&D1)[16], f32x4& acc0, f32x4& acc1) {
  // interleaved issue: super-chunks 0 and 1 are in flight when the loop starts and SC + 2 is
  // requested behind SC's MFMAs -- 32 instructions issued back to back take 2.4 us (the issue
  // blocks on the memory pipeline's queues) during which the matrix pipe would idle.  So at most
  // the 8 instructions of SC + 1 are younger than the ones waited for here.
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(SC < 3 ? 8 : 0) : "memory");
  const float* rmu = l_w + SC * 2048 + (j >> 2) * 256 + (j & 3) * 64;   // row j of the mu block
  const float* rsg = rmu + 1024;
#pragma unroll
  for (int tc = 0; tc < 4; ++tc) {
    constexpr int dummy = 0; (void)dummy;
    const int T = 4 * SC + tc;
    const int u = ((4 * tc + kq) ^ j) & 15;             // where unit 4 tc + kq of row j was stored
    const float4 m4 = *(const float4*)(rmu + 4 * u);
    const float4 s4 = *(const float4*)(rsg + 4 * u);
    const float4 e4 = *(const float4*)(l_eo + 16 * T + 4 * kq);
    const float mm[4] = {m4.x, m4.y, m4.z, m4.w}, ss[4] = {s4.x, s4.y, s4.z, s4.w};
    const float ee[4] = {e4.x, e4.y, e4.z, e4.w};
    const float a0[4] = {D0[T].x, D0[T].y, D0[T].z, D0[T].w}, a1[4] = {D1[T].x, D1[T].y, D1[T].z, D1[T].w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float w = __builtin_fmaf(ss[e], ein * ee[e], mm[e]);
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[e], w, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[e], w, acc1, 0, 0, 0);
    }
  }
  __builtin_amdgcn_sched_barrier(0);   // (the next super-chunk's wait stays behind these MFMAs)
  if constexpr (SC + 2 < 4) {
    dm_issue(q, k0, col0, lane, b_w, SC + 2);
    __builtin_amdgcn_sched_barrier(0);
  }
  if constexpr (SC + 1 < 4)
    dm_super_chunk<SC + 1>(q, k0, col0, lane, b_w, l_w, l_eo, j, kq, ein, D0, D1, acc0, acc1);
}

__device__ __forceinline__ void fc1_dgrad_mfma_block(const Fc1DgradMfma& q, unsigned blk, float* lds) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int k0 = (int)blk * 16;
  const int j = lane & 15, kq = lane >> 4;              // fragment coordinates: row / k-slot
  float* wl = lds + wave * kDmWaveFloats;               // this wave's private block
  float* l_w = wl, *l_eo = wl + 2 * 16 * 256;
  const int col0 = 256 * wave;
  // ---- 1. dY: whole 1 KB rows per instruction (a fragment-shaped load touches 16 rows x 64
  // bytes: twice the cache-line requests for the same bytes), staged in the weight block
  // [b][256 + 4], read back as A operands -- 128 VGPRs -- before the DMA stream reuses the
  // block; the noise; the ReLU mask of this thread's outputs ----
  {
    // (every workgroup reads the same 128 KB: each starts at its own row, so that the 196 of
    // them do not walk the L2 channels in lockstep)
    float4 st[32];
    const int rot = (int)(blk & 31);
#pragma unroll
    for (int b = 0; b < 32; ++b)
      st[b] = dz_ld4(q.dy + (long)min((b + rot) & 31, q.M - 1) * q.ldy + col0 + 4 * lane);
#pragma unroll
    for (int b = 0; b < 32; ++b) {
      const int bb = (b + rot) & 31;
      *(float4*)(wl + bb * 260 + 4 * lane) = bb < q.M ? st[b] : dz_f4zero();
    }
  }
  const float4 eo_mine = dz_ld4(q.noise + q.eps_out + col0 + 4 * lane);   // this wave's 256 columns
  const float ein = q.noise[(wave >= 2 ? q.eps_in[1] : q.eps_in[0]) + k0 + j];
  // epilogue coordinates: thread (lane, r = wave) finishes acc register r of both tiles
  const int ob0 = 4 * kq + wave, ob1 = 16 + ob0;
  const float mk0 = q.mask[(long)min(ob0, q.M - 1) * q.ldo + k0 + j];
  const float mk1 = q.mask[(long)min(ob1, q.M - 1) * q.ldo + k0 + j];
  float4 D0[16], D1[16];
#pragma unroll
  for (int t = 0; t < 16; ++t) {
    D0[t] = *(const float4*)(wl + j * 260 + 16 * t + 4 * kq);
    D1[t] = *(const float4*)(wl + (16 + j) * 260 + 16 * t + 4 * kq);
  }
  // Everything above has to be IN its registers before the first DMA instruction: the asm
  // DMA is invisible to hipcc's vmcnt bookkeeping, and the block it writes is the staging block
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#pragma unroll
  for (int t = 0; t < 16; ++t) asm volatile("" : "+v"(D0[t].x), "+v"(D0[t].y), "+v"(D0[t].z), "+v"(D0[t].w),
                                                "+v"(D1[t].x), "+v"(D1[t].y), "+v"(D1[t].z), "+v"(D1[t].w));
  float ein_p = ein, mk0_p = mk0, mk1_p = mk1;
  float4 eo_p = eo_mine;
  asm volatile("" : "+v"(ein_p), "+v"(mk0_p), "+v"(mk1_p), "+v"(eo_p.x), "+v"(eo_p.y), "+v"(eo_p.z), "+v"(eo_p.w));
  __builtin_amdgcn_sched_barrier(0);
  // ---- 2. the weights: 32 LDS-DMA instructions per wave (dm_issue) ----
  const unsigned b_w = __builtin_amdgcn_readfirstlane((unsigned)(size_t)l_w);
#pragma unroll
  for (int sc = 0; sc < 2; ++sc) dm_issue(q, k0, col0, lane, b_w, sc);
  __builtin_amdgcn_sched_barrier(0);
  *(float4*)(l_eo + 4 * lane) = eo_p;
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  // ---- 3. 4 super-chunks x 4 chunks x 4 steps x 2 batch tiles, each super-chunk as soon as
  // it has landed ----
  f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
  dm_super_chunk<0>(q, k0, col0, lane, b_w, l_w, l_eo, j, kq, ein_p, D0, D1, acc0, acc1);
  // ---- 4. the four column quarters, added in wave order; mask; store ----
  __syncthreads();                                       // every wave is done with its weights
  float* red = lds;                                      // [wave][tile][reg][lane]
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    red[((wave * 2 + 0) * 4 + r) * 64 + lane] = acc0[r];
    red[((wave * 2 + 1) * 4 + r) * 64 + lane] = acc1[r];
  }
  __syncthreads();
  // acc[r] of lane (j, kq) is batch row 4 kq + r (within the tile), weight row j
  {
    const int r = wave;
    float v0 = red[((0 * 2 + 0) * 4 + r) * 64 + lane], v1 = red[((0 * 2 + 1) * 4 + r) * 64 + lane];
#pragma unroll
    for (int w = 1; w < 4; ++w) {
      v0 += red[((w * 2 + 0) * 4 + r) * 64 + lane];
      v1 += red[((w * 2 + 1) * 4 + r) * 64 + lane];
    }
    if (ob0 < q.M) q.out[(long)ob0 * q.ldo + k0 + j] = mk0_p > 0.f ? v0 : 0.f;
    if (ob1 < q.M) q.out[(long)ob1 * q.ldo + k0 + j] = mk1_p > 0.f ? v1 : 0.f;
  }
}

}  // namespace
