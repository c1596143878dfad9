// Rainbow's wide layer (fc1: 3136 -> 2 x 512, noisy): the weight gradient is never
// stored.  G = X^T D has rank <= 32 (the batch), its two factors are 400 KB and 128 KB,
// while the matrix itself is 12.8 MB for mu and as much again for sigma: writing it in
// the backward pass (12.8 MB; sigma was already derived on the fly) and reading it
// back in the optimiser (12.8 MB) cost more than forming each element again from the
// L2-resident factors at the moment the optimiser needs it.  Three pieces:
//   * dz_gram_x_block (dz_gram.h): Grams of the layer input, side job of the loss
//     kernel's launch;
//   * GramDSide (here): side job (16 blocks) of the input-gradient launch -- forms the
//     Grams of dh1 (left finished by fc2's backward launch: its input gradient is a
//     row-owning stream, dz_row_dgrad.h) and the layer's contribution to the global
//     gradient norm  <X X^T, D D^T>  (both parameter matrices, per head), one non-negative
//     float per block into the fused-norm slots;
//   * adam_onfly_kernel (here): the optimiser; a workgroup owns a 56 x 128 tile (rounds 3-5: 112 x 64)
//     of the mu AND sigma matrices, keeps the 32 x 128 strip of dh1 and the 56 x 32 tile of X in
//     LDS, and forms every gradient element (32 FMAs) right before its update.  The rest
//     of the parameter vector (convs, biases, fc2: 4 % of it) is streamed from the stored
//     gradient by 64 more blocks of the same launch.
// Step 166.5 -> 160 us with the input gradients as row-owning streams: the
// backward launch loses its 784 weight-gradient workgroups and its reduce launch (14.4 +
// 4.8 -> 12.4 us); the optimiser itself moves 166 MB instead of 192 MB in the same 32 us
// -- a bare float4 read-modify-write stream of that traffic takes 24.65 us
// (tools/micro/rmw2_micro.hip; round 3's 30.3 us came from a micro-benchmark the compiler had
// narrowed to dword accesses), so ~5 us of the role's arithmetic are exposed: EXPERIMENTS.md.
// ref: rainbow/agent.py:112-127 (grad + optimizer.update), networks.py:150-180 (the layer).
#pragma once
#include "dz_qnet_kernels.h"
namespace {
struct Fc1OnFly {
  const float* feat; const float* dh1; int B;
  unsigned mu_b, sig_b;   // BYTE offsets of the [kFlat][ld] matrices in the parameter vector (< 2^32)
  int ld;
  const float* eps_in0; const float* eps_in1; const float* eps_out;
};
struct AdamRanges { long lo[3]; long n[3]; };   // float4 ranges of the flat remainder

__device__ __forceinline__ float dz_sgpr(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v)));
}
// loads / stores at (uniform base) + (32-bit byte offset): SGPR-base addressing, one
// address register per stream pair instead of two per stream
__device__ __forceinline__ float4 ld_off(const float* base, unsigned off) {
  return *(const float4*)((const char*)base + off);
}
__device__ __forceinline__ void st_off(float* base, unsigned off, float4 v) {
  *(float4*)((char*)base + off) = v;
}

// One tile of the fc1 matrices: (256 / (C/4)) * IT rows x C columns of BOTH mu and sigma.
template <int C, int IT>
struct OfTile {
  static constexpr int TPR = C / 4, RP = 256 / TPR, R = RP * IT, FS = 36;
  static constexpr int kStrips = 1024 / C, kBlocks = (kFlat / R) * kStrips;
  static constexpr int GP = C + 4;                        // row pitch of the gradient tile (MFMA form)
  static constexpr int kFill = 32 * C + R * FS;           // dh1 strip + X tile (dead once the tile exists)
  static constexpr int kLds = (kFill > R * GP ? kFill : R * GP) + C + R;   // floats
  static_assert(kFlat % R == 0, "rows");
  static_assert(C + R <= 256, "one thread per eps_out column and eps_in row of the tile");
};
template <int C, int IT>
__device__ __forceinline__ void adam_fc1_block(unsigned blk, const Fc1OnFly& q,
                                               float* __restrict__ p, float* __restrict__ m,
                                               float* __restrict__ v, const float* part, int nparts,
                                               const int32_t* count, float lr, float b1, float b2,
                                               float eps, float max_norm, float* red, float* lds) {
  using T = OfTile<C, IT>;
  float* s_dh1 = lds; float* s_ft = lds + 32 * C;
  float* s_eo = lds + (T::kFill > T::R * T::GP ? T::kFill : T::R * T::GP);
  const int strip = blk % T::kStrips, rg = blk / T::kStrips;
  const int k0 = rg * T::R, c0 = strip * C;
  const int tid = threadIdx.x, rl = tid / T::TPR, c4 = tid % T::TPR;
  const int col = c0 + 4 * c4;
  const float* ein = c0 < 512 ? q.eps_in0 : q.eps_in1;
  // the norm partials are requested FIRST: their (cold) trip to memory then ends before the six
  // streams' first tiles arrive, instead of behind them (vector loads return in order; same
  // box 6 603-6 625 -> 6 645-6 665 steps/s)
  const AdamPartials parts = adam_partials_request(part, nparts);
#ifndef DZ_OF_EARLY
#define DZ_OF_EARLY 0
#endif
#ifndef DZ_OF_AHEAD   // the next rows of the six streams are requested BEFORE this iteration's arithmetic (round 6)
#define DZ_OF_AHEAD 1
#endif
  unsigned om = q.mu_b + ((unsigned)(k0 + rl) * (unsigned)q.ld + (unsigned)col) * 4u;
  unsigned os = q.sig_b + ((unsigned)(k0 + rl) * (unsigned)q.ld + (unsigned)col) * 4u;
  float4 pm, mm, vm, ps, ms, vs;
  {  // LDS fill: dh1 strip [32][C], X tile [R][32 (+4)], the strip's eps_out and the rows' eps_in.
     // DZ_OF_EARLY: the factors are REQUESTED, then the six streams' first rows, and only then are
     // the factors written to LDS (vmcnt counts in order: that wait leaves the streams in flight) --
     // otherwise the streams start one cold round trip late.
    float4 dreg[C / 32];
    float freg[(32 * T::R + 255) / 256];
#pragma unroll
    for (int i = 0; i < C / 32; ++i) {
      const int e = tid + 256 * i, b = e / T::TPR, cc = e % T::TPR;
      dreg[i] = *(const float4*)(q.dh1 + (unsigned)(min(b, q.B - 1) * 1024 + c0 + 4 * cc));
    }
#pragma unroll
    for (int i = 0; i < (32 * T::R + 255) / 256; ++i) {
      const int e = min(tid + 256 * i, 32 * T::R - 1), b = e / T::R, r = e % T::R;
      freg[i] = q.feat[(unsigned)(min(b, q.B - 1) * kFlat + k0 + r)];
    }
    const float ereg = tid < C ? q.eps_out[c0 + tid] : ein[k0 + min(tid - C, T::R - 1)];
#if DZ_OF_EARLY
    __builtin_amdgcn_sched_barrier(0);
    pm = ld_off(p, om); mm = ld_off(m, om); vm = ld_off(v, om);
    ps = ld_off(p, os); ms = ld_off(m, os); vs = ld_off(v, os);
    __builtin_amdgcn_sched_barrier(0);
#endif
#pragma unroll
    for (int i = 0; i < C / 32; ++i) {
      const int e = tid + 256 * i, b = e / T::TPR, cc = e % T::TPR;
      *(float4*)(s_dh1 + b * C + 4 * cc) = b < q.B ? dreg[i] : dz_f4zero();
    }
#pragma unroll
    for (int i = 0; i < (32 * T::R + 255) / 256; ++i) {
      const int e = tid + 256 * i, b = e / T::R, r = e % T::R;
      if (e < 32 * T::R) s_ft[r * T::FS + b] = b < q.B ? freg[i] : 0.f;
    }
    if (tid < C + T::R) s_eo[tid] = ereg;
  }
#if !DZ_OF_EARLY
  pm = ld_off(p, om); mm = ld_off(m, om); vm = ld_off(v, om);
  ps = ld_off(p, os); ms = ld_off(m, os); vs = ld_off(v, os);
#endif
  const AdamScalars sc0 = adam_scalars_from(parts, nparts, count, b1, b2, max_norm, red);  // (syncs)
  const float gn = dz_sgpr(sc0.gn), bc1 = dz_sgpr(sc0.bc1), bc2 = dz_sgpr(sc0.bc2);
  const bool pass = __builtin_amdgcn_readfirstlane((int)sc0.pass) != 0;
  const unsigned rstep = (unsigned)(T::RP * q.ld) * 4u;
#ifndef DZ_ADAM_MFMA
#define DZ_ADAM_MFMA 0
#endif
#if DZ_ADAM_MFMA
  // The gradient tile G[k][n] = sum_b x[b][k] dh1[b][n] on the matrix pipe (round 6): wave w
  // owns the 16-column block w of the strip (C = 64) and all R / 16 row blocks, depth 32 =
  // eight `v_mfma_f32_16x16x4_f32` per block with k-slot b = 4 s + (lane >> 4): the same
  // ascending-b fma chain per element as the VALU loop below.  The tile then replaces the
  // factors in LDS (row pitch C + 4) and every thread reads its four gradients per row as
  // ONE ds_read_b128 instead of 40 -- the VALU loop spent 128 FMAs and 640 bytes of LDS reads
  // per thread and row on it.
  static_assert(C == 64 && T::R % 16 == 0, "four column blocks of 16, whole row blocks");
  {
    typedef float f32x4m __attribute__((ext_vector_type(4)));
    constexpr int NRB = T::R / 16;
    const int lane = tid & 63, wv = tid >> 6, li = lane & 15, kq = lane >> 4;
    f32x4m acc[NRB];
#pragma unroll
    for (int rb = 0; rb < NRB; ++rb) acc[rb] = f32x4m{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int st = 0; st < 8; ++st) {
      const float bv = s_dh1[(4 * st + kq) * C + 16 * wv + li];
#pragma unroll
      for (int rb = 0; rb < NRB; ++rb)
        acc[rb] = __builtin_amdgcn_mfma_f32_16x16x4f32(s_ft[(16 * rb + li) * T::FS + 4 * st + kq], bv, acc[rb], 0, 0, 0);
    }
    __syncthreads();   // every wave has read the factors
#pragma unroll
    for (int rb = 0; rb < NRB; ++rb)
#pragma unroll
      for (int r = 0; r < 4; ++r) lds[(16 * rb + 4 * kq + r) * T::GP + 16 * wv + li] = acc[rb][r];
    __syncthreads();
  }
#endif
  // Measured alternatives (optimiser role alone; this form 31.4-31.9 us, a bare float4
  // read-modify-write stream of the same 165 MB 24.65 us): one pass over the batch for all seven rows (28
  // accumulators, LDS reads -69 %) 35.1 us -- the rows' streams then start only after the
  // whole product; the next rows' streams requested before the current arithmetic 33.8;
  // an approximate-arithmetic build 30.3; skipping the divisions that are exact no-ops
  // (unused clip scaling, x / bc1 once bc1 == 1.0f): no change.
  {
#if DZ_OF_AHEAD
    // Two register sets: the NEXT rows of the six streams are requested before this iteration's
    // arithmetic (fully unrolled: static set indices, no loop-carried copies of loaded registers --
    // a copy would wait for its load).  At 1.75 tile workgroups per CU a wave that computes with
    // nothing in flight leaves the memory pipe idle for the length of its arithmetic (~0.8 us per
    // iteration).  Same box, rocprofv3: 31.9-32.0 -> 30.3-31.0 us in the step, 30.9 -> 28.5 back to
    // back without the sample blocks (tools/adam_repeat_probe.sh); bit-identical.  Two rows ahead
    // (142 VGPRs, three waves per SIMD): 32.3; the streams' first rows requested before the LDS
    // fill's wait (DZ_OF_EARLY): 28.9 back to back -- no gain; tile blocks dispatched before the
    // sample blocks (DZ_OF_TILES_FIRST): 31.3.
#if DZ_OF_AHEAD >= 2   // (experiment: D rows ahead, D + 1 register sets)
    constexpr int D = DZ_OF_AHEAD, NS = D + 1;
    float4 S[NS][6];
    S[0][0] = pm; S[0][1] = mm; S[0][2] = vm; S[0][3] = ps; S[0][4] = ms; S[0][5] = vs;
#pragma unroll
    for (int d = 1; d < D; ++d) {
      const unsigned nm = om + d * rstep, ns = os + d * rstep;
      S[d][0] = ld_off(p, nm); S[d][1] = ld_off(m, nm); S[d][2] = ld_off(v, nm);
      S[d][3] = ld_off(p, ns); S[d][4] = ld_off(m, ns); S[d][5] = ld_off(v, ns);
    }
#pragma unroll
    for (int it = 0; it < IT; ++it) {
      float4& pm = S[it % NS][0]; float4& mm = S[it % NS][1]; float4& vm = S[it % NS][2];
      float4& ps = S[it % NS][3]; float4& ms = S[it % NS][4]; float4& vs = S[it % NS][5];
      if (it + D < IT) {
        const unsigned nm = om + D * rstep, ns = os + D * rstep;
        S[(it + D) % NS][0] = ld_off(p, nm); S[(it + D) % NS][1] = ld_off(m, nm); S[(it + D) % NS][2] = ld_off(v, nm);
        S[(it + D) % NS][3] = ld_off(p, ns); S[(it + D) % NS][4] = ld_off(m, ns); S[(it + D) % NS][5] = ld_off(v, ns);
        __builtin_amdgcn_sched_barrier(0);
      }
#else
    float4 S[2][6] = {{pm, mm, vm, ps, ms, vs}, {}};
#pragma unroll
    for (int it = 0; it < IT; ++it) {
      float4& pm = S[it & 1][0]; float4& mm = S[it & 1][1]; float4& vm = S[it & 1][2];
      float4& ps = S[it & 1][3]; float4& ms = S[it & 1][4]; float4& vs = S[it & 1][5];
      if (it + 1 < IT) {
        const unsigned nm = om + rstep, ns = os + rstep;
        S[(it + 1) & 1][0] = ld_off(p, nm); S[(it + 1) & 1][1] = ld_off(m, nm); S[(it + 1) & 1][2] = ld_off(v, nm);
        S[(it + 1) & 1][3] = ld_off(p, ns); S[(it + 1) & 1][4] = ld_off(m, ns); S[(it + 1) & 1][5] = ld_off(v, ns);
        __builtin_amdgcn_sched_barrier(0);
      }
#endif
#if DZ_ADAM_MFMA
      const float4 g4 = *(const float4*)(lds + (it * T::RP + rl) * T::GP + 4 * c4);
      float a0 = g4.x, a1 = g4.y, a2 = g4.z, a3 = g4.w;
#else
      const float* ft = s_ft + (it * T::RP + rl) * T::FS;
      float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#ifdef DZ_ADAM_NOG   // (timing probe, variant builds only: wrong numbers)
#define DZ_NOG_TRIPS 1
#else
#define DZ_NOG_TRIPS 8
#endif
#pragma unroll 1
      for (int bq = 0; bq < DZ_NOG_TRIPS; ++bq) {       // G[k][n] = sum_b x[b][k] dh1[b][n], b ascending
        const float4 f = *(const float4*)(ft + 4 * bq);
        const float fx[4] = {f.x, f.y, f.z, f.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float4 d = *(const float4*)(s_dh1 + (4 * bq + j) * C + 4 * c4);
          a0 = __builtin_fmaf(fx[j], d.x, a0); a1 = __builtin_fmaf(fx[j], d.y, a1);
          a2 = __builtin_fmaf(fx[j], d.z, a2); a3 = __builtin_fmaf(fx[j], d.w, a3);
        }
      }
#endif
      const float G[4] = {a0, a1, a2, a3};
      const float4 eo = *(const float4*)(s_eo + 4 * c4);
      const float ei = s_eo[C + it * T::RP + rl];
      const float EO[4] = {eo.x, eo.y, eo.z, eo.w};
      float* PM = (float*)&pm; float* MM = (float*)&mm; float* VM = (float*)&vm;
      float* PS = (float*)&ps; float* MS = (float*)&ms; float* VS = (float*)&vs;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float gm = G[j];
        asm volatile("" : "+v"(gm));
        adam_elem(PM[j], gm, MM[j], VM[j], pass, gn, bc1, bc2, lr, b1, b2, eps, max_norm);
        // sigma: the rounded product the stored-gradient path applies (adam_body)
        float gs = G[j] * (ei * EO[j]);
        asm volatile("" : "+v"(gs));
        adam_elem(PS[j], gs, MS[j], VS[j], pass, gn, bc1, bc2, lr, b1, b2, eps, max_norm);
      }
      st_off(m, om, mm); st_off(v, om, vm); st_off(p, om, pm);
      st_off(m, os, ms); st_off(v, os, vs); st_off(p, os, ps);
      om += rstep; os += rstep;
    }
#else
#pragma unroll 1
    for (int it = 0; it < IT; ++it) {
#if DZ_ADAM_MFMA
      const float4 g4 = *(const float4*)(lds + (it * T::RP + rl) * T::GP + 4 * c4);
      float a0 = g4.x, a1 = g4.y, a2 = g4.z, a3 = g4.w;
#else
      const float* ft = s_ft + (it * T::RP + rl) * T::FS;
      float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#ifdef DZ_ADAM_NOG   // (timing probe, variant builds only: wrong numbers)
#define DZ_NOG_TRIPS 1
#else
#define DZ_NOG_TRIPS 8
#endif
#pragma unroll 1
      for (int bq = 0; bq < DZ_NOG_TRIPS; ++bq) {       // G[k][n] = sum_b x[b][k] dh1[b][n], b ascending
        const float4 f = *(const float4*)(ft + 4 * bq);
        const float fx[4] = {f.x, f.y, f.z, f.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float4 d = *(const float4*)(s_dh1 + (4 * bq + j) * C + 4 * c4);
          a0 = __builtin_fmaf(fx[j], d.x, a0); a1 = __builtin_fmaf(fx[j], d.y, a1);
          a2 = __builtin_fmaf(fx[j], d.z, a2); a3 = __builtin_fmaf(fx[j], d.w, a3);
        }
      }
#endif
      const float G[4] = {a0, a1, a2, a3};
      const float4 eo = *(const float4*)(s_eo + 4 * c4);
      const float ei = s_eo[C + it * T::RP + rl];
      const float EO[4] = {eo.x, eo.y, eo.z, eo.w};
      float* PM = (float*)&pm; float* MM = (float*)&mm; float* VM = (float*)&vm;
      float* PS = (float*)&ps; float* MS = (float*)&ms; float* VS = (float*)&vs;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float gm = G[j];
        asm volatile("" : "+v"(gm));
        adam_elem(PM[j], gm, MM[j], VM[j], pass, gn, bc1, bc2, lr, b1, b2, eps, max_norm);
        // sigma: the rounded product the stored-gradient path applies (adam_body)
        float gs = G[j] * (ei * EO[j]);
        asm volatile("" : "+v"(gs));
        adam_elem(PS[j], gs, MS[j], VS[j], pass, gn, bc1, bc2, lr, b1, b2, eps, max_norm);
      }
      st_off(m, om, mm); st_off(v, om, vm); st_off(p, om, pm);
      st_off(m, os, ms); st_off(v, os, vs); st_off(p, os, ps);
      om += rstep; os += rstep;
      if (it + 1 < IT) {
        pm = ld_off(p, om); mm = ld_off(m, om); vm = ld_off(v, om);
        ps = ld_off(p, os); ms = ld_off(m, os); vs = ld_off(v, os);
      }
    }
#endif
  }
}

struct GramD {
  const float* dh1 = nullptr;   // [M][1024], finished (folded + masked by fc2's backward launch)
  int M = 0;
  const float* eps_out = nullptr;   // [1024]
  const double* gx_part = nullptr;  // GramX::part
  float* dot_out = nullptr;         // [kBlocks] non-negative partial sums of squares
};
struct GramDSide {
  typedef GramD Params;
  static constexpr int kCols = 64, kBlocks = 1024 / kCols, kLd = kCols + 4;
  __device__ static void run(const GramD& q, unsigned blk, float* smem, int smem_bytes) {
    const int h = (int)blk / (kBlocks / 2), n0 = (int)blk * kCols;   // head, first column of [0, 1024)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, i = lane & 15, kq = lane >> 4;
    const dz_d4* gp = (const dz_d4*)q.gx_part;
    // dh1 tile [32][kCols] into LDS
#pragma unroll
    for (int r = 0; r < kCols / 32; ++r) {
      const int e = tid + 256 * r, b = e / (kCols / 4), cc = e % (kCols / 4);
      float4 v = *(const float4*)(q.dh1 + (unsigned)(min(b, q.M - 1) * 1024 + n0 + 4 * cc));
      if (b >= q.M) v = dz_f4zero();
      *(float4*)(smem + b * kLd + 4 * cc) = v;
    }
    __syncthreads();
    const float* ta = smem + ((wave >> 1) * 16 + i) * kLd + 4 * kq;
    const float* tb = smem + ((wave & 1) * 16 + i) * kLd + 4 * kq;
    const float* eo = q.eps_out + n0 + 4 * kq;
    dz_d4 am = {0.0, 0.0, 0.0, 0.0}, as = {0.0, 0.0, 0.0, 0.0};
#pragma unroll 2
    for (int s = 0; s < kCols / 16; ++s) {
      const float4 a = *(const float4*)(ta + 16 * s), b = *(const float4*)(tb + 16 * s);
      const float4 ee = *(const float4*)(eo + 16 * s);
      const float A[4] = {a.x, a.y, a.z, a.w}, Bv[4] = {b.x, b.y, b.z, b.w};
      const float E[4] = {ee.x, ee.y, ee.z, ee.w};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const double da = (double)A[j], db = (double)Bv[j], de = (double)E[j];
        am = __builtin_amdgcn_mfma_f64_16x16x4f64(da, db, am, 0, 0, 0);
        as = __builtin_amdgcn_mfma_f64_16x16x4f64(da * de, db * de, as, 0, 0, 0);
      }
    }
    // fold the seven depth chunks of the two input Grams this head needs (two rounds of
    // seven 32-byte loads: all fourteen in flight would cost 112 registers)
    auto fold = [&](int v) {
      dz_d4 t[kGramChunks];
#pragma unroll
      for (int c = 0; c < kGramChunks; ++c) t[c] = gp[((v * kGramChunks + c) * 4 + wave) * 64 + lane];
      dz_d4 f = t[0];
#pragma unroll
      for (int c = 1; c < kGramChunks; ++c) f += t[c];
      return f;
    };
    const dz_d4 fx = fold(0);
    const dz_d4 fs = fold(1 + h);
    double d = 0.0;
#pragma unroll
    for (int r = 0; r < 4; ++r) d += fx[r] * am[r] + fs[r] * as[r];
    d = dz_wave_sum_f64(d);
    __syncthreads();                       // the tile is dead: its first bytes carry the sums
    double* red = (double*)smem;
    if (lane == 0) red[wave] = d;
    __syncthreads();
    if (tid == 0) q.dot_out[blk] = (float)((red[0] + red[1]) + (red[2] + red[3]));
  }
};

__device__ __forceinline__ void adam_flat_ranges(unsigned bid, unsigned nblk, const AdamRanges& rg,
                                                 float* __restrict__ p, const float* __restrict__ g,
                                                 float* __restrict__ m, float* __restrict__ v,
                                                 const AdamScalars& sc, float lr, float b1, float b2,
                                                 float eps, float max_norm) {
  const long total = rg.n[0] + rg.n[1] + rg.n[2];
  auto at = [&](long i) {
    return i < rg.n[0] ? rg.lo[0] + i
                       : (i < rg.n[0] + rg.n[1] ? rg.lo[1] + (i - rg.n[0])
                                                : rg.lo[2] + (i - rg.n[0] - rg.n[1]));
  };
  for (long i = (long)bid * 256 + threadIdx.x; i < total; i += (long)nblk * 256) {
    const long o = at(i);
    float4 gv = ((const float4*)g)[o], mv = ((const float4*)m)[o], vv = ((const float4*)v)[o];
    float4 pv = ((const float4*)p)[o];
    float* G = (float*)&gv; float* M = (float*)&mv; float* V = (float*)&vv; float* P = (float*)&pv;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float gj = G[j];
      asm volatile("" : "+v"(gj));
      adam_elem(P[j], gj, M[j], V[j], sc.pass, sc.gn, sc.bc1, sc.bc2, lr, b1, b2, eps, max_norm);
    }
    ((float4*)m)[o] = mv; ((float4*)v)[o] = vv; ((float4*)p)[o] = pv;
  }
}

// One optimiser launch: [sample+gather side blocks | fc1 tiles | flat ranges] (the priority
// write-back rides in the conv3 backward launch whenever this optimiser is used).
// Measured tile shapes (rows x columns per workgroup, optimiser role alone): 8 x 128
// 36.6 us, 16 x 128 34.3, 32 x 128 34.6, 56 x 128 33.2, 112 x 64 33.0; the next rows'
// streams requested before the current arithmetic 33.8 (no gain: the arithmetic is the
// exposed part, see adam_fc1_block).
#ifndef DZ_OF_C   // columns of a tile: 128 since round 6 (56 rows x 128 columns: 512 contiguous bytes per row and
                  // stream; same box, six bench lines each: 7 079 -> 7 104 steps/s against 112 x 64, bit-identical)
#define DZ_OF_C 128
#endif
#ifndef DZ_OF_IT
#define DZ_OF_IT 7
#endif
constexpr int kOfC = DZ_OF_C, kOfIT = DZ_OF_IT, kOfFlatBlocks = 64;
// (compiled for FOUR waves per SIMD since round 6: the second register set of the look-ahead needs 118
// VGPRs, and the launch's 801 workgroups -- 289 sample + gather, 448 tiles, 64 flat -- still fit at four
// per CU.  Rounds 3-5: six, the occupancy the 77-register form reached; 5 with the look-ahead spills
// (39 us), 3: the workgroups no longer fit at once (32.4 tiles first, 34.7 sample blocks first).)
#ifndef DZ_OF_OCC
#define DZ_OF_OCC 4
#endif
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(DZ_OF_OCC, DZ_OF_OCC)))
void adam_onfly_kernel(
    float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
    float* __restrict__ v, const float* __restrict__ part, int nparts,
    const int32_t* __restrict__ count, const float* __restrict__ losses,
    const float* __restrict__ weights, int B, float* __restrict__ scal, float lr, float b1,
    float b2, float eps, float max_norm, Fc1OnFly q, AdamRanges rg, SampleGatherParams sg,
    unsigned sg_blocks, const unsigned* abort) {
  __shared__ __attribute__((aligned(16))) float lds[OfTile<kOfC, kOfIT>::kLds];
  __shared__ float red[4];
  constexpr unsigned fc1_blocks = OfTile<kOfC, kOfIT>::kBlocks;
  unsigned bid = blockIdx.x;
#ifndef DZ_OF_TILES_FIRST
#define DZ_OF_TILES_FIRST 0
#endif
#if DZ_OF_TILES_FIRST
  // block ids: [fc1 tiles | flat ranges | next sample + gather]: the streams' workgroups get their
  // slots first, the short side blocks fill what is left of every CU
  if (bid >= fc1_blocks + kOfFlatBlocks) { SampleGatherSide::run(sg, bid - (fc1_blocks + kOfFlatBlocks)); return; }
  if (abort && *abort) return;
#else
  if (bid < sg_blocks) { SampleGatherSide::run(sg, bid); return; }
  if (abort && *abort) return;   // the step is void (adam_kernel); the next step's sample is still good
  bid -= sg_blocks;
#endif
  if (bid < fc1_blocks) {
    adam_fc1_block<kOfC, kOfIT>(bid, q, p, m, v, part, nparts, count, lr, b1, b2, eps, max_norm,
                                red, lds);
    return;
  }
  const AdamScalars sc = adam_scalars(part, nparts, count, b1, b2, max_norm, red);
  if (bid == fc1_blocks && threadIdx.x == 0) adam_publish(sc, scal, losses, weights, B);
  adam_flat_ranges(bid - fc1_blocks, kOfFlatBlocks, rg, p, g, m, v, sc, lr, b1, b2, eps, max_norm);
}
static inline unsigned adam_onfly_blocks(unsigned sg_blocks) {
  return sg_blocks + (unsigned)OfTile<kOfC, kOfIT>::kBlocks + kOfFlatBlocks;
}
}  // namespace
