// Input gradient of a noisy linear layer with a small batch (M <= 32) as a ROW-OWNING
// weight stream:
//     dX[b][k] = relu'(x[b][k]) * sum_n dY[b][n] * W_eff[k][n],
//     W_eff[k][n] = Wmu[k][n] + Wsig[k][n] * (eps_in[k] * eps_out[n])        (networks.py:168-176)
// The reduction index n is the CONTIGUOUS one in memory, so it cannot be the K dimension
// of an MFMA without a transposing stage; the tile-GEMM form (global -> registers -> LDS
// -> MFMA, split over n into slabs + a fold) read fc1's 25.7 MB of weights at 2.7 TB/s and
// needed 10.4 + 4.8 us (contraction + reduce launch).  Here a workgroup OWNS whole rows k:
// a wave streams up to 256 columns of a row as one float4 per lane per matrix (1 KB
// contiguous per instruction), keeps its four columns of dY for all 32 batch rows in
// registers (128 VGPRs, as PAIRS of batch rows: the multiply-adds are v_pk_fma_f32),
// and reduces the 32 partial sums across the wave with the gfx950 lane-swap instructions
// (v_permlane32_swap / v_permlane16_swap: one swap + one add halve two registers into
// one -- a transposed butterfly).  No split over n, no slabs, no reduce launch, the ReLU
// mask in the epilogue.
//
// A "job" is one chunk of <= 256 columns of one head; a row's jobs go to the workgroup's
// four waves round-robin (fc1: 2 heads x 2 chunks = one job per wave; fc2 with 6 actions:
// 2 + 1 jobs).  The heads either add into ONE output (fc1: both read the same input) or
// own one output block each (fc2: adv | val halves of the hidden layer).
// ref: rainbow/agent.py:112-118 (jax.grad through the network), networks.py:150-180.
#pragma once
#include "dz_qnet_ops.h"
#include "dz_seam.h"

#ifndef DZ_HC_NAP_D
#define DZ_HC_NAP_D 24
#endif
namespace {
struct RowDgrad {
  const float* params;
  const float* noise;
  FcHead head[2];         // w_mu, w_sig, ldw, N (columns), eps_in (per row), eps_out (per column),
                          // out_off = column offset of the head's dY
  const float* dy; int ldy;     // [M][ldy]
  const float* mask;      // [M][ldo] post-ReLU activation of the layer input
  float* out;             // [M][ldo]
  int ldo;
  int out_col[2];         // output column of row 0, per head
  int same_out;           // 1: both heads add into out_col[0] + k; 0: head h owns out_col[h] + k
  int M, K;               // batch rows (<= 32), weight rows
  int nblocks;            // row-owning workgroups (rows are split as evenly as K allows)
  // SEAM builds (row_dgrad_block<..., SEAM = true>; dz_head_chain.h): `dy` and `mask` are produced
  // by other workgroups of the SAME launch and read as seams (dz_seam.h); on a timeout the sticky
  // word is set and `poison[0..M)` (the step's losses) becomes NaN
  unsigned* fail = nullptr; int limit = 0; float* poison = nullptr;
  int watch_col = 0;      // column of dy every producer stores last
  long long* dbg = nullptr;
};
// LDS: [row][job][lane group 0..7][32 batch rows + 4]: the +4 skews the eight groups over
// the banks (a lane writes one float4 at group * 36 + 4 * (lane >> 3); without it the
// eight lanes served per cycle hit the same four banks: 704 k conflict cycles per fc1
// launch in the SQ counters).  rows-per-workgroup x jobs <= 32.
constexpr int kRdGroup = 36;
constexpr int kRdLdsFloats = 32 * 8 * kRdGroup;

typedef float dz_f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float dz_dpp_xor8(float v) {   // lane l <- lane l ^ 8
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(
      0, __builtin_bit_cast(int, v), 0x128 /* row_ror:8 */, 0xf, 0xf, false));
}
// v_permlane32_swap: a's lanes 32..63 <-> b's lanes 0..31; v_permlane16_swap: a's odd
// 16-lane rows <-> b's even rows (checked on hardware: tools/micro/sw_test).  Inline
// assembly: this compiler's __builtin_amdgcn_permlane{16,32}_swap loses the second result
// when both feed one add (it emitted v_add vdst, vdst), and inline assembly is opaque to
// the hazard recogniser -- hence the explicit wait states around each group of four.
#define DZ_SWAP4(INSN, A0, A1, A2, A3, B0, B1, B2, B3)                                        \
  asm volatile("s_nop 1\n\t" INSN " %0, %4\n\t" INSN " %1, %5\n\t" INSN " %2, %6\n\t" INSN        \
               " %3, %7\n\ts_nop 1"                                                            \
               : "+v"(A0), "+v"(A1), "+v"(A2), "+v"(A3), "+v"(B0), "+v"(B1), "+v"(B2), "+v"(B3))
__device__ __forceinline__ void dz_swap32x4(dz_f2& a0, dz_f2& a1, dz_f2& b0, dz_f2& b1) {
  float x0 = a0.x, x1 = a0.y, x2 = a1.x, x3 = a1.y, y0 = b0.x, y1 = b0.y, y2 = b1.x, y3 = b1.y;
  DZ_SWAP4("v_permlane32_swap_b32", x0, x1, x2, x3, y0, y1, y2, y3);
  a0.x = x0; a0.y = x1; a1.x = x2; a1.y = x3; b0.x = y0; b0.y = y1; b1.x = y2; b1.y = y3;
}
__device__ __forceinline__ void dz_swap16x4(dz_f2& a0, dz_f2& a1, dz_f2& b0, dz_f2& b1) {
  float x0 = a0.x, x1 = a0.y, x2 = a1.x, x3 = a1.y, y0 = b0.x, y1 = b0.y, y2 = b1.x, y3 = b1.y;
  DZ_SWAP4("v_permlane16_swap_b32", x0, x1, x2, x3, y0, y1, y2, y3);
  a0.x = x0; a0.y = x1; a1.x = x2; a1.y = x3; b0.x = y0; b0.y = y1; b1.x = y2; b1.y = y3;
}

static inline int row_dgrad_jobs(const RowDgrad& q) {
  return (q.head[0].N + 255) / 256 + (q.head[1].N + 255) / 256;
}
static inline int row_dgrad_max_rows(const RowDgrad& q) { return (q.K + q.nblocks - 1) / q.nblocks; }

// `lds`: kRdLdsFloats floats.  P = rows in flight per wave.
// NJ0 / NJ1: column chunks of head 0 / head 1 (compile time: the job count shapes the
// LDS indexing and, with at most four jobs, removes the job loop); SAME = same_out.
// NOISY = false: plain linear layer (W_eff = W: no sigma matrix, no noise).
// With one or two jobs per row the four waves form 4 / jobs ROW GROUPS (group g takes rows
// g, g + groups, ...), so that no wave idles.
template <int NJ0, int NJ1, bool SAME, int P = 2, bool NOISY = true, bool SEAM = false>
__device__ __forceinline__ void row_dgrad_block(const RowDgrad& q, unsigned blk, float* lds) {
  // NJ0 == 0: the job count is a run-time value (head 0 only; more registers, for the
  // rare very wide heads)
  constexpr bool RT = NJ0 == 0;
  constexpr bool ONE = !RT && NJ0 + NJ1 <= 4;
  constexpr int GROUPS = (!RT && NJ0 + NJ1 <= 2) ? 4 / (NJ0 + NJ1) : 1;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int k0 = (int)(((long)blk * q.K) / q.nblocks), k1 = (int)(((long)(blk + 1) * q.K) / q.nblocks);
  const int nrows = k1 - k0;
  const int nj0 = RT ? (q.head[0].N + 255) >> 8 : NJ0, nj = nj0 + NJ1;
  constexpr int nout = SAME ? 1 : 2;
  // the ReLU mask of this thread's outputs (epilogue), requested now
  constexpr int NO = 2;   // outputs per thread: nrows * 32 * nout <= 512
  float mk[NO];
#pragma unroll
  for (int j = 0; j < NO; ++j) {
    const int o = min(tid + 256 * j, nrows * 32 * nout - 1);
    const int sel = o >= nrows * 32 ? 1 : 0, r = (o - sel * nrows * 32) >> 5, b = min(o & 31, q.M - 1);
    const float* mp = q.mask + (long)b * q.ldo + (sel ? q.out_col[1] : q.out_col[0]) + k0 + r;
    // (SEAM: possibly not written yet -- looked at again in the epilogue, when it must be)
    mk[j] = SEAM ? act_load(mp) : *mp;
  }
  bool wave_fail = false;
  const int grp = GROUPS > 1 ? wave / nj : 0;
  const int myrows = (nrows - grp + GROUPS - 1) / GROUPS;   // rows k0 + grp + GROUPS * i
  for (int job = GROUPS > 1 ? wave % nj : wave; job < nj; job += ONE ? 1 << 20 : 4) {   // (wave-uniform)
    const bool h1 = job >= nj0;
    const int c0 = ((h1 ? job - nj0 : job) << 8) + 4 * lane;      // this lane's first column
    const int N = dz_val(h1, q.head[1].N, q.head[0].N), ldw = dz_val(h1, q.head[1].ldw, q.head[0].ldw);
    const int cc = min(c0, ldw - 4);                                // clamped (pitch is a multiple of 4)
    // (uniform base + 32-bit byte offset: SGPR-base addressing, no 64-bit lane addresses)
    const char* __restrict__ wmu = (const char*)(q.params + dz_val(h1, q.head[1].w_mu, q.head[0].w_mu));
    const char* __restrict__ wsg = (const char*)(q.params + dz_val(h1, q.head[1].w_sig, q.head[0].w_sig));
    const float* __restrict__ ein =
        NOISY ? q.noise + dz_val(h1, q.head[1].eps_in, q.head[0].eps_in) : nullptr;
    // the row's noise factor travels WITH the row's weights, P rows ahead: requested in the
    // iteration that uses it (rounds 1-3) it was one exposed memory round trip per row -- the
    // ISA had `s_waitcnt vmcnt(2)` right behind the three loads of each row
    float4 pm[P], ps[P];
    float pe[P];
#pragma unroll
    for (int u = 0; u < P; ++u) {
      const int kr = min(k0 + grp + GROUPS * u, k1 - 1);
      const unsigned o = (unsigned)(kr * ldw + cc) * 4u;
      pm[u] = *(const float4*)(wmu + o);
      if (NOISY) { ps[u] = *(const float4*)(wsg + o); pe[u] = ein[kr]; }
    }
    // this lane's four columns of dY, all batch rows, as pairs of batch rows; columns
    // beyond the head's N (and lanes beyond the chunk) contribute zeros
    const char* dyp = (const char*)(q.dy + dz_val(h1, q.head[1].out_off, q.head[0].out_off));
    const float v0 = c0 < N ? 1.f : 0.f, v1 = c0 + 1 < N ? 1.f : 0.f;
    const float v2 = c0 + 2 < N ? 1.f : 0.f, v3 = c0 + 3 < N ? 1.f : 0.f;
    dz_f2 D[16][4];
    if constexpr (SEAM) {
      // dY comes from the loss role of this launch.  The wave first watches ONE word per sample (the
      // word each sample's loss workgroup stores last: q.watch_col), one load per lane and round;
      // then whole-wave rounds re-read the wave's columns of all batch rows until none of the
      // real ones is missing (normally one round: each costs 64 loads per lane).
      {
        const float* word = q.dy + (unsigned)(min(lane, q.M - 1) * q.ldy) + q.watch_col;
        for (int i = 0; i < q.limit; ++i) {
          const bool m = lane < 32 && act_missing(act_load(word));
          if (__builtin_amdgcn_ballot_w64(m) == 0ull) break;
          __builtin_amdgcn_s_sleep(DZ_HC_NAP_D);
        }
      }
#ifdef DZ_HC_STAMPS
      if (q.dbg && tid == 0) q.dbg[blockIdx.x * 8 + 1] = (long long)wall_clock64();
#endif
      int round = 0;
      bool again;
      do {
        // (all loads first, then the checks, branch-free)
        float4 ra[16][2];
        const __amdgpu_buffer_rsrc_t dr = act_rsrc(q.dy);
        const unsigned ho = (unsigned)(dz_val(h1, q.head[1].out_off, q.head[0].out_off) + cc) * 4u;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          ra[i][0] = act_load4(dr, (unsigned)(min(2 * i, q.M - 1) * q.ldy) * 4u + ho);
          ra[i][1] = act_load4(dr, (unsigned)(min(2 * i + 1, q.M - 1) * q.ldy) * 4u + ho);
        }
        unsigned all = 1u;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const float2 a0 = make_float2(ra[i][0].x, ra[i][0].y), a1 = make_float2(ra[i][0].z, ra[i][0].w);
          const float2 b0 = make_float2(ra[i][1].x, ra[i][1].y), b1 = make_float2(ra[i][1].z, ra[i][1].w);
          const float m0 = 2 * i < q.M ? 1.f : 0.f, m1 = 2 * i + 1 < q.M ? 1.f : 0.f;
          all &= (v0 == 0.f || (!act_missing(a0.x) && !act_missing(b0.x))) ? 1u : 0u;
          all &= (v1 == 0.f || (!act_missing(a0.y) && !act_missing(b0.y))) ? 1u : 0u;
          all &= (v2 == 0.f || (!act_missing(a1.x) && !act_missing(b1.x))) ? 1u : 0u;
          all &= (v3 == 0.f || (!act_missing(a1.y) && !act_missing(b1.y))) ? 1u : 0u;
          D[i][0] = dz_f2{a0.x * (m0 * v0), b0.x * (m1 * v0)}; D[i][1] = dz_f2{a0.y * (m0 * v1), b0.y * (m1 * v1)};
          D[i][2] = dz_f2{a1.x * (m0 * v2), b1.x * (m1 * v2)}; D[i][3] = dz_f2{a1.y * (m0 * v3), b1.y * (m1 * v3)};
        }
        const bool miss = all == 0u;
        again = __builtin_amdgcn_ballot_w64(miss) != 0ull;
        if (again) {
          if (round++ >= q.limit) { wave_fail = true; again = false; }
          else __builtin_amdgcn_s_sleep(1);
        }
      } while (again);
#ifdef DZ_HC_STAMPS
      if (q.dbg && tid == 0) q.dbg[blockIdx.x * 8 + 2] = (long long)wall_clock64();
#endif
    } else {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const float4 d0 = *(const float4*)(dyp + (unsigned)(min(2 * i, q.M - 1) * q.ldy + cc) * 4u);
        const float4 d1 = *(const float4*)(dyp + (unsigned)(min(2 * i + 1, q.M - 1) * q.ldy + cc) * 4u);
        const float m0 = 2 * i < q.M ? 1.f : 0.f, m1 = 2 * i + 1 < q.M ? 1.f : 0.f;
        D[i][0] = dz_f2{d0.x * (m0 * v0), d1.x * (m1 * v0)}; D[i][1] = dz_f2{d0.y * (m0 * v1), d1.y * (m1 * v1)};
        D[i][2] = dz_f2{d0.z * (m0 * v2), d1.z * (m1 * v2)}; D[i][3] = dz_f2{d0.w * (m0 * v3), d1.w * (m1 * v3)};
      }
    }
    float4 eo = dz_f4zero();
    if (NOISY) eo = *(const float4*)(q.noise + dz_val(h1, q.head[1].eps_out, q.head[0].eps_out) + cc);
#pragma unroll 1
    for (int i0 = 0; i0 < myrows; i0 += P) {
#pragma unroll
      for (int u = 0; u < P; ++u) {
        if (i0 + u < myrows) {   // (wave-uniform)
          const int r = grp + GROUPS * (i0 + u);   // row of the workgroup
          // W_eff of THIS row and its 48 multiply-adds first, from the registers loaded P rows
          // ago; only then (below, in front of the butterfly) are those registers re-requested
          // for row r + P.  (Loads first, as in rounds 1-3, leaves the old values -- W_eff is
          // formed in place -- live under the new ones: the new ones land in temporaries and are
          // `v_mov`ed into the loop-carried registers at the end of the iteration, behind an
          // `s_waitcnt vmcnt(0..2)`: one exposed memory round trip per row.)
          float w0 = pm[u].x, w1 = pm[u].y, w2 = pm[u].z, w3 = pm[u].w;
          if (NOISY) {
            const float e = pe[u];
            w0 = __builtin_fmaf(ps[u].x, e * eo.x, w0); w1 = __builtin_fmaf(ps[u].y, e * eo.y, w1);
            w2 = __builtin_fmaf(ps[u].z, e * eo.z, w2); w3 = __builtin_fmaf(ps[u].w, e * eo.w, w3);
          }
          dz_f2 a[16];
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            dz_f2 t = D[i][0] * w0;
            t = __builtin_elementwise_fma(D[i][1], dz_f2{w1, w1}, t);
            t = __builtin_elementwise_fma(D[i][2], dz_f2{w2, w2}, t);
            a[i] = __builtin_elementwise_fma(D[i][3], dz_f2{w3, w3}, t);
          }
          __builtin_amdgcn_sched_barrier(0);
          {
            const int kr = min(k0 + r + GROUPS * P, k1 - 1);
            const unsigned o = (unsigned)(kr * ldw + cc) * 4u;
            pm[u] = *(const float4*)(wmu + o);
            if (NOISY) { ps[u] = *(const float4*)(wsg + o); pe[u] = ein[kr]; }
          }
          __builtin_amdgcn_sched_barrier(0);
          // transposed butterfly over lane bits 5, 4, 3: 32 -> 16 -> 8 -> 4 values per
          // lane (a[i] = batch rows 2i, 2i+1)
#pragma unroll
          for (int i = 0; i < 8; i += 2) { dz_swap32x4(a[i], a[i + 1], a[i + 8], a[i + 9]); }
#pragma unroll
          for (int i = 0; i < 8; ++i) a[i] += a[i + 8];
#pragma unroll
          for (int i = 0; i < 4; i += 2) { dz_swap16x4(a[i], a[i + 1], a[i + 4], a[i + 5]); }
#pragma unroll
          for (int i = 0; i < 4; ++i) a[i] += a[i + 4];
          const bool hi8 = (lane & 8) != 0;
          float4 v;
          float* V = (float*)&v;
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            const dz_f2 own = hi8 ? a[i + 2] : a[i], oth = hi8 ? a[i] : a[i + 2];
            V[2 * i] = own.x + dz_dpp_xor8(oth.x); V[2 * i + 1] = own.y + dz_dpp_xor8(oth.y);
          }
          // lane l now holds batch rows 4 (l >> 3) + {0..3}, summed over the 8 lanes that
          // share l & 7
          *(float4*)(lds + ((r * nj + job) * 8 + (lane & 7)) * kRdGroup + 4 * (lane >> 3)) = v;
        }
      }
    }
  }
  if constexpr (SEAM) {
    if (__syncthreads_or(wave_fail ? 1 : 0)) {   // dY never came: sticky word, the step's losses are not numbers
      if (tid == 0) __hip_atomic_store(q.fail, 1u, DZ_ACT_RLX);
      if (q.poison && tid < q.M) q.poison[tid] = __builtin_nanf("");
      return;
    }
  } else {
    __syncthreads();
  }
#pragma unroll
  for (int j = 0; j < NO; ++j) {
    const int o = tid + 256 * j;
    if (o < nrows * 32 * nout) {
      const int sel = o >= nrows * 32 ? 1 : 0, r = (o - sel * nrows * 32) >> 5, b = o & 31;
      if constexpr (SEAM) {
        // dY was seen, so the activation it was computed from is at the coherence point
        if (act_missing(mk[j]))
          mk[j] = act_load(q.mask + (long)min(b, q.M - 1) * q.ldo + (sel ? q.out_col[1] : q.out_col[0]) + k0 + r);
      }
      // this output's jobs (ascending), eight lane groups each
      const int j0 = SAME ? 0 : (sel ? nj0 : 0), j1 = SAME ? nj : (sel ? nj : nj0);
      float s = 0.f;
      for (int jb = j0; jb < j1; ++jb) {
        float x[8];
#pragma unroll
        for (int g = 0; g < 8; ++g) x[g] = lds[((r * nj + jb) * 8 + g) * kRdGroup + b];
#pragma unroll
        for (int g = 0; g < 8; ++g) s += x[g];
      }
      if (b < q.M)
        q.out[(long)b * q.ldo + (sel ? q.out_col[1] : q.out_col[0]) + k0 + r] = mk[j] > 0.f ? s : 0.f;
    }
  }
}
}  // namespace
