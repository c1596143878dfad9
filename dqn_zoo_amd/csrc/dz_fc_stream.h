// Weight-streaming kernels for the wide linear layer (fc1: 3136 -> 2x512, batch
// <= 32): the one layer of the DQN-family nets whose cost is reading its weights
// (25.7 MB for mu+sigma) rather than arithmetic.
//
// With M <= 32 rows the contraction is a "fat GEMV": the tile-GEMM skeleton
// (global -> registers -> LDS -> barrier -> MFMA) has only one stage of loads in
// flight per workgroup and measured 1.2 TB/s.  Here every wave streams its slice
// of W from HBM straight into MFMA operand registers -- no LDS, no barrier in
// the loop -- with the next group of loads issued before the current group's
// MFMAs (register double buffer), as cdna_hip_programming.md prescribes for
// M <= 16 decode weights ("load straight to VGPRs, deep unroll, late vmcnt").
//
// Operand trick: v_mfma_f32_32x32x2_f32 wants lane l to hold B[k=l>>5][j=l&31].
// A 16-byte load along the contiguous dimension of W gives a lane 4 neighbouring
// elements instead; they are consumed by 4 MFMAs:
//   forward : W[k][n..n+3] -> 4 output-column sets (4 accumulators, column 4j+c)
//   dgrad   : W[k][n..n+3] -> 4 reduction steps (one accumulator)
// and the two lane halves work on different k (forward) / n (dgrad) groups, so a
// wave-load is 2 x 512 contiguous bytes (forward) or 32 rows x 32 bytes (dgrad).
#pragma once

#include "dz_qnet_ops.h"

namespace {  // internal linkage: this header is included by several .hip files

// ------------------------------- forward ------------------------------------ //
// part[split][g*M + m][out_off + n] = sum over the split's k of
//     x[m][k] Wmu[k][n]  (+ (x[m][k] eps_in[k]) (Wsig[k][n] eps_out[n]) when noisy)
// grid = (strips of 128 columns over all heads, G * S); block = 256 = 4 waves,
// each wave takes a quarter of the block's k-range; LDS reduce at the end.
struct FcStreamFwdParams {
  const float* x;   // [G*M][ldx]
  int ldx;
  int M;            // <= 32
  int G;
  int NH;
  int S;            // k-splits per (strip, group); partial slabs written
  int noisy;
  const float* params[DZ_MAX_GROUPS];
  const float* noise[DZ_MAX_GROUPS];
  FcHead head[2];   // N must be a multiple of 128, K a multiple of 8
  float* part;      // [S][G*M][ldo]
  int ldo;
};

__global__ __launch_bounds__(256) void dz_fc_stream_fwd(FcStreamFwdParams p) {
  __shared__ __attribute__((aligned(16))) float red[3 * 64 * 64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int half = lane >> 5, l31 = lane & 31;
  const int strips0 = p.head[0].N / 128;
  const int h_idx = blockIdx.x >= strips0 ? 1 : 0;
  const FcHead hd = p.head[h_idx];
  const int n0 = (blockIdx.x - (h_idx ? strips0 : 0)) * 128;
  const int g = blockIdx.y / p.S, split = blockIdx.y % p.S;
  const float* __restrict__ prm = p.params[g];
  const float* __restrict__ nz = p.noise[g];

  // reduction rows in units of 8: [mu rows | sigma rows]
  const int it_per_part = hd.K / 8;
  const int it_total = it_per_part * (p.noisy ? 2 : 1);
  const int nsplit = p.S * 4;
  const int per = (it_total + nsplit - 1) / nsplit;
  const int it_begin = (split * 4 + wave) * per;
  const int it_end = min(it_total, it_begin + per);

  const int m = min(l31, p.M - 1);
  const float mrow_ok = l31 < p.M ? 1.f : 0.f;
  const float* xrow = p.x + (long)(g * p.M + m) * p.ldx + hd.x_off + 4 * half;
  const int ncol = n0 + 4 * l31;
  const float4 eo = dz_ld4(nz + hd.eps_out + ncol);

  f32x16 acc[4];
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[c][i] = 0.f;

  constexpr int U = 2;  // iterations (of 8 rows) per register buffer
  float4 xa[2][U], wb[2][U][4];

  auto issue = [&](int buf, int it0) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int it = min(it0 + u, it_total - 1);
      const bool sig = it >= it_per_part;
      const int k = (it - (sig ? it_per_part : 0)) * 8;  // + 4*half + t
      float4 a = dz_ld4(xrow + k);
      const float4 e = dz_ld4(nz + hd.eps_in + k + 4 * half);
      a = sig ? dz_mul4(a, e) : a;
      const bool live = (it0 + u) < it_end;
      xa[buf][u] = dz_scale4(a, live ? mrow_ok : 0.f);
      const float* wrow = prm + (sig ? hd.w_sig : hd.w_mu) + (long)(k + 4 * half) * hd.ldw + ncol;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const float4 w = dz_ld4(wrow + (long)t * hd.ldw);
        wb[buf][u][t] = sig ? dz_mul4(w, eo) : w;
      }
    }
  };
  auto consume = [&](int buf) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const float a4[4] = {xa[buf][u].x, xa[buf][u].y, xa[buf][u].z, xa[buf][u].w};
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const float4 w = wb[buf][u][t];
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[t], w.x, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[t], w.y, acc[1], 0, 0, 0);
        acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[t], w.z, acc[2], 0, 0, 0);
        acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[t], w.w, acc[3], 0, 0, 0);
      }
    }
  };

  if (it_begin < it_end) {
    issue(0, it_begin);
    int it = it_begin;
    while (true) {
      if (it + U < it_end) issue(1, it + U);
      consume(0);
      it += U;
      if (it >= it_end) break;
      if (it + U < it_end) issue(0, it + U);
      consume(1);
      it += U;
      if (it >= it_end) break;
    }
  }

  // cross-wave reduction through LDS: [wave-1][c][reg][lane]
  if (wave > 0) {
    float* dst = red + (wave - 1) * 64 * 64 + lane;
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int i = 0; i < 16; ++i) dst[(c * 16 + i) * 64] = acc[c][i];
  }
  __syncthreads();
  if (wave > 0) return;
#pragma unroll
  for (int w = 0; w < 3; ++w) {
    const float* src = red + w * 64 * 64 + lane;
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[c][i] += src[(c * 16 + i) * 64];
  }
  float* base = p.part + ((long)split * p.G * p.M + (long)g * p.M) * p.ldo + hd.out_off + ncol;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int mm = dz_acc_row(r, lane);
    if (mm < p.M)
      *(float4*)(base + (long)mm * p.ldo) = dz_f4(acc[0][r], acc[1][r], acc[2][r], acc[3][r]);
  }
}


// ------------------------- forward, dword-operand form ----------------------- //
// Same contraction as dz_fc_stream_fwd, organised for DEPTH of loads in flight
// rather than width: lane l loads the single float W[k + (l>>5)][n0 + (l&31)],
// which IS its MFMA B operand, so a load costs one VGPR and 16 of them (4 KB per
// wave) fly per register buffer at ~100 VGPRs per wave => 4-5 waves/SIMD, i.e.
// tens of KB in flight per CU, which is what HBM latency needs.  The A operand
// (x, or x.eps_in for sigma rows) is staged once per workgroup in LDS, k-major,
// so the per-MFMA ds_read_b32 is conflict-free.  One wave = 32 output columns;
// a workgroup = 4 adjacent column strips sharing one k-range.
struct FcStreamFwd2Params {
  const float* x; int ldx; int M; int G; int NH; int S; int noisy;
  const float* params[DZ_MAX_GROUPS];
  const float* noise[DZ_MAX_GROUPS];
  FcHead head[2];   // N multiple of 128, K even
  float* part;      // [S][G*M][ldo]
  int ldo;
  int rows_per_split;  // even, <= DZ_FC2_MAX_ROWS
  int blocked;         // EXPERIMENT: address W as [strip][K][128] (timing only)
};
#define DZ_FC2_NLOAD 98                     // dword loads per lane
#define DZ_FC2_MAX_ROWS (2 * DZ_FC2_NLOAD)  // rows one workgroup covers

__global__ __launch_bounds__(256) void dz_fc_stream_fwd2(FcStreamFwd2Params p) {
  extern __shared__ __attribute__((aligned(16))) float xs[];  // [rows_per_split][32]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int half = lane >> 5, l31 = lane & 31;
  const int strips0 = p.head[0].N / 128;
  const int h_idx = blockIdx.x >= strips0 ? 1 : 0;
  const FcHead hd = p.head[h_idx];
  const int n0 = (blockIdx.x - (h_idx ? strips0 : 0)) * 128 + 32 * wave;
  const int g = blockIdx.y / p.S, split = blockIdx.y % p.S;
  const float* __restrict__ prm = p.params[g];
  const float* __restrict__ nz = p.noise[g];
  const int K = hd.K;
  const int rtotal = K * (p.noisy ? 2 : 1);
  const int r0 = split * p.rows_per_split;
  const int r1 = min(rtotal, r0 + p.rows_per_split);
  const int nrows = max(r1 - r0, 0);

  const int ncol = n0 + l31;
  const float eo = nz[hd.eps_out + ncol];

  // (1) ALL of the wave's weight loads are issued first (NL dword loads = NL
  // VGPRs): they do not depend on LDS, so they fly during the staging below;
  // the MFMAs later drain them in order behind counted vmcnt waits.  This is
  // the deepest pipeline the register file allows.
  constexpr int NL = DZ_FC2_NLOAD;
  float wb[NL];
#pragma unroll
  for (int u = 0; u < NL; ++u) {
    const int r = min(r0 + 2 * u + half, rtotal - 1);
    const bool sig = r >= K;
    const int k = r - (sig ? K : 0);
    const long off = p.blocked
        ? (long)blockIdx.x * K * 128 + (long)k * 128 + 32 * wave + l31
        : (long)k * hd.ldw + ncol;
    const float w = prm[(sig ? hd.w_sig : hd.w_mu) + off];
    wb[u] = sig ? w * eo : w;
  }

  // (2) stage A: xs[r - r0][m] = x[m][k] (* eps_in[k] on sigma rows).  All
  // gathers are issued before the first LDS write (static unroll, masked).
  {
    const int mm = threadIdx.x & 31;       // batch row
    const int q0 = threadIdx.x >> 5;       // 8 row-quads per pass
    const int mc = min(mm, p.M - 1);
    const float ok = mm < p.M ? 1.f : 0.f;
    const float* xrow = p.x + (long)(g * p.M + mc) * p.ldx + hd.x_off;
    constexpr int NP = (DZ_FC2_MAX_ROWS / 4 + 7) / 8;  // passes
    float4 v[NP];
#pragma unroll
    for (int j = 0; j < NP; ++j) {
      const int r = min(r0 + 4 * (q0 + 8 * j), rtotal - 4);  // r0, K multiples of 4
      const bool sig = r >= K;
      const int k = r - (sig ? K : 0);
      const float4 xv = dz_ld4(xrow + k);
      const float4 e = dz_ld4(nz + hd.eps_in + k);
      v[j] = dz_scale4(sig ? dz_mul4(xv, e) : xv, ok);
    }
#pragma unroll
    for (int j = 0; j < NP; ++j) {
      const int q = q0 + 8 * j;
      if (4 * q < nrows) {
        float* dst = xs + (4 * q) * 32 + mm;
        dst[0] = v[j].x; dst[32] = v[j].y; dst[64] = v[j].z; dst[96] = v[j].w;
      }
    }
  }
  __syncthreads();

  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
  for (int u = 0; u < NL; ++u) {
    const int rl = 2 * u + half;
    const float a = rl < nrows ? xs[min(rl, p.rows_per_split - 1) * 32 + l31] : 0.f;
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, wb[u], acc, 0, 0, 0);
  }
  float* base = p.part + ((long)split * p.G * p.M + (long)g * p.M) * p.ldo + hd.out_off + ncol;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int mm = dz_acc_row(r, lane);
    if (mm < p.M) base[(long)mm * p.ldo] = acc[r];
  }
}

// ------------------ forward, shared weight stream across applies -------------- //
// The learner evaluates the layer for up to three applies, two of which use the
// SAME parameter set (Rainbow: online(s_tm1) and online(s_t); only the noise
// differs).  dz_fc_stream_fwd2 streams the weights once per apply and, for noisy
// layers, runs the contraction at depth 2K.  Here a workgroup streams its slice
// of one parameter SET once (mu and sigma), builds each apply's effective weight
//     W_eff[k][n] = Wmu[k][n] + Wsig[k][n] * (eps_in[k] * eps_out[n])
// in registers (2 VALU per element -- nothing next to a 64-cycle MFMA) and feeds
// one depth-K MFMA chain per apply:  y = x . W_eff  ==  x.Wmu + ((x.eps_in).Wsig).eps_out
// (networks.py:168-176) up to float32 rounding order.  Rainbow fc1: weight bytes
// 77 MB -> 51 MB (online and target once each), MFMA work halved.
// grid = (strips of 128 columns over both heads, S k-splits, parameter sets).
struct FcStreamFwd3Params {
  const float* x; int ldx; int M;      // x: [groups*M][ldx]
  int noisy;
  const float* params[2];              // parameter sets
  int ng[2];                           // applies per set (1 or 2)
  int grp[2][2];                       // their group indices
  const float* noise[DZ_MAX_GROUPS];   // noise block per group
  int G;                               // total groups (row count of a partial slab / M)
  FcHead head[2];                      // N multiple of 128
  float* part;                         // [S][G*M][ldo]
  int ldo;
  int rows_per_split;                  // multiple of 4, <= 2*NL
};

// NL = k-pairs per lane (x2 dword loads when noisy): 50 with 32 k-splits, 100 with 16.
template <int NOISY, int NL>
__global__ __launch_bounds__(256) void dz_fc_stream_fwd3(FcStreamFwd3Params p) {
  extern __shared__ __attribute__((aligned(16))) float lds3[];
  const int R = p.rows_per_split;
  float* xs = lds3;                  // [2][R][32]  x (batch-row minor)
  float* es = lds3 + 2 * R * 32;     // [2][R]      eps_in
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int half = lane >> 5, l31 = lane & 31;
  const int strips0 = p.head[0].N / 128;
  const int h_idx = blockIdx.x >= strips0 ? 1 : 0;
  const FcHead hd = dz_pick_head(p.head, h_idx);
  const int n0 = (blockIdx.x - (h_idx ? strips0 : 0)) * 128 + 32 * wave;
  const int split = blockIdx.y, set = blockIdx.z;
  const float* __restrict__ prm = set ? p.params[1] : p.params[0];
  const int ng = set ? p.ng[1] : p.ng[0];
  const int g0 = set ? p.grp[1][0] : p.grp[0][0];
  const int g1 = set ? p.grp[1][1] : p.grp[0][1];
  const float* __restrict__ nz0 = dz_pick3(p.noise, g0);
  const float* __restrict__ nz1 = dz_pick3(p.noise, g1);
  const int K = hd.K;
  const int r0 = split * R;
  const int nrows = max(min(K, r0 + R) - r0, 0);
  const int ncol = n0 + l31;

  // (1) every weight load of the wave first: they fly during the LDS staging.
  float wm[NL], wg[NOISY ? NL : 1];
#pragma unroll
  for (int u = 0; u < NL; ++u) {
    const int k = min(r0 + 2 * u + half, K - 1);
    const long off = (long)k * hd.ldw + ncol;
    wm[u] = prm[hd.w_mu + off];
    if (NOISY) wg[u] = prm[hd.w_sig + off];
  }

  // (2) stage x[g][k] (k-major, 32 batch rows minor) and eps_in[g][k].
  {
    const int mm = threadIdx.x & 31, q0 = threadIdx.x >> 5;
    const int mc = min(mm, p.M - 1);
    const float ok = mm < p.M ? 1.f : 0.f;
    constexpr int NP = (2 * NL / 4 + 7) / 8;
    float4 v[2][NP];
#pragma unroll
    for (int j = 0; j < NP; ++j) {
      const int k = min(r0 + 4 * (q0 + 8 * j), K - 4);  // r0, K multiples of 4
      v[0][j] = dz_scale4(dz_ld4(p.x + (long)(g0 * p.M + mc) * p.ldx + hd.x_off + k), ok);
      v[1][j] = dz_scale4(dz_ld4(p.x + (long)(g1 * p.M + mc) * p.ldx + hd.x_off + k), ok);
    }
    float e0 = 0.f, e1 = 0.f;
    if (NOISY) {
      const int k = min(r0 + (int)threadIdx.x, K - 1);
      e0 = nz0[hd.eps_in + k]; e1 = nz1[hd.eps_in + k];
    }
#pragma unroll
    for (int j = 0; j < NP; ++j) {
      const int q = q0 + 8 * j;
      if (4 * q < nrows) {
        float* d0 = xs + (4 * q) * 32 + mm;
        d0[0] = v[0][j].x; d0[32] = v[0][j].y; d0[64] = v[0][j].z; d0[96] = v[0][j].w;
        float* d1 = d0 + R * 32;
        d1[0] = v[1][j].x; d1[32] = v[1][j].y; d1[64] = v[1][j].z; d1[96] = v[1][j].w;
      }
    }
    if (NOISY && (int)threadIdx.x < R) { es[threadIdx.x] = e0; es[R + threadIdx.x] = e1; }
  }
  __syncthreads();

  const float eo0 = NOISY ? nz0[hd.eps_out + ncol] : 0.f;
  const float eo1 = NOISY ? nz1[hd.eps_out + ncol] : 0.f;
  f32x16 acc0, acc1;
#pragma unroll
  for (int i = 0; i < 16; ++i) { acc0[i] = 0.f; acc1[i] = 0.f; }
  if (ng > 1) {
#pragma unroll
    for (int u = 0; u < NL; ++u) {
      const int rl = 2 * u + half;
      const int rc = min(rl, R - 1);
      const bool live = rl < nrows;
      const float a0 = live ? xs[rc * 32 + l31] : 0.f;
      const float a1 = live ? xs[(R + rc) * 32 + l31] : 0.f;
      float w0 = wm[u], w1 = wm[u];
      if (NOISY) {
        w0 = __builtin_fmaf(wg[u], es[rc] * eo0, wm[u]);
        w1 = __builtin_fmaf(wg[u], es[R + rc] * eo1, wm[u]);
      }
      acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, w0, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, w1, acc1, 0, 0, 0);
    }
  } else {
#pragma unroll
    for (int u = 0; u < NL; ++u) {
      const int rl = 2 * u + half;
      const int rc = min(rl, R - 1);
      const float a0 = rl < nrows ? xs[rc * 32 + l31] : 0.f;
      float w0 = wm[u];
      if (NOISY) w0 = __builtin_fmaf(wg[u], es[rc] * eo0, wm[u]);
      acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, w0, acc0, 0, 0, 0);
    }
  }
  float* base = p.part + (long)split * p.G * p.M * p.ldo + hd.out_off + ncol;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int mm = dz_acc_row(r, lane);
    if (mm < p.M) {
      base[(long)(g0 * p.M + mm) * p.ldo] = acc0[r];
      if (ng > 1) base[(long)(g1 * p.M + mm) * p.ldo] = acc1[r];
    }
  }
}

// -------------------------------- dgrad -------------------------------------- //
// dX[m][k] = relu'(act[m][k]) * sum_{head} ( sum_n dY[m][n] Wmu[k][n]
//                                  + eps_in[k] sum_n dY[m][n] eps_out[n] Wsig[k][n] )
// One workgroup per 32 output columns k; its 4 waves take (head, mu|sigma)
// pairs round-robin (fc1: exactly one each), reduce in LDS, apply the sigma row
// scale and the ReLU mask, and write dX directly: no partial slabs, no second
// kernel.
struct FcStreamDgradParams {
  const float* dy;   // [M][ldy]
  int ldy;
  int M;             // <= 32
  int NH;
  int noisy;
  const float* params;
  const float* noise;
  FcHead head[2];    // N multiple of 8
  const float* act;  // [M][ldo] activation that produced the layer input (mask)
  float* dx;         // [M][ldo]
  int ldo;
  int K;             // output columns, multiple of 32
};

__global__ __launch_bounds__(256) void dz_fc_stream_dgrad(FcStreamDgradParams p) {
  __shared__ float red[4][16][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int half = lane >> 5, l31 = lane & 31;
  const int k0 = blockIdx.x * 32;
  const int krow = k0 + l31;
  const int m = min(l31, p.M - 1);
  const float mrow_ok = l31 < p.M ? 1.f : 0.f;

  f32x16 acc_mu, acc_sig;
#pragma unroll
  for (int i = 0; i < 16; ++i) { acc_mu[i] = 0.f; acc_sig[i] = 0.f; }

  const int parts = p.NH * (p.noisy ? 2 : 1);
  for (int part = wave; part < parts; part += 4) {
    const int h = p.noisy ? (part >> 1) : part;
    const bool sig = p.noisy && (part & 1);
    const FcHead hd = p.head[h];
    const float* __restrict__ wrow = p.params + (sig ? hd.w_sig : hd.w_mu) +
                                     (long)krow * hd.ldw + 4 * half;
    const float* __restrict__ dyrow = p.dy + (long)m * p.ldy + hd.out_off + 4 * half;
    const float* __restrict__ eorow = p.noise + hd.eps_out + 4 * half;
    const int iters = hd.N / 8;  // 8 reduction columns per iteration
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    constexpr int U = 8;
    for (int it0 = 0; it0 < iters; it0 += U) {
      float4 w[U], d[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int it = min(it0 + u, iters - 1);
        w[u] = dz_ld4(wrow + it * 8);
        float4 dv = dz_ld4(dyrow + it * 8);
        const float4 e = dz_ld4(eorow + it * 8);
        dv = sig ? dz_mul4(dv, e) : dv;
        d[u] = dz_scale4(dv, (it0 + u) < iters ? mrow_ok : 0.f);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        // rows i = k (weights as the A operand), columns j = batch row m
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w[u].x, d[u].x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w[u].y, d[u].y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w[u].z, d[u].z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w[u].w, d[u].w, acc, 0, 0, 0);
      }
    }
    if (sig) {
#pragma unroll
      for (int i = 0; i < 16; ++i) acc_sig[i] += acc[i];
    } else {
#pragma unroll
      for (int i = 0; i < 16; ++i) acc_mu[i] += acc[i];
    }
  }
  // acc element r: row (k - k0) = dz_acc_row(r, lane), column m = l31.
  // sigma contributions carry the row scale eps_in[k] (same for every head that
  // shares the input, since eps_in is per head: applied per part above would be
  // wrong for NH heads with different eps_in, so scale here per wave's head).
  {
    // each wave handled parts {wave, wave+4, ...}; with <= 4 parts it is one
    // part, whose head's eps_in applies.
    const int part = wave;
    const int h = p.noisy ? (part >> 1) : part;
    const FcHead hd = p.head[min(h, p.NH - 1)];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int kk = k0 + dz_acc_row(r, lane);
      const float e = p.noisy ? p.noise[hd.eps_in + kk] : 0.f;
      red[wave][r][lane] = acc_mu[r] + e * acc_sig[r];
    }
  }
  __syncthreads();
  if (wave > 0) return;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const float v = (red[0][r][lane] + red[1][r][lane]) + (red[2][r][lane] + red[3][r][lane]);
    const int kk = k0 + dz_acc_row(r, lane);
    if (l31 < p.M) {
      const long o = (long)l31 * p.ldo + kk;
      p.dx[o] = p.act[o] > 0.f ? v : 0.f;
    }
  }
}

}  // namespace
