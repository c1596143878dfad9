// Operand views and small kernels of the IQN learner step
// (ref: networks.py:264-292 iqn_atari_network, iqn/agent.py:176-216).
//
// IQN evaluates the value head on batch x tau-samples rows, so unlike the other
// agents its linear layers are real GEMMs (2048 x 3136 x 512 per apply at the
// reference configuration): 64x64 output tiles, full-depth contraction, and the
// layer's elementwise tail fused into the accumulator store.
#pragma once

#include "dz_qnet_kernels.h"

namespace {

enum { IQN_EPI_BIAS = 0, IQN_EPI_BIAS_RELU = 1, IQN_EPI_MIX = 2 };

// out[row][n] = epi( sum_k x[row][k] W_g[k][n] + b_g[n] )  for the rows of up to
// three groups (row ranges [row0[g], row0[g]+rows[g]) with parameters params[g]).
//   IQN_EPI_MIX (tau embedding, networks.py:281-285):
//     e = relu(.);  temb[row] = e (group 0 only);  out = e * feat[feat_row0[g] + r/samples[g]]
struct IqnLinParams {
  const float* x;
  int ldx;
  int G;
  int row0[DZ_MAX_GROUPS];
  int rows[DZ_MAX_GROUPS];
  const float* params[DZ_MAX_GROUPS];
  long w_off;
  long b_off;
  int ldw;
  int K;      // multiple of 16
  int N;
  int epi;
  float* out;
  int ldo;
  const float* feat;  // [*][N] (MIX)
  int feat_row0[DZ_MAX_GROUPS];
  int samples[DZ_MAX_GROUPS];
  float* temb;        // [rows[0]][N] or null (MIX)
};

// FULL_ = 1: every tile is whole (the rows of every group a multiple of BM, N of BN, K of BK --
// the launcher checks): the loaders carry no masks at all.
template <int WM_, int WN_, int WK_, int KT_, int MI_ = 1, int NI_ = 1, int FULL_ = 0>
struct IqnLinOp {
  static constexpr int WM = WM_, WN = WN_, WK = WK_, KT = KT_, CPS = WK_ * KT_;
  static constexpr int MI = MI_, NI = NI_;
  static constexpr int A_LAYOUT = DZ_KC, B_LAYOUT = DZ_RC, A_MAP = DZ_MAP_QUAD;
  static constexpr int BM = 32 * WM * MI, BN = 32 * WN * NI, BK = 16 * CPS;
  typedef IqnLinParams Params;
  struct Tile : DzTile { const float* prm; int row0, rows, feat_row0, samples; };

  __device__ static bool tile(const Params& p, const dim3& bid, Tile& t) {
    t.z = bid.z;
    t.m0 = bid.y * BM;
    t.n0 = bid.x * BN;
    t.st_begin = 0;
    t.st_end = (p.K / 16 + CPS - 1) / CPS;
    t.prm = dz_pick3(p.params, t.z);
    t.row0 = dz_pick3(p.row0, t.z); t.rows = dz_pick3(p.rows, t.z);
    t.feat_row0 = dz_pick3(p.feat_row0, t.z); t.samples = dz_pick3(p.samples, t.z);
    return t.z < p.G && t.m0 < t.rows && t.n0 < p.N;
  }
  __device__ static float4 load_a(const Params& p, const Tile& t, int st, int c,
                                  int row, int q) {
    // (third loader rule, dz_qnet_ops.h: a masked slot selects on the ADDRESS; a select on the
    // loaded value puts the stage's vmcnt waits in front of its MFMA block -- rounds 2-4 of this
    // Op did exactly that and ran the weight stream un-overlapped)
    if constexpr (FULL_ != 0)
      return dz_ld4(p.x + (long)(t.row0 + t.m0 + row) * p.ldx + (st * CPS + c) * 16 + 4 * q);
    const int gc = st * CPS + c, total = p.K / 16;
    const int m = t.m0 + row;
    const bool ok = (m < t.rows) & (gc < total);
    const int k = min(gc, total - 1) * 16 + 4 * q;
    const float* src = p.x + (long)(t.row0 + min(m, t.rows - 1)) * p.ldx + k;
    return dz_ld4(ok ? src : dz_page_zero);
  }
  __device__ static float4 load_b(const Params& p, const Tile& t, int st, int c,
                                  int kk, int rq) {
    if constexpr (FULL_ != 0)
      return dz_ld4(t.prm + p.w_off + (long)((st * CPS + c) * 16 + kk) * p.ldw + t.n0 + 4 * rq);
    const int gc = st * CPS + c, total = p.K / 16;
    const int k = min(gc, total - 1) * 16 + kk;
    const int n = min(t.n0 + 4 * rq, p.ldw - 4);
    const float* src = t.prm + p.w_off + (long)k * p.ldw + n;
    return dz_ld4(gc < total ? src : dz_page_zero);
  }
  __device__ static void store(const Params& p, const Tile& t, int wm, int wn,
                               int lane, const f32x16& acc) {
    const int col = t.n0 + wn * 32 + (lane & 31);
    if (col >= p.N) return;
    const int g = t.z;
    const float b = t.prm[p.b_off + col];
    // the 16 feature factors of the MIX epilogue up front (clamped rows): loaded in
    // the row loop they are 16 serial load -> wait -> store round trips
    float fm[16];
    if (p.epi == IQN_EPI_MIX) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int mc = min(t.m0 + wm * 32 + dz_acc_row(r, lane), t.rows - 1);
        fm[r] = p.feat[(long)(t.feat_row0 + mc / t.samples) * p.N + col];
      }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = t.m0 + wm * 32 + dz_acc_row(r, lane);
      if (m >= t.rows) continue;
      float v = acc[r] + b;
      if (p.epi != IQN_EPI_BIAS) v = v > 0.f ? v : 0.f;
      if (p.epi == IQN_EPI_MIX) {
        if (g == 0 && p.temb) p.temb[(long)m * p.N + col] = v;
        v = v * fm[r];
      }
      p.out[(long)(t.row0 + m) * p.ldo + col] = v;
    }
  }
};

// Weight gradient with the batch-row reduction split over the grid:
// part[split][k][n] = sum_{m in split} x[m][k] dy[m][n]   (k < K, n < ldw)
struct IqnWgradParams {
  const float* x; int ldx;
  const float* dy; int ldy;
  int M;       // reduction rows
  int K, N, ldw;
  int S;
  float* part; // [S][K][ldw]
};

template <int WM_, int WN_, int WK_, int KT_, int MI_ = 1, int NI_ = 1, int FULL_ = 0>
struct IqnWgradOp {
  static constexpr int WM = WM_, WN = WN_, WK = WK_, KT = KT_, CPS = WK_ * KT_;
  static constexpr int MI = MI_, NI = NI_;
  static constexpr int A_LAYOUT = DZ_RC, B_LAYOUT = DZ_RC, A_MAP = DZ_MAP_QUAD;
  static constexpr int BM = 32 * WM * MI, BN = 32 * WN * NI, BK = 16 * CPS;
  typedef IqnWgradParams Params;
  typedef DzTile Tile;

  __device__ static bool tile(const Params& p, const dim3& bid, Tile& t) {
    t.z = bid.z;
    t.m0 = bid.y * BM;  // k rows
    t.n0 = bid.x * BN;
    const int stages = (p.M + BK - 1) / BK;
    const int per = (stages + p.S - 1) / p.S;
    t.st_begin = t.z * per;
    t.st_end = min(stages, t.st_begin + per);
    return t.m0 < p.K && t.n0 < p.N;
  }
  __device__ static float4 load_a(const Params& p, const Tile& t, int st, int c,
                                  int kk, int rq) {
    if constexpr (FULL_ != 0)   // (whole tiles, M a multiple of BK: no masks)
      return dz_ld4(p.x + (long)(st * BK + c * 16 + kk) * p.ldx + t.m0 + 4 * rq);
    const int m = st * BK + c * 16 + kk;
    const int k = min(t.m0 + 4 * rq, p.K - 4);
    const float* src = p.x + (long)min(m, p.M - 1) * p.ldx + k;
    return dz_ld4(m < p.M ? src : dz_page_zero);
  }
  __device__ static float4 load_b(const Params& p, const Tile& t, int st, int c,
                                  int kk, int rq) {
    if constexpr (FULL_ != 0)
      return dz_ld4(p.dy + (long)(st * BK + c * 16 + kk) * p.ldy + t.n0 + 4 * rq);
    const int m = st * BK + c * 16 + kk;
    const int n = min(t.n0 + 4 * rq, p.ldw - 4);
    const float* src = p.dy + (long)min(m, p.M - 1) * p.ldy + n;
    return dz_ld4(m < p.M ? src : dz_page_zero);
  }
  __device__ static void store(const Params& p, const Tile& t, int wm, int wn,
                               int lane, const f32x16& acc) {
    const int col = t.n0 + wn * 32 + (lane & 31);
    if (col >= p.ldw) return;
    float* base = p.part + (long)t.z * p.K * p.ldw + col;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int k = t.m0 + wm * 32 + dz_acc_row(r, lane);
      if (k < p.K) base[(long)k * p.ldw] = col < p.N ? acc[r] : 0.f;
    }
  }
};

// Input gradient of a plain linear layer:  dx[m][k] = relu'(.) sum_n dy[m][n] W[k][n]  (n < N,
// N a multiple of 4: no float4 straddles the edge).  The general FcDgradOp (noisy layers,
// two heads) scales its operands in the loaders, which waits for every load in front of the
// MFMA block (third loader rule); here both operands are plain float4 loads.
struct IqnDgradParams {
  const float* dy; int ldy;     // [M][ldy]
  const float* w; int ldw;      // [K][ldw]
  int M, N, K;
  float* dx; int ldo;           // [M][ldo]
  const float* relu_mask;       // [M][ldo] or null: dx *= (mask > 0)
};
template <int WM_, int WN_, int WK_, int KT_, int FULL_ = 0>
struct IqnDgradOp {
  static constexpr int WM = WM_, WN = WN_, WK = WK_, KT = KT_, CPS = WK_ * KT_;
  static constexpr int A_LAYOUT = DZ_KC, B_LAYOUT = DZ_KC, A_MAP = DZ_MAP_QUAD;
  static constexpr int BM = 32 * WM, BN = 32 * WN, BK = 16 * CPS;
  typedef IqnDgradParams Params;
  typedef DzTile Tile;
  __device__ static bool tile(const Params& p, const dim3& bid, Tile& t) {
    t.z = 0; t.m0 = bid.y * BM; t.n0 = bid.x * BN;
    t.st_begin = 0; t.st_end = (p.N + BK - 1) / BK;
    return t.m0 < p.M && t.n0 < p.K;
  }
  // A tile row = batch row m, 4 consecutive reduction indices n
  __device__ static float4 load_a(const Params& p, const Tile& t, int st, int c, int row, int q) {
    if constexpr (FULL_ != 0)
      return dz_ld4(p.dy + (long)(t.m0 + row) * p.ldy + (st * CPS + c) * 16 + 4 * q);
    const int m = t.m0 + row, n = (st * CPS + c) * 16 + 4 * q;
    const float* src = p.dy + (long)min(m, p.M - 1) * p.ldy + min(n, p.N - 4);
    return dz_ld4(((m < p.M) & (n < p.N)) ? src : dz_page_zero);
  }
  // B tile row = output column k, 4 consecutive reduction indices n
  __device__ static float4 load_b(const Params& p, const Tile& t, int st, int c, int row, int q) {
    if constexpr (FULL_ != 0)
      return dz_ld4(p.w + (long)(t.n0 + row) * p.ldw + (st * CPS + c) * 16 + 4 * q);
    const int k = min(t.n0 + row, p.K - 1), n = (st * CPS + c) * 16 + 4 * q;
    const float* src = p.w + (long)k * p.ldw + min(n, p.N - 4);
    return dz_ld4(n < p.N ? src : dz_page_zero);
  }
  __device__ static void store(const Params& p, const Tile& t, int wm, int wn, int lane,
                               const f32x16& acc) {
    const int col = t.n0 + wn * 32 + (lane & 31);
    if (col >= p.K) return;
    float mv[16];
    if (p.relu_mask) {  // (uniform) all 16 mask values first, then the stores
#pragma unroll
      for (int r = 0; r < 16; ++r)
        mv[r] = p.relu_mask[(long)min(t.m0 + wm * 32 + dz_acc_row(r, lane), p.M - 1) * p.ldo + col];
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = t.m0 + wm * 32 + dz_acc_row(r, lane);
      if (m < p.M) p.dx[(long)m * p.ldo + col] = (!p.relu_mask || mv[r] > 0.f) ? acc[r] : 0.f;
    }
  }
};

using IqnLin = IqnLinOp<2, 2, 1, 2>;
using IqnWg = IqnWgradOp<2, 2, 1, 2>;
using IqnDg = FcDgradOp<2, 2, 1, 2>;

// cosemb[row][i] = cos(pi_i * tau[row]),  pi_i = float32(i+1) * float32(pi)
// (networks.py:277-278; both products in float32 as in the reference).
struct IqnCosParams {
  const float* t0; const float* t1; const float* t2;
  int n0, n1, n2, latent;
  float* out;
};
__device__ __forceinline__ void iqn_cos_at(const IqnCosParams& q, long i) {
  const long total = (long)(q.n0 + q.n1 + q.n2) * q.latent;
  if (i >= total) return;
  const int row = (int)(i / q.latent), l = (int)(i % q.latent);
  const float tau = row < q.n0 ? q.t0[row]
                               : (row < q.n0 + q.n1 ? q.t1[row - q.n0] : q.t2[row - q.n0 - q.n1]);
  const float pm = (float)(l + 1) * 3.14159274101257324f;
  q.out[i] = cosf(pm * tau);
}
// ... as extra workgroups of the step's conv1 launch (torso_forward_side): the taus exist
// before the step starts and conv1 does not read the table
struct IqnCosSide {
  typedef IqnCosParams Params;
  __device__ static void run(const Params& q, unsigned block) {
    iqn_cos_at(q, (long)block * 256 + threadIdx.x);
  }
  static unsigned blocks(const Params& q) {
    return (unsigned)(((long)(q.n0 + q.n1 + q.n2) * q.latent + 255) / 256);
  }
};

// U[0,1) samples for the tau draws (iqn/agent.py:47-51 jax.random.uniform: 23
// random mantissa bits), counter-based so a step's draws depend only on (seed,
// step, index).
__global__ void uniform_fill_kernel(float* __restrict__ out, long n, uint64_t seed,
                                    uint64_t counter, const int32_t* __restrict__ step) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (step) counter += (uint64_t)(*step) * (uint64_t)n;
  const uint64_t h = mix64(mix64(seed) ^ mix64(counter + (uint64_t)i));
  out[i] = (float)(h >> 41) * (1.0f / 8388608.0f);
}

// mean over the n rows of column a, one wave: lane i sums rows i, i+64, ... in
// that order (8 clamped loads in flight per round), then the wave folds.
__device__ __forceinline__ float iqn_col_mean(const float* __restrict__ o, int ld, int n, int a,
                                              int lane) {
  float sum = 0.f;
  for (int base = 0; base < n; base += 8 * 64) {
    float x[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int r = base + 64 * j + lane;
      const float y = o[(long)min(r, n - 1) * ld + a];
      x[j] = r < n ? y : 0.f;
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) sum += x[j];
  }
  return wave_sum(sum) / (float)n;
}

// vmap(rlax.quantile_q_learning) over the batch (iqn/agent.py:199-209).  One block
// per batch element; rows of `out` are (batch element, sample) pairs.
//   a* = argmax_a mean_n out_sel[b][n][a];  target_j = r + g * out_t[b][j][a*]
//   loss_b = sum_i mean_j |tau_i - 1{delta_ij < 0}| huber_kappa(delta_ij),
//   delta_ij = target_j - out_0[b][i][a_tm1];  dout = d(mean_b loss_b)/d out_0
__global__ __launch_bounds__(256) void iqn_loss_kernel(
    const float* __restrict__ out, int ld, int B, int A, int n0, int n1, int n2,
    const float* __restrict__ tau0, const int64_t* __restrict__ a_tm1,
    const double* __restrict__ r_t, const double* __restrict__ d_t, float kappa,
    float* __restrict__ dout, float* __restrict__ losses) {
  __shared__ float s_t[256];
  __shared__ float s_red[4];
  __shared__ int s_astar;
  const int b = blockIdx.x, i = threadIdx.x;
  const float* o0 = out + (long)b * n0 * ld;
  const float* o1 = out + ((long)B * n0 + (long)b * n1) * ld;
  const float* o2 = out + ((long)B * (n0 + n1) + (long)b * n2) * ld;
  if (i < 64) {
    float best = -__builtin_inff();
    int arg = 0;
    for (int a = 0; a < A; ++a) {
      const float sum = iqn_col_mean(o1, ld, n1, a, i);
      if (sum > best) { best = sum; arg = a; }
    }
    if (i == 0) s_astar = arg;
  }
  __syncthreads();
  const int a_star = s_astar, a0 = (int)a_tm1[b];
  const float r = (float)r_t[b], g = (float)d_t[b];
  if (i < n2) s_t[i] = r + g * o2[(long)i * ld + a_star];
  __syncthreads();
  float li = 0.f, gi = 0.f;
  if (i < n0) {
    const float theta = o0[(long)i * ld + a0], ti = tau0[b * n0 + i];
    for (int j = 0; j < n2; ++j) {
      const float delta = s_t[j] - theta;
      const float wgt = fabsf(ti - (delta < 0.f ? 1.f : 0.f));
      const float ad = fabsf(delta);
      float hub, dh;
      if (kappa > 0.f) {
        const float q = fminf(ad, kappa);
        hub = 0.5f * q * q + kappa * (ad - q);
        dh = ad <= kappa ? delta : (delta > 0.f ? kappa : -kappa);
      } else {
        hub = ad;
        dh = delta > 0.f ? 1.f : (delta < 0.f ? -1.f : 0.f);
      }
      li += wgt * hub;
      gi += wgt * dh;
    }
    li /= (float)n2;
    gi = -gi / (float)n2;
  }
  const float s = wave_sum(li);
  if ((i & 63) == 0) s_red[i >> 6] = s;
  __syncthreads();
  if (i == 0) losses[b] = (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]);
  if (i < n0) {
    float* d = dout + ((long)b * n0 + i) * ld;
    for (int a = 0; a < ld; ++a) d[a] = (a == a0) ? gi / (float)B : 0.f;
  }
}

// Backward of head_in = temb * feat[b] (networks.py:285) for the online apply, from head_in
// itself (the forward pass does not store temb: 26 MB less written and kept):
//   dzt[row][c]  = dhin[row][c] * feat[b][c] * (temb[row][c] > 0)     (in place)
//   dfeat[b][c]  = (feat[b][c] > 0) * sum_n dhin[b*N+n][c] * temb[b*N+n][c]
// With f = feat[b][c] > 0: temb > 0 <=> head_in = temb f > 0, and sum_n dhin temb =
// (sum_n dhin head_in) / f; with f == 0 both results are 0 whatever temb was.
//   bias_part[b][c] = sum_n dzt[b*N+n][c]   (the embedding bias gradient's partial of batch
//                     element b: B slabs folded by reduce_jobs_kernel -- the 26 MB this pass
//                     has in registers anyway, instead of a second pass over it)
__global__ __launch_bounds__(256) void iqn_mix_bwd_kernel(float* __restrict__ dhin,
                                                          const float* __restrict__ hin,
                                                          const float* __restrict__ feat,
                                                          int B, int samples, int F,
                                                          float* __restrict__ dfeat,
                                                          float* __restrict__ bias_part) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  const int b = blockIdx.y;
  if (c >= F) return;
  const float f = feat[(long)b * F + c];
  float acc = 0.f, bsum = 0.f;
  const long o0 = (long)b * samples * F + c;
  // 8 rows per round, all 16 loads first: dhin is updated in place, so the compiler
  // cannot move a later row's load above an earlier row's store by itself
  for (int n0 = 0; n0 < samples; n0 += 8) {
    float d[8], e[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const long o = o0 + (long)min(n0 + j, samples - 1) * F;
      d[j] = dhin[o]; e[j] = hin[o];
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      if (n0 + j < samples) {
        acc += d[j] * e[j];
        const float dz = e[j] > 0.f ? d[j] * f : 0.f;
        dhin[o0 + (long)(n0 + j) * F] = dz;
        bsum += dz;
      }
    }
  }
  dfeat[(long)b * F + c] = f > 0.f ? acc / f : 0.f;
  bias_part[(long)b * F + c] = bsum;
}

// part[split][c] = sum of rows [split*rps, (split+1)*rps) of m[.][c]; a set of
// matrices per launch (blockIdx.z).  The S partial rows are folded by
// reduce_jobs_kernel.
struct ColPartJob { const float* m; int rows; int cols; int ld; float* part; };
struct ColPartJobs { ColPartJob j[2]; unsigned end0; int S; };  // blocks [0, end0) = job 0
__device__ __forceinline__ void colsum_part_block(const ColPartJob& jb, int S, unsigned bx,
                                                  unsigned by) {
  __shared__ float red[4][64];
  const int l = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int c = bx * 64 + l;
  const int rps = (jb.rows + S - 1) / S;
  const int r0 = by * rps, r1 = min(jb.rows, r0 + rps);
  const float v = r1 > r0 ? dz_slab_sum(jb.m + (long)r0 * jb.ld, r1 - r0, jb.ld,
                                        min(c, jb.cols - 1), w) : 0.f;
  red[w][l] = v;
  __syncthreads();
  if (w == 0 && c < jb.cols)
    jb.part[(long)by * jb.cols + c] = (red[0][l] + red[1][l]) + (red[2][l] + red[3][l]);
}
// ... as extra workgroups of the embedding weight-gradient launch (it reads neither result)
struct ColsumSide {
  typedef ColPartJobs Params;
  __device__ static void run(const Params& q, unsigned block) {
    const bool first = block < q.end0;
    const ColPartJob jb = first ? q.j[0] : q.j[1];
    const unsigned i = first ? block : block - q.end0;
    colsum_part_block(jb, q.S, i / (unsigned)q.S, i % (unsigned)q.S);
  }
  static unsigned blocks_of(const ColPartJob& jb, int S) { return (unsigned)((jb.cols + 63) / 64) * S; }
};

// dfeat[b][c] = (feat[b][c] > 0) * (sum of the samples/32 row-block partials s1) / feat[b][c]
// (IqnDgradMixEpi, dz_iqn_fc1_dma.h), one thread per (b, c).
struct DfeatJob { const float* s1; const float* feat; float* dfeat; int B, F, blocks_per_b; };
// The two column sums (ColsumSide) and that fold as ONE side job of the embedding
// weight-gradient launch.
struct IqnBwdSideParams { ColPartJobs col; unsigned col_blocks; DfeatJob df; };
struct IqnBwdSide {
  typedef IqnBwdSideParams Params;
  __device__ static void run(const Params& q, unsigned block) {
    if (block < q.col_blocks) { ColsumSide::run(q.col, block); return; }
    const long i = (long)(block - q.col_blocks) * 256 + threadIdx.x;
    const DfeatJob& d = q.df;
    if (i >= (long)d.B * d.F) return;
    const int b = (int)(i / d.F), c = (int)(i % d.F);
    float s = 0.f;
    for (int j = 0; j < d.blocks_per_b; ++j) s += d.s1[(long)(b * d.blocks_per_b + j) * d.F + c];
    const float f = d.feat[i];
    d.dfeat[i] = f > 0.f ? s / f : 0.f;
  }
};

// out[i] = sum_s part[s][i] for up to 8 jobs in one launch.
// `bump_count` (optional): optax's `count_inc = count + 1` for the Adam launch that follows,
// when no norm launch (sumsq_kernel) is there to do it.
struct ReduceJobs8 { ReduceJob r[8]; unsigned r_end[8]; int n; int32_t* bump_count = nullptr; };
__global__ __launch_bounds__(256) void reduce_jobs_kernel(ReduceJobs8 J) {
  __shared__ float red[4][64];
  const int l = threadIdx.x & 63, w = threadIdx.x >> 6;
  const unsigned b = blockIdx.x;
  if (J.bump_count && b == 0 && threadIdx.x == 0) *J.bump_count = *J.bump_count + 1;
  int j = 0;
  while (j < J.n - 1 && b >= J.r_end[j]) ++j;
  const ReduceJob jb = J.r[j];
  const long i = (long)(b - (j ? J.r_end[j - 1] : 0)) * 64 + l;
  const float v = dz_slab_sum(jb.part, jb.S, jb.n, i < jb.n ? i : jb.n - 1, w);
  red[w][l] = v;
  __syncthreads();
  if (w == 0 && i < jb.n) jb.out[i] = (red[0][l] + red[1][l]) + (red[2][l] + red[3][l]);
}

// q_values[b][a] = mean_n q_dist[b][n][a] (networks.py:288), greedy action (first
// maximum) and its value.
__global__ __launch_bounds__(64) void iqn_q_values_kernel(const float* __restrict__ out,
                                                          int ld, int A, int samples,
                                                          float* __restrict__ q_out,
                                                          int32_t* __restrict__ greedy_out,
                                                          float* __restrict__ vmax_out) {
  const int b = blockIdx.x, i = threadIdx.x;
  const float* o = out + (long)b * samples * ld;
  float best = -__builtin_inff();
  int arg = 0;
  for (int a = 0; a < A; ++a) {
    const float sum = iqn_col_mean(o, ld, samples, a, i);
    if (i == 0 && q_out) q_out[b * A + a] = sum;
    if (sum > best) { best = sum; arg = a; }
  }
  if (i == 0) {
    if (greedy_out) greedy_out[b] = arg;
    if (vmax_out) vmax_out[b] = best;
  }
}

}  // namespace
