// Kernels and tile configurations shared by the learner steps of every agent
// (dz_rainbow.hip: noisy dueling C51; dz_dense.hip: DQN / double-Q / prioritized /
// C51 / QR dense heads): split-K epilogues, partial reductions, gradient
// finalisation, global norm, optimisers, noise, loss heads.
#pragma once

#include "dz_fc_stream.h"
#include "dz_gram.h"
#include "dz_seam.h"
#include "dz_sumtree_dev.h"

namespace {

constexpr int kFlat = 3136;   // 7*7*64 torso features
constexpr int kHid = 512;
constexpr int kG = 3;         // applies: online(s_tm1), online(s_t), target(s_t)
constexpr int kS_fc1 = 7;     // grid split-K factors
constexpr int kS_fc2 = 4;
constexpr int kS_dh1 = 5;
constexpr int kS_dfeat = 8;
constexpr int kMaxS_dfeat = 32;
constexpr int kMaxS_fc2 = 8;
// conv weight-gradient split-K counts.  kS_cw3 = 9 makes conv3's backward launch 90 + 162 = 252
// workgroups: one round on 256 CUs (10 splits = 262 workgroups: 15.6 us; 9: 13.0; 8: 14.3 by events)
#ifndef DZ_S_CW1
#define DZ_S_CW1 50
#endif
#ifndef DZ_S_CW2
#define DZ_S_CW2 27
#endif
constexpr int kS_cw1 = DZ_S_CW1, kS_cw2 = DZ_S_CW2, kS_cw3 = 9;   // (re-swept in round 4: 25/34/40/67 x 14/18/22/36 all slower)
constexpr int kNormBlocks = 512;
constexpr int kNormFinal = 4096;    // fused-norm partials: one per finalize block
constexpr int kNormSlots = 12288;   // per-wave slots of the weight-gradient kernels

inline int64_t align4(int64_t v) { return (v + 3) & ~(int64_t)3; }
constexpr int kMaxSplitFc1 = 32;
// Rainbow's fc1 forward k-splits (dz_fc_stream.h): 32 x 100 rows, or 16 x 196 (A/B switch)
#ifndef DZ_FC1_SPLITS
#define DZ_FC1_SPLITS 32
#endif
constexpr int kFc1Splits = DZ_FC1_SPLITS;
static_assert(kFc1Splits % 4 == 0 && kFc1Splits <= kMaxSplitFc1, "four waves fold the slabs");

// conv geometries (networks.py:194-198)
//                      U8  H   W   C  KS S  OH  OW  CO
//                                                       WM WN WK KT

// (conv1: 1 200 workgroups of 32 rows, K over the four waves -- 12.6 vs 13.1 us for <2,1,2,2>
// once its loader stopped waiting; round 4 re-sweep of six shapes per layer, tools/ab.sh libs)
using Conv1Fwd = ConvFwdOp<1, 84, 84, 4, 8, 4, 20, 20, 32, 1, 1, 4, 1>;
using Conv2Fwd = ConvFwdOp<0, 20, 20, 32, 4, 2, 9, 9, 64, 1, 1, 4, 2>;
using Conv3Fwd = ConvFwdOp<0, 9, 9, 64, 3, 1, 7, 7, 64, 1, 1, 4, 3, 0>;
// acting (a handful of images: 7-13 workgroups, pure latency): conv1's whole K = 256 in ONE
// stage instead of four (one memory round trip): 10.3 -> 6.8 us at batch 1.  (conv2 with 2 or 1
// stages: 7.9 / 8.2 vs 8.15 us; conv3 with 2: 11.7 vs 8.3 -- left on the learner's shapes.)
using Conv1FwdAct = ConvFwdOp<1, 84, 84, 4, 8, 4, 20, 20, 32, 1, 1, 4, 4>;
using Conv1Wg = ConvWgradOp<1, 84, 84, 4, 8, 4, 20, 20, 32, 2, 1, 2, 2>;
using Conv2Wg = ConvWgradOp<0, 20, 20, 32, 4, 2, 9, 9, 64, 2, 2, 1, 2, 0>;
using Conv3Wg = ConvWgradOp<0, 9, 9, 64, 3, 1, 7, 7, 64, 2, 2, 1, 2>;
using Conv2Dg = ConvDgradOp<20, 20, 32, 4, 2, 9, 9, 64, 1, 1, 4, 1, 0>;
using Conv3Dg = ConvDgradOp<9, 9, 64, 3, 1, 7, 7, 64, 1, 1, 4, 3>;
using FcFwd = FcFwdOp<1, 2, 2, 4>;
using FcDg = FcDgradOp<1, 2, 2, 4>;
using FcWg = FcWgradOp<2, 2, 1, 2>;

// ---- small kernels ----------------------------------------------------------

// One lane's strided slab sum  src[w*stride + off] + src[(w+4)*stride + off] + ...
// (s < S), added in that order.  A plain `for (s = w; s < S; s += 4) v += ...`
// compiles to one load + one wait per trip (the trip count differs per wave), i.e.
// S/4 serial L2 round trips; here 8 clamped loads are in flight per round and the
// out-of-range slots add +0.
__device__ __forceinline__ float dz_slab_sum(const float* __restrict__ src, int S, long stride,
                                             long off, int w) {
  float v = 0.f;
  for (int s0 = 0; s0 < S; s0 += 32) {
    float x[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int s = s0 + w + 4 * j;
      const float y = src[(long)min(s, S - 1) * stride + off];
      x[j] = s < S ? y : 0.f;
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) v += x[j];
  }
  return v;
}

// out[r][c] = act( sum_s part[s][r][c] + b_mu[c] + b_sig[c]*eps_out[g][c] )
// block = 64 columns x 4 waves striding over the split slabs (LDS combine).
__global__ __launch_bounds__(256) void fc_epilogue_kernel(
    const float* __restrict__ part, int S, int rows, int cols, int ld,
    int rows_per_group, const float* p0, const float* p1, const float* p2, long b_mu,
    long b_sig, const float* n0, const float* n1, const float* n2, int eps_out, int relu,
    float* __restrict__ out, int bias_shared = 0) {
  __shared__ float red[4][64];
  const int l = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + l;
  const int r = blockIdx.y;
  const int cc = min(c, cols - 1);
  // bias operands first: in flight together with the slab loads
  const int g = r / rows_per_group;
  const float* prm = g == 0 ? p0 : (g == 1 ? p1 : p2);
  const float* nz = g == 0 ? n0 : (g == 1 ? n1 : n2);
  float bm = 0.f, bs = 0.f, be = 0.f;
  if (b_mu >= 0) bm = prm[b_mu + (bias_shared ? 0 : cc)];  // networks.py:120-134
  if (b_sig >= 0) { bs = prm[b_sig + cc]; be = nz[eps_out + cc]; }
  float v = dz_slab_sum(part, S, (long)rows * ld, (long)r * ld + cc, w);
  red[w][l] = v;
  __syncthreads();
  if (w != 0 || c >= cols) return;
  v = (red[0][l] + red[1][l]) + (red[2][l] + red[3][l]);
  if (b_mu >= 0) v += bm;
  if (b_sig >= 0) v += bs * be;
  if (relu) v = v > 0.f ? v : 0.f;
  out[(long)r * ld + c] = v;
}

// out[i] = (mask? mask[i] > 0 : 1) * sum_s part[s][i].  64 outputs per block;
// the 4 waves stride over the S partial slabs and combine through LDS, so the
// dependent-add chain is S/4 long and every load is a coalesced 256-byte row.
__global__ __launch_bounds__(256) void reduce_parts_kernel(const float* part, int S,
                                                           long n, const float* mask,
                                                           float* out) {
  __shared__ float red[4][64];
  const int l = threadIdx.x & 63, w = threadIdx.x >> 6;
  const long i = (long)blockIdx.x * 64 + l;
  const long ic = i < n ? i : n - 1;
  // unconditional (pointer select, no branch + wait): in flight with the slab loads
  const float mk = *(mask ? mask + ic : part);
  float v = dz_slab_sum(part, S, n, ic, w);
  red[w][l] = v;
  __syncthreads();
  if (w == 0 && i < n) {
    v = (red[0][l] + red[1][l]) + (red[2][l] + red[3][l]);
    if (mask && !(mk > 0.f)) v = 0.f;
    out[i] = v;
  }
}

// Column sums of the (short) linear-layer output gradients: out[c] = sum_r m[r][c],
// out_scaled[c] = out[c] * scale[c] (the sigma-bias gradient).  Convolution
// bias gradients come out of the wgrad GEMM (ConvWgradOp's extra row).
struct ColsumJob {
  const float* m; int rows; int cols; int ld; float* out; const float* scale;
  float* out_scaled;
};
struct ColsumJobs { ColsumJob j[4]; int n; };
__global__ __launch_bounds__(256) void colsum_kernel(ColsumJobs jobs) {
  __shared__ float red[4][64];
  const ColsumJob jb = jobs.j[blockIdx.y];
  const int l = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + l;
  if (blockIdx.x * 64 >= jb.cols) return;
  const float v = dz_slab_sum(jb.m, jb.rows, jb.ld, min(c, jb.cols - 1), w);
  red[w][l] = v;
  __syncthreads();
  if (w == 0 && c < jb.cols) {
    const float s = (red[0][l] + red[1][l]) + (red[2][l] + red[3][l]);
    if (jb.out) jb.out[c] = s;
    if (jb.out_scaled) jb.out_scaled[c] = s * jb.scale[c];
  }
}

// Gradient finalisation in ONE launch: the split-K partial slabs of the three
// convolution weight(+bias) gradients are reduced into the gradient buffer and
// the linear-layer bias gradients (column sums) are formed.  blockIdx.x ranges:
// [0, t0) reduce job 0, [t0, t1) job 1, [t1, t2) job 2, then the colsum tiles.
struct ReduceJob { const float* part; int S; long n; float* out; };
// Optional optimiser riding in the finalize launch (dense learners with RMSProp, which
// needs no global norm): every gradient value a finalize block produces is applied to
// its parameter on the spot, and `flat_blocks` extra blocks run the flat update over the
// (at most two) parameter ranges whose gradients were stored by GEMM launches.
// optax.rmsprop(lr, decay, eps, centered=True) + apply_updates (dqn/run_atari.py:205-210):
// eps INSIDE the sqrt, no bias correction -- the arithmetic of rmsprop_kernel.
struct RmsApply {
  float* p = nullptr; float* mu = nullptr; float* nu = nullptr;
  const float* grad = nullptr;           // gradient buffer base (parameter index = g - grad)
  float lr = 0.f, decay = 0.f, eps = 0.f;
  long lo4[2] = {0, 0}, n4[2] = {0, 0};  // float4 ranges of the flat part
  unsigned flat_blocks = 0;
};
__device__ __forceinline__ void dz_rms_one(float g, float& m, float& v, float& p, float lr,
                                           float decay, float eps) {
  m = (1.0f - decay) * g + decay * m;
  v = (1.0f - decay) * (g * g) + decay * v;
  const float upd = g / sqrtf(v - m * m + eps);
  p = p + (-lr) * upd;
}
__device__ __forceinline__ void dz_rms_apply_at(const RmsApply& R, const float* gptr, float g) {
  const long idx = gptr - R.grad;
  float m = R.mu[idx], v = R.nu[idx], pp = R.p[idx];
  dz_rms_one(g, m, v, pp, R.lr, R.decay, R.eps);
  R.mu[idx] = m; R.nu[idx] = v; R.p[idx] = pp;
}
// The dense learners' wide layer (fc1: 3136 x 512) WITHOUT a stored weight gradient, as
// Rainbow's (dz_fc1_onfly.h): G = X^T D has rank <= 32, its factors (X: 400 KB, D = dh1:
// 64 KB) are L2-resident, the matrix is 6.4 MB -- written by 392 MFMA workgroups of the
// backward launch and read back here.  A workgroup owns a 16-row x 64-column tile of the
// matrix and its two RMSProp moments (one float4 of each per thread, requested before anything
// else), keeps the 32 x 64 strip of dh1 and the 16 x 32 tile of X in LDS and forms every
// gradient element (32 FMAs, batch ascending) right before its update.  (Measured, same box:
// 64-row tiles with one row group of loads ahead 10.52 k steps/s on BASELINE config 2; all four
// row groups' loads up front 10.31 k -- 124 VGPRs for every role of the launch; 32-row tiles
// 10.6 k; 16-row tiles, 1568 workgroups 10.98 k.)  RMSProp has no global norm: nothing else is needed (the Adam dense learners keep
// the stored form).  ref: dqn/agent.py:109-117, dqn/run_atari.py:205-210.
struct RmsOnFly {
  const float* feat = nullptr;   // [B][3136] input of the layer (online s_tm1 apply); nullptr: off
  const float* dh1 = nullptr;    // [B][512]  d loss / d pre-activation, finished
  int B = 0;
  long w = 0; int ld = 0;        // the [3136][ld] matrix in the parameter vector (512 columns used)
  unsigned blocks = 0;
};
constexpr int kRofC = 64, kRofIT = 1, kRofRP = 256 / (kRofC / 4), kRofR = kRofRP * kRofIT, kRofFS = 36;
constexpr int kRofStrips = 512 / kRofC, kRofBlocks = (3136 / kRofR) * kRofStrips;   // 49 x 8
constexpr int kRofLds = 32 * kRofC + kRofR * kRofFS;
static_assert(3136 % kRofR == 0, "rows");
__device__ __forceinline__ void rms_fc1_block(unsigned blk, const RmsOnFly& q, const RmsApply& R,
                                              float* lds) {
  float* s_dh1 = lds; float* s_ft = lds + 32 * kRofC;
  const int strip = blk % kRofStrips, rg = blk / kRofStrips;
  const int k0 = rg * kRofR, c0 = strip * kRofC;
  const int tid = threadIdx.x, rl = tid / (kRofC / 4), c4 = tid % (kRofC / 4);
  const long o0 = (q.w + (long)(k0 + rl) * q.ld + c0 + 4 * c4) >> 2;      // float4 index
  const long rstep = ((long)kRofRP * q.ld) >> 2;
  // every parameter / moment load of the tile is in flight before the factors are staged (one
  // iteration ahead left 48 bytes per thread in flight: 15.8 us for 51 MB)
  float4 pv[kRofIT], mv[kRofIT], vv[kRofIT];
#pragma unroll
  for (int it = 0; it < kRofIT; ++it) {
    const long o = o0 + it * rstep;
    pv[it] = ((const float4*)R.p)[o]; mv[it] = ((const float4*)R.mu)[o]; vv[it] = ((const float4*)R.nu)[o];
  }
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int i = 0; i < kRofC / 32; ++i) {       // dh1 strip [32][C]
    const int e = tid + 256 * i, b = e / (kRofC / 4), cc = e % (kRofC / 4);
    const float4 d = *(const float4*)(q.dh1 + (unsigned)(min(b, q.B - 1) * 512 + c0 + 4 * cc));
    *(float4*)(s_dh1 + b * kRofC + 4 * cc) = b < q.B ? d : dz_f4zero();
  }
#pragma unroll
  for (int i = 0; i < (32 * kRofR + 255) / 256; ++i) {   // X tile [R][32 (+4)]
    const int e = tid + 256 * i, b = e / kRofR, r = e % kRofR;
    if (e < 32 * kRofR) {
      const float f = q.feat[(unsigned)(min(b, q.B - 1) * 3136 + k0 + r)];
      s_ft[r * kRofFS + b] = b < q.B ? f : 0.f;
    }
  }
  __syncthreads();
#pragma unroll
  for (int it = 0; it < kRofIT; ++it) {
    const float* ft = s_ft + (it * kRofRP + rl) * kRofFS;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll 2
    for (int bq = 0; bq < 8; ++bq) {           // G[k][n] = sum_b x[b][k] dh1[b][n], b ascending
      const float4 f = *(const float4*)(ft + 4 * bq);
      const float fx[4] = {f.x, f.y, f.z, f.w};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float4 d = *(const float4*)(s_dh1 + (4 * bq + j) * kRofC + 4 * c4);
        a0 = __builtin_fmaf(fx[j], d.x, a0); a1 = __builtin_fmaf(fx[j], d.y, a1);
        a2 = __builtin_fmaf(fx[j], d.z, a2); a3 = __builtin_fmaf(fx[j], d.w, a3);
      }
    }
    const float G[4] = {a0, a1, a2, a3};
    float* P = (float*)&pv[it]; float* M = (float*)&mv[it]; float* V = (float*)&vv[it];
#pragma unroll
    for (int j = 0; j < 4; ++j) dz_rms_one(G[j], M[j], V[j], P[j], R.lr, R.decay, R.eps);
    const long o = o0 + it * rstep;
    ((float4*)R.mu)[o] = mv[it]; ((float4*)R.nu)[o] = vv[it]; ((float4*)R.p)[o] = pv[it];
  }
}
__device__ __forceinline__ void dz_rms_flat(const RmsApply& R, unsigned fb) {
  // software-pipelined like rmsprop_kernel: the next element's loads (clamped,
  // unconditional) are issued before the current element's arithmetic
  const long total = R.n4[0] + R.n4[1];
  const long stride = (long)R.flat_blocks * 256;
  auto at = [&](long i) { return i < R.n4[0] ? R.lo4[0] + i : R.lo4[1] + (i - R.n4[0]); };
  long i = (long)fb * 256 + threadIdx.x;
  long ic = at(min(i, total - 1));
  float4 gv = ((const float4*)R.grad)[ic], mv = ((const float4*)R.mu)[ic];
  float4 vv = ((const float4*)R.nu)[ic], pv = ((const float4*)R.p)[ic];
  while (i < total) {
    const long inext = i + stride;
    const long cur = ic;
    ic = at(min(inext, total - 1));
    const float4 gn = ((const float4*)R.grad)[ic], mn = ((const float4*)R.mu)[ic];
    const float4 vn = ((const float4*)R.nu)[ic], pn = ((const float4*)R.p)[ic];
    float* G = (float*)&gv; float* M = (float*)&mv; float* V = (float*)&vv;
    float* P = (float*)&pv;
#pragma unroll
    for (int j = 0; j < 4; ++j) dz_rms_one(G[j], M[j], V[j], P[j], R.lr, R.decay, R.eps);
    ((float4*)R.mu)[cur] = mv; ((float4*)R.nu)[cur] = vv; ((float4*)R.p)[cur] = pv;
    gv = gn; mv = mn; vv = vn; pv = pn; i = inext;
  }
}
struct FinalizeJobs {
  ReduceJob r[3];
  unsigned r_end[3];      // exclusive prefix of 64-wide tiles
  ColsumJob c[2];
  unsigned c_tiles[2];    // tiles per colsum job
  // Optional fused global norm (see FcWgradParams::sumsq): every block writes
  // the sum of squares of the gradient entries it produced to sumsq[blockIdx.x];
  // `presum_blocks` extra blocks fold the weight-gradient kernels' per-wave slots
  // (presum_src[0..presum_n), 1024 per block) the same way, so that
  // sumsq[0..gridDim.x) is the complete list of partials for adam_kernel.
  // Optional narrow-head weight gradient  out[k][n] = sum_b x[b][k] dy[b][n]  (k < 512,
  // n < ld; flat index k*ld + n, 64 per block, after the colsum tiles)
  const float* o_x = nullptr; const float* o_dy = nullptr; float* o_out = nullptr;
  int o_B = 0, o_ld = 0; unsigned o_tiles = 0;
  float* sumsq = nullptr;           // null: off
  const float* presum_src = nullptr;
  int presum_n = 0;
  int32_t* bump_count = nullptr;    // optax count_inc, done here when the optimiser follows
  const unsigned* abort = nullptr;  // non-zero word: the step is void (dz_head_chain.h gave up): no count_inc
  RmsApply rms;                     // p == nullptr: off
  RmsOnFly of;                      // feat == nullptr: off; `of.blocks` tile blocks in front of the flat ones
};
__device__ __forceinline__ void finalize_grads_block(const FinalizeJobs& J, unsigned b);
__global__ __launch_bounds__(256) void finalize_grads_kernel(FinalizeJobs J) {
  finalize_grads_block(J, blockIdx.x);
}
// The same launch with the NEXT step's replay sample + gather as its first sg_blocks
// blocks (dense learners whose optimiser lives in this launch: RMSProp; see
// adam_sg_kernel for the Adam learners).
__global__ __launch_bounds__(256) void finalize_grads_sg_kernel(FinalizeJobs J, SampleGatherParams sg,
                                                                unsigned sg_blocks) {
  if (blockIdx.x < sg_blocks) { SampleGatherSide::run(sg, blockIdx.x); return; }
  finalize_grads_block(J, blockIdx.x - sg_blocks);
}
__device__ __forceinline__ void finalize_grads_block(const FinalizeJobs& J, unsigned b) {
  __shared__ float red[4][64];
  const int l = threadIdx.x & 63, w = threadIdx.x >> 6;
  const unsigned b_in = b;   // this block's index in the partial list
  if (b == 0 && threadIdx.x == 0 && J.bump_count && !(J.abort && *J.abort)) *J.bump_count = *J.bump_count + 1;
  float v = 0.f;
  if (b < J.r_end[2]) {
    const int j = b < J.r_end[0] ? 0 : (b < J.r_end[1] ? 1 : 2);
    const ReduceJob jb = J.r[j];
    const long i = (long)(b - (j ? J.r_end[j - 1] : 0)) * 64 + l;
    v = dz_slab_sum(jb.part, jb.S, jb.n, i < jb.n ? i : jb.n - 1, w);
    red[w][l] = v;
    __syncthreads();
    if (w != 0) return;
    float o = 0.f;
    if (i < jb.n) {
      o = (red[0][l] + red[1][l]) + (red[2][l] + red[3][l]);
      jb.out[i] = o;
      if (J.rms.p) dz_rms_apply_at(J.rms, jb.out + i, o);
    }
    if (J.sumsq) {
      o = dz_wave_sum(o * o);
      if (l == 0) J.sumsq[b_in] = o;
    }
    return;
  }
  b -= J.r_end[2];
  if (b < J.c_tiles[0] + J.c_tiles[1]) {
    const int j = b < J.c_tiles[0] ? 0 : 1;
    const ColsumJob jb = J.c[j];
    const int c = (int)(b - (j ? J.c_tiles[0] : 0)) * 64 + l;
    v = dz_slab_sum(jb.m, jb.rows, jb.ld, min(c, jb.cols - 1), w);
    red[w][l] = v;
    __syncthreads();
    if (w != 0) return;
    float sq = 0.f;
    if (c < jb.cols) {
      const float s = (red[0][l] + red[1][l]) + (red[2][l] + red[3][l]);
      if (jb.out) { jb.out[c] = s; sq += s * s; if (J.rms.p) dz_rms_apply_at(J.rms, jb.out + c, s); }
      if (jb.out_scaled) { const float t = s * jb.scale[c]; jb.out_scaled[c] = t; sq += t * t; }
    }
    if (J.sumsq) {
      sq = dz_wave_sum(sq);
      if (l == 0) J.sumsq[b_in] = sq;
    }
    return;
  }
  b -= J.c_tiles[0] + J.c_tiles[1];
  if (b < J.o_tiles) {
    const int i = (int)b * 64 + l, k = i / J.o_ld, n = i % J.o_ld;
    for (int s0 = 0; s0 < J.o_B; s0 += 32) {
      float x[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int s = s0 + w + 4 * j, sc = min(s, J.o_B - 1);
        const float y = J.o_x[(long)sc * 512 + k] * J.o_dy[(long)sc * J.o_ld + n];
        x[j] = s < J.o_B ? y : 0.f;
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) v += x[j];
    }
    red[w][l] = v;
    __syncthreads();
    if (w == 0) {
      const float s = (red[0][l] + red[1][l]) + (red[2][l] + red[3][l]);
      J.o_out[i] = s;
      if (J.rms.p) dz_rms_apply_at(J.rms, J.o_out + i, s);
      if (J.sumsq) {
        const float sq = dz_wave_sum(s * s);
        if (l == 0) J.sumsq[b_in] = sq;
      }
    }
    return;
  }
  b -= J.o_tiles;
  if (J.of.feat) {   // fc1 tiles whose gradient is formed here (RMSProp dense learners)
    if (b < J.of.blocks) {
      __shared__ __attribute__((aligned(16))) float of_lds[kRofLds];
      rms_fc1_block(b, J.of, J.rms, of_lds);
      return;
    }
    b -= J.of.blocks;
  }
  if (J.rms.p) {  // flat optimiser blocks (dense learners: no presum blocks)
    if (J.rms.flat_blocks) dz_rms_flat(J.rms, b);
    return;
  }
  // presum block
  {
    const int i0 = (int)b * 1024 + threadIdx.x;
#pragma unroll
    for (int j = 0; j < 4; ++j) v += (i0 + 256 * j) < J.presum_n ? J.presum_src[i0 + 256 * j] : 0.f;
    v = dz_wave_sum(v);
    if (l == 0) red[0][w] = v;
    __syncthreads();
    if (threadIdx.x == 0)
      J.sumsq[b_in] = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
  }
}

__device__ __forceinline__ float wave_max(float v) { return dz_wave_max(v); }
__device__ __forceinline__ float wave_sum(float v) { return dz_wave_sum(v); }

// One wave per sample; lane k owns atom k (K <= 64).
// ref: networks.py:254-258 (dueling, softmax, expectation),
//      rainbow/agent.py:97-109 + rlax.categorical_double_q_learning,
//      rainbow/agent.py:194 (priorities).
// PRE = 1: the fc2 split-K epilogue is done here (no separate launch): all 256
// threads first form the block's three output rows
//     out[g][c] = sum_s part[s][g*B + b][c] + sig_b[c] * eps_out_g[c]
// in LDS (and write them to fc2_out for inspection), then wave 0 runs the loss
// on LDS-resident rows.  PRE = 0: fc2_out is read as given (64 threads suffice).
struct HeadPre {
  const float* part;   // [S][rows][ld]
  int S, rows;
  const float* prm[3];
  const float* nz[3];
  long b_sig;          // sigma-bias offset in the parameter vector
  int eps_out;         // its noise offset
  int plain_bias;      // 1: the bias at b_sig is an ordinary one (no noise factor): C51
  int groups;          // rows g*B + b formed per sample (0 = 3: Rainbow)
};
// Launch with 256 threads: the 4 waves share the selector's per-action softmaxes
// (A of them, 3 wave reductions each); wave 0 alone runs the rest.
struct HeadLossArgs {
  float* fc2_out; int ld, val_off, B, A, K, dueling, sel_group, tgt_group;
  const int64_t* a_tm1; const double* r_t; const double* d_t; const float* weights;
  const float* support; float* dout2; float* losses; float* priorities;
  float* q_sel_out; float* target_out; HeadPre pre;
};
// SEAM = 1 (the multi-role head launch, dz_head_chain.h): the fc2 slabs are produced by other
// workgroups of the SAME launch -- they are read with coherent loads until none of the real
// columns is missing (dz_seam.h) -- and dlogits are stored as a seam for the backward role.
#ifndef DZ_HC_NAP_C
#define DZ_HC_NAP_C 8
#endif
struct HeadSeam {
  unsigned* fail = nullptr; int limit = 0;
  int stages = 0, tiles0 = 0, tiles = 0, groups = 0;   // the producers: (stage, 64-column tile, apply)
  long long* dbg = nullptr;                            // (DZ_HC_STAMPS builds)
};
template <int PRE, int SEAM>
__device__ __forceinline__ void rainbow_head_loss_block(const HeadLossArgs& q, const int b,
                                                        float* s_rows, const HeadSeam seam) {
  // (block-scope restrict: the kernel's pointer arguments were __restrict__ before they moved into
  // HeadLossArgs; without it every store to fc2_out / q_sel_out orders the loads behind it)
  float* __restrict__ const fc2_out = q.fc2_out;
  const int ld = q.ld, val_off = q.val_off, B = q.B, A = q.A, K = q.K, dueling = q.dueling;
  const int sel_group = q.sel_group, tgt_group = q.tgt_group;
  const int64_t* __restrict__ const a_tm1 = q.a_tm1; const double* __restrict__ const r_t = q.r_t;
  const double* __restrict__ const d_t = q.d_t;
  const float* __restrict__ const weights = q.weights; const float* __restrict__ const support = q.support;
  float* __restrict__ const dout2 = q.dout2; float* __restrict__ const losses = q.losses;
  float* __restrict__ const priorities = q.priorities;
  float* __restrict__ const q_sel_out = q.q_sel_out; float* __restrict__ const target_out = q.target_out;
  const HeadPre pre = q.pre;
  __shared__ float s_p[64];
  __shared__ float s_z[64];
  __shared__ float s_q[256];         // selector q-values (A <= 256)
  const int k = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
  // every per-sample scalar and the support are requested NOW, so that their
  // trips to memory overlap the slab loads below instead of following them
  const int kk = min(k, K - 1);
  const float z_ld = support[kk];
  const float zn_ld = support[min(kk + 1, K - 1)], zp_ld = support[max(kk - 1, 0)];
  const float vmin = support[0], vmax = support[K - 1];
  const float r = (float)r_t[b], g = (float)d_t[b];  // f64 -> f32 at the jit boundary
  const int a0 = (int)a_tm1[b];
  const float w_b = weights[b];
  if (PRE) {
    // All loads of a round are issued before any is consumed (the slabs were
    // written by the previous kernel: every load is a ~2 us trip past L2, so the
    // number of round trips, not the 34 KB, is what this phase costs).
    constexpr int E = 5, SMAX = 8;
    const int n = (pre.groups ? pre.groups : 3) * ld;
    if constexpr (SEAM) {
      // one word per producer first: this sample's row of every (stage, column tile, apply)
      // workgroup of the fc2 role (the first column of its tile)
      const int np = seam.stages * seam.tiles * seam.groups;
      for (int t0 = 0; t0 < np; t0 += 256) {   // (np <= 192 for every Atari action set: one trip)
        const int t = t0 + (int)threadIdx.x;
        const int st = t % seam.stages, gt = t / seam.stages, ct = gt % seam.tiles, g = gt / seam.tiles;
        const int col = ct < seam.tiles0 ? 64 * ct : val_off + 64 * (ct - seam.tiles0);
        const float* word = t < np ? pre.part + ((long)st * pre.rows + (long)g * B + b) * ld + col : nullptr;
        if (act_watch_each<DZ_HC_NAP_C>(word, seam.fail, seam.limit)) {
          if (threadIdx.x == 0) { losses[b] = __builtin_nanf(""); priorities[b] = __builtin_nanf(""); }
          return;
        }
      }
#ifdef DZ_HC_STAMPS
      if (seam.dbg && threadIdx.x == 0) seam.dbg[1] = (long long)wall_clock64();
#endif
    }
    if constexpr (SEAM) {
      // 16-byte coherent loads: a thread owns four columns of one apply's row and reads their
      // four stage slabs; fold order as below (slab order from 0.f, then the sigma bias)
      const int ld4 = ld >> 2, n4 = (pre.groups ? pre.groups : 3) * ld4;
      const __amdgpu_buffer_rsrc_t pr = act_rsrc(pre.part);
      for (int base = 0; base < n4; base += 512) {
        float4 v[2][4], bs4[2];
        unsigned okc[2][4];
        int round = 0;
        bool miss = false, give_up = false;
        do {
#pragma unroll
          for (int e = 0; e < 2; ++e) {
            const int i = min(base + (int)threadIdx.x + 256 * e, n4 - 1);
            const int g = i / ld4, c = 4 * (i - g * ld4);
#pragma unroll
            for (int st = 0; st < 4; ++st)
              v[e][st] = act_load4(pr, (unsigned)((st * pre.rows + g * B + b) * ld + c) * 4u);
            const float* prm = g == 0 ? pre.prm[0] : (g == 1 ? pre.prm[1] : pre.prm[2]);
            const float* nz = g == 0 ? pre.nz[0] : (g == 1 ? pre.nz[1] : pre.nz[2]);
            const float4 ez = dz_ld4(nz + pre.eps_out + c), sb = dz_ld4(prm + pre.b_sig + c);
            bs4[e] = dz_f4(sb.x * (pre.plain_bias ? 1.0f : ez.x), sb.y * (pre.plain_bias ? 1.0f : ez.y),
                           sb.z * (pre.plain_bias ? 1.0f : ez.z), sb.w * (pre.plain_bias ? 1.0f : ez.w));
#pragma unroll
            for (int j = 0; j < 4; ++j)   // (pad columns: nothing to wait for)
              okc[e][j] = (c + j < A * K || (dueling && c + j >= val_off && c + j < val_off + K)) ? 0u : 1u;
          }
          unsigned all = 1u;
#pragma unroll
          for (int e = 0; e < 2; ++e)
#pragma unroll
            for (int st = 0; st < 4; ++st) {
              all &= okc[e][0] | (__builtin_bit_cast(unsigned, v[e][st].x) != 0u ? 1u : 0u);
              all &= okc[e][1] | (__builtin_bit_cast(unsigned, v[e][st].y) != 0u ? 1u : 0u);
              all &= okc[e][2] | (__builtin_bit_cast(unsigned, v[e][st].z) != 0u ? 1u : 0u);
              all &= okc[e][3] | (__builtin_bit_cast(unsigned, v[e][st].w) != 0u ? 1u : 0u);
            }
          miss = all == 0u;
        } while (act_again(miss, round++, seam.fail, &give_up, seam.limit));
        if (give_up) {   // (workgroup-uniform) the slabs never came: this sample's results are not numbers
          if (threadIdx.x == 0) { losses[b] = __builtin_nanf(""); priorities[b] = __builtin_nanf(""); }
          return;
        }
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          const int i = base + (int)threadIdx.x + 256 * e;
          if (i < n4) {
            const int g = i / ld4, c = 4 * (i - g * ld4);
            float4 acc = dz_f4zero();
#pragma unroll
            for (int st = 0; st < 4; ++st) {
              acc.x += v[e][st].x; acc.y += v[e][st].y; acc.z += v[e][st].z; acc.w += v[e][st].w;
            }
            // (the one-launch-per-stage form adds four empty slabs here: x + 0.f)
            acc.x += 0.f; acc.y += 0.f; acc.z += 0.f; acc.w += 0.f;
            acc.x += bs4[e].x; acc.y += bs4[e].y; acc.z += bs4[e].z; acc.w += bs4[e].w;
            *(float4*)(s_rows + g * ld + c) = acc;
            *(float4*)(fc2_out + ((long)g * B + b) * ld + c) = acc;
          }
        }
      }
    } else
    for (int base = 0; base < n; base += 256 * E) {
      float v[E][SMAX];
      float bs[E];
      unsigned okc[E];
      int round = 0;
      bool miss = false, give_up = false;
      do {
        miss = false;
#pragma unroll
        for (int e = 0; e < E; ++e) {
          const int i = min(base + (int)threadIdx.x + 256 * e, n - 1);
          const int g = i / ld, c = i - g * ld;
          const float* src = pre.part;
          const long row = (long)g * B + b;
          const long rows = (long)pre.rows;
          // (pad columns are never written by the producers: not waited for, and zero)
          const bool real = c < A * K || (dueling && c >= val_off && c < val_off + K);
#pragma unroll
          for (int sidx = 0; sidx < SMAX; ++sidx) {
            const float* sp = src + ((long)min(sidx, pre.S - 1) * rows + row) * ld + c;
            float t;
            if constexpr (SEAM) t = act_load(sp);
            else t = *sp;
            v[e][sidx] = sidx < pre.S ? t : 0.f;
          }
          okc[e] = real ? 0u : 1u;   // (pad columns: nothing to wait for)
          const float* prm = g == 0 ? pre.prm[0] : (g == 1 ? pre.prm[1] : pre.prm[2]);
          const float* nz = g == 0 ? pre.nz[0] : (g == 1 ? pre.nz[1] : pre.nz[2]);
          const float ez = nz[pre.eps_out + c];
          bs[e] = prm[pre.b_sig + c] * (pre.plain_bias ? 1.0f : ez);
        }
        if constexpr (SEAM) {   // the checks behind ALL the loads, branch-free
          unsigned all = 1u;
#pragma unroll
          for (int e = 0; e < E; ++e)
#pragma unroll
            for (int sidx = 0; sidx < SMAX; ++sidx)
              all &= (okc[e] | (sidx >= pre.S ? 1u : 0u) |
                      (__builtin_bit_cast(unsigned, v[e][sidx]) != 0u ? 1u : 0u));
          miss = all == 0u;
        }
      } while (SEAM && act_again(miss, round++, seam.fail, &give_up, seam.limit));
      if (SEAM && give_up) {   // (workgroup-uniform) the slabs never came: this sample's results are not numbers
        if (threadIdx.x == 0) { losses[b] = __builtin_nanf(""); priorities[b] = __builtin_nanf(""); }
        return;
      }
#pragma unroll
      for (int e = 0; e < E; ++e) {
        const int i = base + (int)threadIdx.x + 256 * e;
        if (i < n) {
          float acc = 0.f;
#pragma unroll
          for (int sidx = 0; sidx < SMAX; ++sidx) acc += v[e][sidx];
          acc += bs[e];
          s_rows[i] = acc;
          const int g = i / ld, c = i - g * ld;
          fc2_out[((long)g * B + b) * ld + c] = acc;
        }
      }
    }
    __syncthreads();
#ifdef DZ_HC_STAMPS
    if (SEAM && seam.dbg && threadIdx.x == 0) seam.dbg[2] = (long long)wall_clock64();
#endif
  }
  const bool on = k < K;
  const int NA = val_off;  // value-head columns start at the padded offset
  const float z = on ? z_ld : 0.f;
  const float invA = 1.0f / (float)A;

  // logits of action a: dueling  v + adv_a - mean_a adv  (networks.py:254) or the
  // head output itself (c51_atari_network, networks.py:329).
  // ---- selector network: q_values -> argmax.  Rainbow: online(s_t)
  // (double-Q, rainbow/agent.py:91-93); C51: the target network itself
  // (rlax.categorical_q_learning) ----
  const float* o1 = PRE ? s_rows + sel_group * ld : fc2_out + (long)(sel_group * B + b) * ld;
  float mean_adv = 0.f;
  if (dueling) {
    if constexpr (PRE) {   // rows in LDS: the plain loop is the fastest form measured
      for (int a = 0; a < A; ++a) mean_adv += on ? o1[a * K + k] : 0.f;
    } else {               // rows in global memory: 8 loads in flight per round, same order
      for (int a0 = 0; a0 < A; a0 += 8) {
        float t8[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) t8[u] = o1[min(a0 + u, A - 1) * K + kk];
#pragma unroll
        for (int u = 0; u < 8; ++u) mean_adv += (on && a0 + u < A) ? t8[u] : 0.f;
      }
    }
    mean_adv /= (float)A;
  }
  const float v1 = (on && dueling) ? o1[NA + k] : 0.f;
  for (int a = wave; a < A; a += nwaves) {
    const float lg = on ? (v1 + o1[a * K + k] - mean_adv) : -__builtin_inff();
    const float mx = wave_max(lg);
    const float e = on ? expf(lg - mx) : 0.f;
    const float sm = wave_sum(e);
    const float q = wave_sum((e / sm) * z);
    if (k == 0) {
      s_q[a] = q;
      if (q_sel_out) q_sel_out[b * A + a] = q;
    }
  }
  __syncthreads();
  float best_q = -__builtin_inff();
  int a_star = 0;
  for (int a = 0; a < A; ++a) {
    const float q = s_q[a];
    if (q > best_q) { best_q = q; a_star = a; }  // first maximum, as jnp.argmax
  }
  // From here on the waves split the work (256-thread launches; a one-wave launch runs
  // everything in order): wave 0 the target distribution of the selected action, wave 1
  // the log-softmax of group 0, then all of them a quarter of the projection's K x K
  // loop each; wave 0 finishes.
  const bool multi = nwaves == 4;
  __shared__ float s_lsm[64], s_sm0[64], s_m[3][64];
  const float* o2 = PRE ? s_rows + tgt_group * ld : fc2_out + (long)(tgt_group * B + b) * ld;
  if (wave == 0) {
    // ---- target distribution of the selected action ----
    float mean2 = 0.f;
    if (dueling) {
      if constexpr (PRE) {   // rows in LDS: the plain loop is the fastest form measured
        for (int a = 0; a < A; ++a) mean2 += on ? o2[a * K + k] : 0.f;
      } else {               // rows in global memory: 8 loads in flight per round, same order
        for (int a0 = 0; a0 < A; a0 += 8) {
          float t8[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) t8[u] = o2[min(a0 + u, A - 1) * K + kk];
#pragma unroll
          for (int u = 0; u < 8; ++u) mean2 += (on && a0 + u < A) ? t8[u] : 0.f;
        }
      }
      mean2 /= (float)A;
    }
    const float lg2 = on ? ((dueling ? o2[NA + k] : 0.f) + o2[a_star * K + k] - mean2)
                         : -__builtin_inff();
    const float mx2 = wave_max(lg2);
    const float e2 = on ? expf(lg2 - mx2) : 0.f;
    const float p_t = e2 / wave_sum(e2);
    // ---- Cramer projection of (r + g z, p_t) onto the support: operands ----
    float zp = r + g * z;
    zp = fminf(fmaxf(zp, vmin), vmax);
    s_p[k] = on ? p_t : 0.f;
    s_z[k] = zp;
  }
  // ---- group 0: log_softmax(logits_tm1[a_tm1]) (wave 1, or wave 0 of a one-wave launch) ----
  float sh = 0.f, e0 = 0.f, se0 = 1.f, lsm = 0.f;
  if (wave == (multi ? 1 : 0)) {
    const float* o0 = PRE ? s_rows : fc2_out + (long)(0 * B + b) * ld;
    float mean0 = 0.f;
    if (dueling) {
      if constexpr (PRE) {   // rows in LDS: the plain loop is the fastest form measured
        for (int a = 0; a < A; ++a) mean0 += on ? o0[a * K + k] : 0.f;
      } else {               // rows in global memory: 8 loads in flight per round, same order
        for (int a0 = 0; a0 < A; a0 += 8) {
          float t8[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) t8[u] = o0[min(a0 + u, A - 1) * K + kk];
#pragma unroll
          for (int u = 0; u < 8; ++u) mean0 += (on && a0 + u < A) ? t8[u] : 0.f;
        }
      }
      mean0 /= (float)A;
    }
    const float lg0 = on ? ((dueling ? o0[NA + k] : 0.f) + o0[a0 * K + k] - mean0)
                         : -__builtin_inff();
    const float mx0 = wave_max(lg0);
    sh = lg0 - mx0;
    e0 = on ? expf(sh) : 0.f;
    se0 = wave_sum(e0);
    lsm = sh - logf(se0);
    if (multi) { s_lsm[k] = lsm; s_sm0[k] = e0 / se0; }
  }
  __syncthreads();
  if (!multi && wave != 0) return;
  // ---- projection: m[k] = sum_j clip(1 - |z'_j - z_k| / dz)_[0,1] p_j; wave w takes
  // j in [w K/4, (w+1) K/4), the four partial sums are added in wave order ----
  float m = 0.f;
  if (on) {
    const float zq = z;
    const float dpos = (k + 1 < K ? zn_ld : vmin) - zq;
    const float dneg = zq - (k > 0 ? zp_ld : vmax);
    const float rpos = dpos > 0.f ? 1.0f / dpos : 0.f;
    const float rneg = dneg > 0.f ? 1.0f / dneg : 0.f;
    const int jq = (K + 3) / 4;
    const int j0 = multi ? wave * jq : 0, j1 = multi ? min(K, j0 + jq) : K;
    for (int j = j0; j < j1; ++j) {
      const float delta = s_z[j] - zq;
      const float dh = delta >= 0.f ? delta * rpos : -(delta * rneg);
      const float c = fminf(fmaxf(1.0f - dh, 0.f), 1.f);
      m += c * s_p[j];
    }
  }
  if (multi) {
    if (wave > 0) s_m[wave - 1][k] = m;
    __syncthreads();
    if (wave != 0) return;
    m = ((m + s_m[0][k]) + s_m[1][k]) + s_m[2][k];
    lsm = s_lsm[k];
  }
  if (target_out && on) target_out[b * K + k] = m;
  const float sm0 = multi ? s_sm0[k] : e0 / se0;
  const float loss = -wave_sum(on ? m * lsm : 0.f);
  const float msum = wave_sum(m);
  // d loss / d logits_tm1[a0][k], scaled by w/B (loss = mean(losses*w))
  const float gk = on ? (sm0 * msum - m) * (w_b / (float)B) : 0.f;
  if (on) {
    float* d = dout2 + (long)b * ld;
    for (int a = 0; a < A; ++a) {  // dueling: dadv[a][k] = G[a][k] - mean_a G[.][k]
      const float dv = (a == a0 ? gk : 0.f) - (dueling ? gk * invA : 0.f);
      if constexpr (SEAM) act_store(d + a * K + k, act_mark(dv));
      else d[a * K + k] = dv;
    }
    if (dueling) {  // dval[k] = sum_a G[a][k]
      if constexpr (SEAM) act_store(d + NA + k, act_mark(gk));
      else d[NA + k] = gk;
    }
  }
  if (k == 0) {
    losses[b] = loss;
    priorities[b] = fminf(fmaxf(fabsf(loss), 0.f), 100.f);
  }
}


template <int PRE>
__global__ __launch_bounds__(256) void rainbow_head_loss_kernel(
    float* __restrict__ fc2_out, int ld, int val_off, int B, int A, int K,
    int dueling, int sel_group, int tgt_group, const int64_t* __restrict__ a_tm1, const double* __restrict__ r_t,
    const double* __restrict__ d_t, const float* __restrict__ weights,
    const float* __restrict__ support, float* __restrict__ dout2,
    float* __restrict__ losses, float* __restrict__ priorities,
    float* __restrict__ q_sel_out, float* __restrict__ target_out, HeadPre pre,
    GramX gram = GramX{}) {
  extern __shared__ float s_rows[];  // PRE: [3][ld]
  // workgroups beyond the batch: the wide layer's input Grams (dz_gram.h), which need
  // nothing this kernel computes and run on CUs it leaves idle
  if ((int)blockIdx.x >= B) {
    __shared__ dz_d4 s_gram[3 * 64];
    dz_gram_x_block(gram, blockIdx.x - (unsigned)B, s_gram);
    return;
  }
  const HeadLossArgs q = {fc2_out, ld, val_off, B, A, K, dueling, sel_group, tgt_group, a_tm1, r_t, d_t,
                          weights, support, dout2, losses, priorities, q_sel_out, target_out, pre};
  rainbow_head_loss_block<PRE, 0>(q, (int)blockIdx.x, s_rows, HeadSeam{});
}

// Partial sums of squares of the gradient (global norm) -- and the optimiser step
// count: thread 0 of block 0 performs optax's `count_inc = count + 1`, so the
// Adam launch that follows reads the incremented count without a race.
__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ g,
                                                    long n, float* __restrict__ part,
                                                    int32_t* __restrict__ count) {
  __shared__ float red[4];
  float s = 0.f;
  const long n4 = n >> 2;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    const float4 v = ((const float4*)g)[i];
    s += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
  }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    part[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
    if (blockIdx.x == 0) *count = *count + 1;
  }
}

// A block of the gradient vector that is not stored but derived on the fly:
//   g[dst_off + k*ld + n] = g[src_off + k*ld + n] * (eps_in_h[k] * eps_out[n])
// (the sigma-weight gradient of a noisy layer is its mu-weight gradient times the
// noise outer product, networks.py:168-176): the weight-gradient kernel then
// writes 12.85 MB less for Rainbow's fc1 (this kernel reads the mu block twice
// instead of mu and sigma once each: same bytes).
struct DerivedGrad {
  long dst_off, src_off;    // multiples of 4
  int rows, ld;             // ld multiple of 4
  int split_col;            // columns < split_col use eps_in0, the rest eps_in1
  const float* eps_in0;
  const float* eps_in1;
  const float* eps_out;     // indexed by column
  int on;
};

// a / b for a divisor shared by the whole launch, given rb = 1.0f / b (IEEE): quotient
// estimate, exact residual, one correction (Markstein) -- 3 instructions instead of the
// ~11 of the IEEE expansion, the correctly rounded quotient whenever the quotient and the
// residual are normal numbers.
__device__ __forceinline__ float dz_div_by(float a, float b, float rb) {
  const float q = a * rb;
  return __builtin_fmaf(__builtin_fmaf(-q, b, a), rb, q);
}
// One element of clip_by_global_norm + adam + apply_updates (the single
// definition every optimiser path inlines, so that all of them round alike).
__device__ __forceinline__ void adam_elem(float& P, float G, float& M, float& V, bool pass,
                                          float gn, float bc1, float bc2, float lr, float b1,
                                          float b2, float eps, float max_norm) {
  // (launch-uniform, hoisted.)  An overflowed norm: G / inf must be sign(G)*0 as IEEE division --
  // and optax's clip -- give it; q = G * (1/inf) = 0 times b = inf in the residual would be NaN,
  // so the pair (b, 1/b) becomes (0, 0): residual G, result fma(G, 0, +-0) = sign(G)*0, NaN for
  // G = inf / NaN as the division.  A NaN norm stays NaN.
#ifdef DZ_ADAM_CHEAP   // (timing probe, variant builds only: wrong numbers)
  M = M + G; V = V + G * gn; P = P + M * lr;
  return;
#endif
  const bool gn_inf = gn == __builtin_inff();
  const float gb = gn_inf ? 0.0f : gn;
  const float rg = gn_inf ? 0.0f : 1.0f / gn, r1 = 1.0f / bc1, r2 = 1.0f / bc2;
  const float gj = pass ? G : dz_div_by(G, gb, rg) * max_norm;
  M = (1.0f - b1) * gj + b1 * M;
  V = (1.0f - b2) * (gj * gj) + b2 * V;
  const float upd = dz_div_by(M, bc1, r1) / (sqrtf(dz_div_by(V, bc2, r2)) + eps);
  P = P + (-lr) * upd;
}

// optax.clip_by_global_norm then optax.adam, then apply_updates.  Every block
// folds the `nparts` norm partials itself (same order everywhere: identical
// scalars in all blocks) instead of waiting on a one-block "scalars" launch
// (that launch cost ~5 us + a ~2 us gap for 2 KB of work); block 0 publishes
// the scalars (global norm, bias corrections, clip flag, mean weighted loss).
struct AdamScalars { float gn, bc1, bc2; bool pass; };
__device__ __forceinline__ AdamScalars adam_scalars(const float* __restrict__ part, int nparts,
                                                    const int32_t* __restrict__ count, float b1,
                                                    float b2, float max_norm, float* red) {
  // thread t sums part[t], part[t+256], ... in that order; 8 clamped loads are in
  // flight per round (a plain loop serialises one L2 round trip per partial, and
  // every block of this one-wave launch pays that chain before its first byte);
  // out-of-range slots add +0 to a non-negative sum: no change
  float s = 0.f;
  for (int base = 0; base < nparts; base += 8 * 256) {
    float x[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int i = base + j * 256 + (int)threadIdx.x;
      const float y = part[min(i, nparts - 1)];
      x[j] = i < nparts ? y : 0.f;
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) s += x[j];
  }
  s = dz_wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  AdamScalars o;
  o.gn = sqrtf((red[0] + red[1]) + (red[2] + red[3]));
  const int c = *count;  // already incremented (finalize / sumsq_kernel)
  o.bc1 = 1.0f - powf(b1, (float)c); o.bc2 = 1.0f - powf(b2, (float)c);
  o.pass = !(max_norm > 0.f && !(o.gn < max_norm));
  return o;
}
// The same in two halves, for callers that want the partials' trip to memory in flight before
// their own streams (adam_fc1_block): `adam_partials_request` issues the loads (at most 16 x 256
// partials), `adam_scalars_from` is the rest -- identical arithmetic and order.
constexpr int kAdamPartRounds = 2;   // x 8 loads per thread
static_assert(kNormFinal <= kAdamPartRounds * 8 * 256, "adam_partials_request covers every fused-norm partial");
struct AdamPartials { float x[kAdamPartRounds][8]; };
__device__ __forceinline__ AdamPartials adam_partials_request(const float* __restrict__ part, int nparts) {
  AdamPartials r;
#pragma unroll
  for (int q = 0; q < kAdamPartRounds; ++q)
#pragma unroll
    for (int j = 0; j < 8; ++j)
      r.x[q][j] = part[max(min(q * 8 * 256 + j * 256 + (int)threadIdx.x, nparts - 1), 0)];   // (nparts == 0: unused)
  return r;
}
__device__ __forceinline__ AdamScalars adam_scalars_from(const AdamPartials& r, int nparts,
                                                         const int32_t* __restrict__ count, float b1,
                                                         float b2, float max_norm, float* red) {
  float s = 0.f;
#pragma unroll
  for (int q = 0; q < kAdamPartRounds; ++q) {
    if (q * 8 * 256 < nparts) {   // (uniform) adam_scalars' rounds: `base < nparts`
#pragma unroll
      for (int j = 0; j < 8; ++j) s += (q * 8 * 256 + j * 256 + (int)threadIdx.x) < nparts ? r.x[q][j] : 0.f;
    }
  }
  s = dz_wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  AdamScalars o;
  o.gn = sqrtf((red[0] + red[1]) + (red[2] + red[3]));
  const int c = *count;
  o.bc1 = 1.0f - powf(b1, (float)c); o.bc2 = 1.0f - powf(b2, (float)c);
  o.pass = !(max_norm > 0.f && !(o.gn < max_norm));
  return o;
}
__device__ __forceinline__ void adam_publish(const AdamScalars& o, float* __restrict__ sc,
                                             const float* __restrict__ losses,
                                             const float* __restrict__ weights, int B) {
  sc[DZ_SC_GNORM] = o.gn; sc[DZ_SC_BC1] = o.bc1; sc[DZ_SC_BC2] = o.bc2;
  sc[DZ_SC_CLIP] = o.pass ? 1.f : 0.f;
  float l = 0.f;
  for (int i = 0; i < B; ++i) l += losses[i] * weights[i];
  sc[DZ_SC_LOSS] = l / (float)B;
}

// Optimiser block `bid` of `nblk` (the kernels below map their grids onto these).
__device__ __forceinline__ void adam_body(
    unsigned bid, unsigned nblk,
    float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
    float* __restrict__ v, long n4, const float* __restrict__ part, int nparts,
    const int32_t* __restrict__ count, const float* __restrict__ losses,
    const float* __restrict__ weights, int B, float* __restrict__ sc, float lr, float b1,
    float b2, float eps, float max_norm, const DerivedGrad& dg) {
  __shared__ float red[4];
  // Branch-free loads (derived block or not: clamped pointers, the noise factors
  // select to 1, x * (1 * 1) == x), the first element's loads issued before the
  // scalar prelude and each next element's before the current arithmetic.
  struct Elem { float4 g, m, v, p, eo; float ei; bool der; };
  const long d0 = dg.on ? (dg.dst_off >> 2) : 0, d1 = dg.on ? d0 + (((long)dg.rows * dg.ld) >> 2) : 0;
  auto load = [&](long i, Elem& e) {
    const bool der = i >= d0 && i < d1;
    const long rel4 = der ? i - d0 : 0;
    const unsigned rel = (unsigned)rel4 << 2;  // < 2^31
    const unsigned k = dg.on ? rel / (unsigned)dg.ld : 0u, n = rel - k * (unsigned)dg.ld;
    const float4* gsrc = der ? (const float4*)(g + dg.src_off) + rel4 : (const float4*)g + i;
    const float* eip = der ? ((int)n < dg.split_col ? dg.eps_in0 : dg.eps_in1) + k : (const float*)p;
    const float4* eop = der ? (const float4*)(dg.eps_out + n) : (const float4*)p;
    e.g = *gsrc; e.m = ((const float4*)m)[i]; e.v = ((const float4*)v)[i];
    e.p = ((const float4*)p)[i];
    e.ei = *eip; e.eo = *eop; e.der = der;  // no use of a loaded value here: nothing waits
  };
  const long stride = (long)nblk * 256;
  long ip = (long)bid * 256 + threadIdx.x;
  Elem cur;
  load(min(ip, n4 - 1), cur);  // clamped, unconditional (no exec-mask block)
  const AdamScalars sc0 = adam_scalars(part, nparts, count, b1, b2, max_norm, red);
  const float gn = sc0.gn, bc1 = sc0.bc1, bc2 = sc0.bc2;
  const bool pass = sc0.pass;
  if (bid == 0 && threadIdx.x == 0) adam_publish(sc0, sc, losses, weights, B);
  while (ip < n4) {
    const long inext = ip + stride;
    Elem nxt;
    load(min(inext, n4 - 1), nxt);
    float* G = (float*)&cur.g; float* M = (float*)&cur.m; float* V = (float*)&cur.v;
    float* P = (float*)&cur.p; const float* EO = (const float*)&cur.eo;
    const float ei = cur.der ? cur.ei : 1.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float gj = G[j] * (ei * (cur.der ? EO[j] : 1.f));
      asm volatile("" : "+v"(gj));  // a rounded product (never contracted into the update)
      adam_elem(P[j], gj, M[j], V[j], pass, gn, bc1, bc2, lr, b1, b2, eps, max_norm);
    }
    ((float4*)m)[ip] = cur.m; ((float4*)v)[ip] = cur.v; ((float4*)p)[ip] = cur.p;
    cur = nxt; ip = inext;
  }
}

__global__ __launch_bounds__(256) void adam_kernel(
    float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
    float* __restrict__ v, long n4, const float* __restrict__ part, int nparts,
    const int32_t* __restrict__ count, const float* __restrict__ losses,
    const float* __restrict__ weights, int B, float* __restrict__ sc, float lr, float b1,
    float b2, float eps, float max_norm, DerivedGrad dg = DerivedGrad{},
    PrioUpdateParams prio = PrioUpdateParams{}, const unsigned* abort = nullptr) {
  // `abort` (nullable, launch-uniform): a multi-role launch of THIS step gave up on a seam
  // (ws_scalars[DZ_SC_CHAIN_FAIL]): the step is void -- no parameter, moment or tree changes
  if (abort && *abort) return;
  // Optional side job (prio.node != null): block 0 is the sum-tree priority
  // write-back (a ~10 us chain of dependent loads on one workgroup that needs only
  // the loss kernel's priorities); this launch is the longest of the step and does
  // not touch the tree, so the chain disappears inside it.  The optimiser blocks
  // are [1, gridDim.x).
  unsigned bid = blockIdx.x, nblk = gridDim.x;
  if (prio.node) {
    if (bid == 0) { PrioUpdateSide::run(prio, 0); return; }
    bid -= 1; nblk -= 1;
  }
  adam_body(bid, nblk, p, g, m, v, n4, part, nparts, count, losses, weights, B, sc, lr, b1, b2,
            eps, max_norm, dg);
}

// The optimiser with the NEXT step's replay sample + gather as its first `sg_blocks`
// blocks (SampleGatherSide: one sampler block, the rest copy 4 KB each after a 5-trip
// descent): a learner that consumes a static replay has nothing between write-back(k)
// -- which an earlier launch of the step carried -- and sample(k+1), so the 8 us
// sample launch of the next step disappears inside this step's longest launch.
__global__ __launch_bounds__(256) void adam_sg_kernel(
    float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
    float* __restrict__ v, long n4, const float* __restrict__ part, int nparts,
    const int32_t* __restrict__ count, const float* __restrict__ losses,
    const float* __restrict__ weights, int B, float* __restrict__ sc, float lr, float b1,
    float b2, float eps, float max_norm, DerivedGrad dg, SampleGatherParams sg,
    unsigned sg_blocks, const unsigned* abort = nullptr) {
  if (blockIdx.x < sg_blocks) { SampleGatherSide::run(sg, blockIdx.x); return; }
  if (abort && *abort) return;   // (see adam_kernel; the next step's sample is still good)
  adam_body(blockIdx.x - sg_blocks, gridDim.x - sg_blocks, p, g, m, v, n4, part, nparts, count,
            losses, weights, B, sc, lr, b1, b2, eps, max_norm, dg);
}

__global__ void copy_kernel(float* __restrict__ dst, const float* __restrict__ src,
                            long n4) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4;
       i += (long)gridDim.x * blockDim.x)
    ((float4*)dst)[i] = ((const float4*)src)[i];
}

// Counter-based generator (splitmix64 finaliser on (seed, counter+i)), two
// 24-bit uniforms -> standard normal via inverse CDF restricted to [-2,2].
__device__ __forceinline__ uint64_t mix64(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}
// Element i of the call is stream position  counter + *step * step_mul + i0 + i
// (step_mul == 0: n, i.e. consecutive calls tile the stream).  step_mul / i0 let a
// call draw a SUB-RANGE of a step's block -- the precomputed target apply draws
// block 2 of 3 at the positions the three-apply step would use.
struct NoiseParams {
  float* out; long n; uint64_t seed; uint64_t counter; const int32_t* step;
  long step_mul = 0; long i0 = 0;
};
// The value at stream position `pos` of seed `seed` (a pure function: any kernel can draw the
// elements it needs itself instead of reading them from a buffer another launch filled).
__device__ __forceinline__ float dz_noise_at(uint64_t seed, uint64_t pos) {
  const uint64_t h = mix64(mix64(seed) ^ mix64(pos));
  // jax.random.truncated_normal: sqrt2 * erfinv(U(erf(lo/sqrt2), erf(hi/sqrt2)))
  const float u01 = ((float)(h >> 40) + 0.5f) * (1.0f / 16777216.0f);
  const float e = 0.9544997361036416f;  // erf(2/sqrt(2))
  const float u = (2.0f * u01 - 1.0f) * e;
  float x = 1.4142135623730951f * erfinvf(u);
  x = fminf(fmaxf(x, -2.0f), 2.0f);
  const float s = sqrtf(fabsf(x));
  return x < 0.f ? -s : (x > 0.f ? s : 0.f);  // sign(x) * sqrt|x| (networks.py:144)
}
__device__ __forceinline__ uint64_t noise_base(const NoiseParams& q) {
  uint64_t counter = q.counter + (uint64_t)q.i0;
  if (q.step) counter += (uint64_t)(*q.step) * (uint64_t)(q.step_mul ? q.step_mul : q.n);  // per-step stream offset
  return counter;
}
__device__ __forceinline__ void noise_fill_at(const NoiseParams& q, long i) {
  if (i >= q.n) return;
  q.out[i] = dz_noise_at(q.seed, noise_base(q) + (uint64_t)i);
}
__global__ void noise_fill_kernel(NoiseParams q) {
  noise_fill_at(q, (long)blockIdx.x * blockDim.x + threadIdx.x);
}
// The same job as extra blocks of a GEMM launch (dz_mfma_gemm_side).
struct NoiseSide {
  typedef NoiseParams Params;
  __device__ static void run(const Params& q, unsigned block) {
    noise_fill_at(q, (long)block * 256 + threadIdx.x);
  }
};

// ---- the seam buffers go back to all-zero bits: side job of the step's first launch ---------------
struct SeamClear {
  float* ptr[3] = {nullptr, nullptr, nullptr};
  int n4[3] = {0, 0, 0};                 // float4 counts
  unsigned blocks = 0;                   // ceil(sum n4 / 256)
};
__device__ __forceinline__ void seam_clear_block(const SeamClear& q, unsigned blk) {
  int i = (int)blk * 256 + (int)threadIdx.x;
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    if (i >= 0 && i < q.n4[j]) {
      float* d = q.ptr[j] + 4 * (long)i;
      act_store2(d, 0.f, 0.f); act_store2(d + 2, 0.f, 0.f);
    }
    i -= q.n4[j];
  }
}
// Noise draw + seam clear as ONE side job of the conv1 forward launch.
struct StepPre { NoiseParams noise; unsigned noise_blocks; SeamClear clr; };
struct StepPreSide {
  typedef StepPre Params;
  __device__ static void run(const Params& q, unsigned block) {
    if (block < q.noise_blocks) noise_fill_at(q.noise, (long)block * 256 + threadIdx.x);
    else seam_clear_block(q.clr, block - q.noise_blocks);
  }
};


// q_values[b][a] = sum_k softmax(v + adv_a - mean_a adv)[k] * support[k]
// (ref: networks.py:254-258); also the greedy action and its value
// (ref: rainbow/agent.py:125-131, first maximum).
// PRE = 1: the fc2 split-K slabs are folded here (see rainbow_head_loss_kernel).
template <int PRE>
__global__ __launch_bounds__(PRE ? 256 : 64) void rainbow_q_values_kernel(
    float* __restrict__ fc2_out, int ld, int val_off, int A, int K,
    const float* __restrict__ support, float* __restrict__ q_out,
    int32_t* __restrict__ greedy_out, float* __restrict__ vmax_out, HeadPre pre,
    int32_t* __restrict__ bump = nullptr) {
  extern __shared__ float s_row[];  // PRE: [ld]
  // the actor's noise-stream position (dz_rainbow_act step_counter): advanced by the
  // LAST launch of an apply, read by the noise draw in the FIRST launch of the next
  if (bump && blockIdx.x == 0 && threadIdx.x == 0) *bump = *bump + 1;
  const int b = blockIdx.x, k = threadIdx.x;
  const float z_ld = support[min(k, K - 1)];
  if (PRE) {
    constexpr int E = 4, SMAX = 8;
    for (int base = 0; base < ld; base += 256 * E) {
      float v[E][SMAX];
      float bs[E];
#pragma unroll
      for (int e = 0; e < E; ++e) {
        const int c = min(base + (int)threadIdx.x + 256 * e, ld - 1);
#pragma unroll
        for (int sidx = 0; sidx < SMAX; ++sidx) {
          const float t = pre.part[((long)min(sidx, pre.S - 1) * pre.rows + b) * ld + c];
          v[e][sidx] = sidx < pre.S ? t : 0.f;
        }
        bs[e] = pre.prm[0][pre.b_sig + c] * pre.nz[0][pre.eps_out + c];
      }
#pragma unroll
      for (int e = 0; e < E; ++e) {
        const int c = base + (int)threadIdx.x + 256 * e;
        if (c < ld) {
          float acc = 0.f;
#pragma unroll
          for (int sidx = 0; sidx < SMAX; ++sidx) acc += v[e][sidx];
          acc += bs[e];
          s_row[c] = acc;
          fc2_out[(long)b * ld + c] = acc;
        }
      }
    }
    __syncthreads();
    if (threadIdx.x >= 64) return;
  }
  const bool on = k < K;
  const float z = on ? z_ld : 0.f;
  const float* o = PRE ? s_row : fc2_out + (long)b * ld;
  float mean_adv = 0.f;
  for (int a = 0; a < A; ++a) mean_adv += on ? o[a * K + k] : 0.f;
  mean_adv /= (float)A;
  const float v = on ? o[val_off + k] : 0.f;
  float best = -__builtin_inff();
  int arg = 0;
  for (int a = 0; a < A; ++a) {
    const float lg = on ? (v + o[a * K + k] - mean_adv) : -__builtin_inff();
    const float mx = wave_max(lg);
    const float e = on ? expf(lg - mx) : 0.f;
    const float sm = wave_sum(e);
    const float q = wave_sum((e / sm) * z);
    if (k == 0) q_out[b * A + a] = q;
    if (q > best) { best = q; arg = a; }
  }
  if (k == 0) {
    if (greedy_out) greedy_out[b] = arg;
    if (vmax_out) vmax_out[b] = best;
  }
}

// q-values, greedy action and its value of ONE finished fc2 row held in LDS (256 threads):
// the 4 waves take the actions round-robin (3 wave reductions each).
__device__ __forceinline__ void dz_q_from_row(const float* s_row, int A, int K, int val_off,
                                              const float* __restrict__ support, float* q_out,
                                              int32_t* greedy_out, float* vmax_out, float* s_q,
                                              float* s_best, int* s_arg) {
  const int tid = threadIdx.x;
  const int k = tid & 63, wave = tid >> 6;
  const bool on = k < K;
  const float z = on ? support[k] : 0.f;
  float mean_adv = 0.f;
  for (int a = 0; a < A; ++a) mean_adv += on ? s_row[a * K + k] : 0.f;
  mean_adv /= (float)A;
  const float vv = on ? s_row[val_off + k] : 0.f;
  for (int a0 = 0; a0 < A; a0 += 64) {        // A <= 256: 64 actions per round of s_q
    for (int a = a0 + wave; a < min(A, a0 + 64); a += 4) {
      const float lg = on ? (vv + s_row[a * K + k] - mean_adv) : -__builtin_inff();
      const float mx = wave_max(lg);
      const float e = on ? expf(lg - mx) : 0.f;
      const float sm = wave_sum(e);
      const float q = wave_sum((e / sm) * z);
      if (k == 0) { s_q[a - a0] = q; q_out[a] = q; }
    }
    __syncthreads();
    if (tid == 0) {   // first maximum, as jnp.argmax
      float best = a0 ? *s_best : -__builtin_inff();
      int arg = a0 ? *s_arg : 0;
      for (int a = a0; a < min(A, a0 + 64); ++a)
        if (s_q[a - a0] > best) { best = s_q[a - a0]; arg = a; }
      *s_best = best; *s_arg = arg;
      if (a0 + 64 >= A) {
        // (action, value) adjacent and 8-byte aligned -- the acting slot in pinned host
        // memory: ONE 8-byte store, so a host that polls the slot never sees half a pair
        if (greedy_out && vmax_out == (float*)(greedy_out + 1) && ((uintptr_t)greedy_out & 7) == 0) {
          const unsigned long long pr = (unsigned long long)(unsigned)arg |
                                        ((unsigned long long)__builtin_bit_cast(unsigned, best) << 32);
          *(volatile unsigned long long*)greedy_out = pr;
        } else {
          if (greedy_out) *greedy_out = arg;
          if (vmax_out) *vmax_out = best;
        }
      }
    }
    __syncthreads();
  }
}

// --------------------------------------------------------------------------- //
//  The actor's tail in ONE launch (batch <= 8): fc1 split-K epilogue, noisy fc2
//  (adv2, val2) as a GEMV over W_eff, dueling + softmax + expectation + argmax.
//  ref: networks.py:239-258 (heads), rainbow/agent.py:125-131 (select_action).
//
//  At batch 1 the three launches this replaces (fc_epilogue 4.5 us, split-K fc2
//  5.2 us, q-values 7.2 us: tools/act_trace.py) are all launch floor.  Here
//  grid = (column tiles of 32 over [adv2 | val2], batch rows): every workgroup
//    1. rebuilds the row's h1 = relu(sum_s part[s] + b_mu + b_sig eps_out) in LDS
//       (128 KB of L2 reads; cheaper than a launch boundary),
//    2. computes its 32 output columns: thread (kg, c) sums 64 of the 512 k's of
//       h1[k] (Wmu[k][c] + Wsig[k][c] eps_in[k] eps_out[c]), 8 partials per column
//       combined through LDS in a fixed order, plus the sigma bias,
//    3. publishes them and takes a ticket; the LAST workgroup of the row (agent-
//       scope release / acquire around the ticket, MI355X_MICROARCH.md "Workgroup
//       dispatch ... visibility") reads the whole row and emits q-values, greedy
//       action and its value -- which may go straight to pinned host memory.
// --------------------------------------------------------------------------- //
struct ActTailParams {
  const float* part; int S; int rows;        // fc1 slabs [S][rows][1024]
  const float* prm; const float* nz;         // parameters, the apply's noise block
  long fc1_mu_b, fc1_sig_b; int n_fc1_out;   // fc1 biases / their output noise
  FcHead head[2];                            // adv2, val2 (K = 512, x_off 0 / 512)
  long fc2_sig_b; int n_fc2_out;             // sigma bias [ld2] and its noise offset
  int ld2, val_off, A, K;                    // padded row pitch, value-head column offset
  const float* support;
  float* fc2_out;                            // [rows][ld2]
  float* q_out; int32_t* greedy_out; float* vmax_out;
  int* tickets;                              // [rows], zero before the first launch
  int tiles0, tiles;                         // column tiles of head 0 / in total
  int32_t* bump;                             // actor noise-stream counter (nullable)
};

__global__ __launch_bounds__(256) void rainbow_act_tail_kernel(ActTailParams p) {
  __shared__ float s_h1[512];     // this head's half of h1
  __shared__ float s_ein[512];
  __shared__ float s_red[8][32];
  __shared__ float s_row[1024];   // the finished fc2 row (ld2 <= 1024 checked by the host)
  __shared__ float s_q[64];
  __shared__ float s_best;
  __shared__ int s_arg;
  __shared__ int s_last;
  const int row = blockIdx.y, tile = blockIdx.x, tid = threadIdx.x;
  const int hsel = tile >= p.tiles0 ? 1 : 0;
  const FcHead hd = dz_pick_head(p.head, hsel);
  const int c = tid & 31, kg = tid >> 5;
  const int col = (tile - (hsel ? p.tiles0 : 0)) * 32 + c;   // within the head
  const int colc = min(col, hd.ldw - 1);
  // Every load of the kernel's first two phases is issued before anything is
  // consumed (the phases are dependent round trips otherwise: 26 us measured):
  //   (a) the 32 fc1 slabs of this head's 512 h1 columns (2 per thread) + biases,
  //   (b) this thread's 64 k's of Wmu / Wsig for its output column.
  constexpr int SMAX = 32;
  float x[2][SMAX];
#pragma unroll
  for (int e = 0; e < 2; ++e)
#pragma unroll
    for (int j = 0; j < SMAX; ++j)
      x[e][j] = p.part[((long)min(j, p.S - 1) * p.rows + row) * 1024 + hd.x_off + tid + 256 * e];
  float bm[2], bs[2], be[2], ei[2];
#pragma unroll
  for (int e = 0; e < 2; ++e) {
    const int cc = hd.x_off + tid + 256 * e;
    bm[e] = p.prm[p.fc1_mu_b + cc]; bs[e] = p.prm[p.fc1_sig_b + cc]; be[e] = p.nz[p.n_fc1_out + cc];
    ei[e] = p.nz[hd.eps_in + tid + 256 * e];
  }
  const float eo = p.nz[hd.eps_out + colc];
  const float sb = p.prm[p.fc2_sig_b + hd.out_off + colc] * p.nz[p.n_fc2_out + hd.out_off + colc];
  float m[64], g[64];
  {
    const float* wmu = p.prm + hd.w_mu + (long)(kg * 64) * hd.ldw + colc;
    const float* wsg = p.prm + hd.w_sig + (long)(kg * 64) * hd.ldw + colc;
#pragma unroll
    for (int j = 0; j < 64; ++j) { m[j] = wmu[(long)j * hd.ldw]; g[j] = wsg[(long)j * hd.ldw]; }
  }
  // ---- 1. h1 (this head's half) -------------------------------------------------
#pragma unroll
  for (int e = 0; e < 2; ++e) {
    float v = 0.f;
#pragma unroll
    for (int j = 0; j < SMAX; ++j) v += j < p.S ? x[e][j] : 0.f;
    const float h = v + bm[e] + bs[e] * be[e];
    s_h1[tid + 256 * e] = h > 0.f ? h : 0.f;
    s_ein[tid + 256 * e] = ei[e];
  }
  __syncthreads();
  // ---- 2. this workgroup's 32 output columns ------------------------------------
  {
    float acc = 0.f;
#pragma unroll
    for (int j = 0; j < 64; ++j) {
      const int k = kg * 64 + j;
      acc = __builtin_fmaf(s_h1[k], __builtin_fmaf(g[j], s_ein[k] * eo, m[j]), acc);
    }
    s_red[kg][c] = acc;
  }
  __syncthreads();
  if (tid < 32 && col < hd.ldw) {
    const float o = (((s_red[0][c] + s_red[1][c]) + (s_red[2][c] + s_red[3][c])) +
                     ((s_red[4][c] + s_red[5][c]) + (s_red[6][c] + s_red[7][c]))) + sb;
    p.fc2_out[(long)row * p.ld2 + hd.out_off + col] = col < hd.N ? o : 0.f;
  }
  // ---- 3. ticket: the last workgroup of the row finishes it -----------------------
  __syncthreads();
  if (tid == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const int t = __hip_atomic_fetch_add(p.tickets + row, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int last = t == p.tiles - 1;
    if (last) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    s_last = last;
  }
  __syncthreads();
  if (!s_last) return;
  for (int i = tid; i < p.ld2; i += 256)
    s_row[i] = __hip_atomic_load(p.fc2_out + (long)row * p.ld2 + i, __ATOMIC_RELAXED,
                                 __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  if (tid == 0) {
    p.tickets[row] = 0;   // ready for the next apply (ordered by the kernel boundary)
    if (p.bump && row == 0) *p.bump = *p.bump + 1;
  }
  dz_q_from_row(s_row, p.A, p.K, p.val_off, p.support, p.q_out + row * p.A,
                p.greedy_out ? p.greedy_out + row : nullptr, p.vmax_out ? p.vmax_out + row : nullptr,
                s_q, &s_best, &s_arg);
}

// rlax.q_learning / double_q_learning + clip_gradient + l2_loss (+ IS weights):
//   td = r + g * q_target[a*] - q_tm1[a],  a* = argmax(selector)
//   loss = mean(0.5 td^2 w);  d loss / d q_tm1[a] = -clip(w td / B, +-bound)
// ref: dqn/agent.py:94-106, double_q/agent.py:97-111, prioritized/agent.py:98-113.
// One thread per sample.  out rows: group g at (g*B + b)*ld.
__global__ void td_loss_kernel(const float* __restrict__ out, int ld, int B, int A,
                               int sel_group, int tgt_group,
                               const int64_t* __restrict__ a_tm1,
                               const double* __restrict__ r_t,
                               const double* __restrict__ d_t,
                               const float* __restrict__ weights, float bound,
                               float* __restrict__ dout, float* __restrict__ td_out,
                               float* __restrict__ prio_out) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const float* q0 = out + (long)b * ld;
  const float* qs = out + (long)(sel_group * B + b) * ld;
  const float* qt = out + (long)(tgt_group * B + b) * ld;
  int a_star = 0;
  float best = qs[0];
  for (int a = 1; a < A; ++a)
    if (qs[a] > best) { best = qs[a]; a_star = a; }
  const int a0 = (int)a_tm1[b];
  const float td = ((float)r_t[b] + (float)d_t[b] * qt[a_star]) - q0[a0];
  const float w = weights ? weights[b] : 1.0f;
  float g = td * w / (float)B;
  g = fminf(fmaxf(g, -bound), bound);
  float* d = dout + (long)b * ld;
  for (int a = 0; a < A; ++a) d[a] = (a == a0) ? -g : 0.f;
  td_out[b] = td;
  if (prio_out) prio_out[b] = fabsf(td);  // prioritized/agent.py:202
}

// vmap(rlax.quantile_q_learning), no double-Q (qrdqn/agent.py:98-107).
// dist layout [N][A] (quantile-major, networks.py:308).  One block per sample,
// thread i owns quantile theta_i.  The per-action means and the dout row are
// formed by coalesced sweeps over the contiguous N*A row (a fixed thread->element
// map and a fixed-order combine: deterministic sums, first-maximum argmax).
__global__ __launch_bounds__(1024) void quantile_loss_kernel(
    const float* __restrict__ out, int ld, int B, int A, int N, int sel_group,
    int tgt_group, const float* __restrict__ tau, const int64_t* __restrict__ a_tm1,
    const double* __restrict__ r_t, const double* __restrict__ d_t, float kappa,
    float* __restrict__ dout, float* __restrict__ losses) {
  // 1024 threads: quantile i = t % 256, the N targets split over the 4 quarters
  // jq = t / 256 (the N x N pair loop is the cost: 40 k pairs per sample at N = 201)
  __shared__ float s_t[256];
  __shared__ float s_g[256];
  __shared__ float s_part[256];
  __shared__ float s_li[4][256];
  __shared__ float s_gi[4][256];
  __shared__ float s_red[4];
  __shared__ int s_astar;
  const int b = blockIdx.x, t = threadIdx.x, i = t & 255, jq = t >> 8;
  const float* ds = out + (long)(sel_group * B + b) * ld;
  const float* dt = out + (long)(tgt_group * B + b) * ld;
  const float* d0 = out + (long)b * ld;
  // a* = argmax_a mean_n dist_sel[n][a]: thread t < GA*A sums elements
  // t, t + GA*A, ... (all of action t % A), then thread a folds its GA partials.
  const int GA = 256 / A;         // row groups sweeping in parallel (A <= 256)
  const int span = GA * A;
  if (t < 256) {
    float part = 0.f;
    if (t < span)
      for (int e = t; e < N * A; e += span) part += ds[e];
    s_part[t] = part;
  }
  __syncthreads();
  if (t < A) {
    float sum = 0.f;
    for (int g = 0; g < GA; ++g) sum += s_part[g * A + t];
    s_t[t] = sum / (float)N;
  }
  __syncthreads();
  if (t == 0) {
    float best = -__builtin_inff();
    int arg = 0;
    for (int a = 0; a < A; ++a)
      if (s_t[a] > best) { best = s_t[a]; arg = a; }
    s_astar = arg;
  }
  __syncthreads();
  const int a_star = s_astar, a0 = (int)a_tm1[b];
  const float r = (float)r_t[b], g = (float)d_t[b];
  if (t < N) s_t[t] = r + g * dt[t * A + a_star];
  __syncthreads();
  float li = 0.f, gi = 0.f;
  if (i < N) {
    const float theta = d0[i * A + a0], ti = tau[i];
    const int per = (N + 3) / 4;
    const int j1 = min(N, (jq + 1) * per);
    for (int j = jq * per; j < j1; ++j) {
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
  }
  s_li[jq][i] = li;
  s_gi[jq][i] = gi;
  __syncthreads();
  if (t < 256) {
    li = ((s_li[0][t] + s_li[1][t]) + (s_li[2][t] + s_li[3][t])) / (float)N;   // mean over targets
    gi = -((s_gi[0][t] + s_gi[1][t]) + (s_gi[2][t] + s_gi[3][t])) / (float)N;  // d loss_i / d theta_i
    s_g[t] = gi / (float)B;     // scaled for the batch mean
    const float s = wave_sum(li);  // loss = sum_i mean_j
    if ((t & 63) == 0) s_red[t >> 6] = s;
  }
  __syncthreads();
  if (t == 0) losses[b] = (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]);
  float* d = dout + (long)b * ld;
  for (int e = t; e < N * A; e += 1024) {
    const int q = e / A;
    d[e] = (e - q * A == a0) ? s_g[q] : 0.f;
  }
}

// optax.rmsprop(lr, decay, eps, centered=True) + apply_updates
// (dqn/run_atari.py:205-210): eps INSIDE the sqrt, no bias correction.
__global__ __launch_bounds__(256) void rmsprop_kernel(
    float* __restrict__ p, const float* __restrict__ g, float* __restrict__ mu,
    float* __restrict__ nu, long n4, float lr, float decay, float eps) {
  // software-pipelined like adam_kernel<1>: the next element's loads (clamped,
  // unconditional) are issued before the current element's arithmetic
  const long stride = (long)gridDim.x * 256;
  long i = (long)blockIdx.x * 256 + threadIdx.x;
  long ic = min(i, n4 - 1);
  float4 gv = ((const float4*)g)[ic], mv = ((const float4*)mu)[ic];
  float4 vv = ((const float4*)nu)[ic], pv = ((const float4*)p)[ic];
  while (i < n4) {
    const long inext = i + stride;
    ic = min(inext, n4 - 1);
    const float4 gn = ((const float4*)g)[ic], mn = ((const float4*)mu)[ic];
    const float4 vn = ((const float4*)nu)[ic], pn = ((const float4*)p)[ic];
    float* G = (float*)&gv; float* M = (float*)&mv; float* V = (float*)&vv;
    float* P = (float*)&pv;
#pragma unroll
    for (int j = 0; j < 4; ++j) dz_rms_one(G[j], M[j], V[j], P[j], lr, decay, eps);
    ((float4*)mu)[i] = mv; ((float4*)nu)[i] = vv; ((float4*)p)[i] = pv;
    gv = gn; mv = mn; vv = vn; pv = pn; i = inext;
  }
}

// Whole Q head for narrow outputs (N <= 32: DQN / double-Q / prioritized, i.e. one value
// per action) in ONE launch, one workgroup per sample b over its G rows g*B + b:
//   h1  = relu(sum of the fc1 split-K slabs + b1)       (networks.py:117-119; written
//         out for the backward pass)                     -- was fc_epilogue_kernel
//   out = h1 . W2 + b2 (vector or shared scalar bias)    -- was FcFwdOp + fc_epilogue_kernel
//   then the TD loss of td_loss_kernel (mode 1) or q-values / greedy action (mode 2).
// At 512 x N the second layer is 3 x 512 x N MACs per sample: a 4-wave tile GEMM
// launch plus its split-K epilogue were two launch floors (~4.5 us each) for < 1 us
// of arithmetic.  The slab sums reproduce fc_epilogue_kernel's order exactly
// (four interleaved chains (v0 + v1) + (v2 + v3)), so h1 is bit-identical.
struct DenseHeadParams {
  const float* part; int S; int rows;   // fc1 slabs [S][rows][512], rows = G*B
  int B; int G;
  const float* prm[3];                  // parameter set of each group
  long fc1_b, fc2_w, fc2_b; int ld2;    // offsets into prm; W2 is [512][ld2]
  int N; int bias_shared;
  float* h1;                            // [rows][512]
  float* out;                           // [rows][ld2]
  int mode;                             // 0: outputs only, 1: TD loss, 2: acting
  int sel_group, tgt_group;
  const int64_t* a_tm1; const double* r_t; const double* d_t; const float* weights;
  float bound;
  float* dout; float* td_out; float* prio_out;
  float* dout_rowsum;                   // optional [B]: sum of sample b's dout row (= -g)
  float* dh1;                           // optional [B][512]: d loss / d (fc1 pre-activation)
  float* q_values; int32_t* greedy; float* vmax;
};

__global__ __launch_bounds__(512) void dense_head_kernel(DenseHeadParams p) {
  constexpr int KH = 512, MAXS = 32, NMAX = 32, KS = 16, KPS = KH / KS;
  __shared__ float s_h[3][KH];
  __shared__ float s_red[KS][3][NMAX];
  __shared__ float s_out[3][NMAX];
  __shared__ float s_w[KH];   // W2_online[:, a_tm1]
  __shared__ float s_g;
  const int t = threadIdx.x;
  const int b = blockIdx.x;
  // (1) every load of the kernel is issued before the first use, in order of use:
  //     the slabs of all rows, the fc1 biases, the second-layer weights, the loss scalars
  float x[3][MAXS];
#pragma unroll
  for (int g = 0; g < 3; ++g) {
    const int r = min(g, p.G - 1) * p.B + b;
#pragma unroll
    for (int s = 0; s < MAXS; ++s)
      x[g][s] = p.part[((long)min(s, p.S - 1) * p.rows + r) * KH + t];
  }
  float b1[3];
#pragma unroll
  for (int g = 0; g < 3; ++g) b1[g] = dz_pick3(p.prm, min(g, p.G - 1))[p.fc1_b + t];
  // second layer: thread (ks, n) = (t / 32, t % 32) owns k in [ks*32, ks*32+32) of column n;
  // the parameter sets of group 0/2 (online) and group 1 (target)
  const int n = t & 31, ks = t >> 5, nc = min(n, p.N - 1);
  float w_on[KPS], w_tg[KPS];
  {
    const float* w0 = p.prm[0] + p.fc2_w + (long)(ks * KPS) * p.ld2 + nc;
    const float* w1 = p.prm[1] + p.fc2_w + (long)(ks * KPS) * p.ld2 + nc;
#pragma unroll
    for (int k = 0; k < KPS; ++k) { w_on[k] = w0[(long)k * p.ld2]; w_tg[k] = w1[(long)k * p.ld2]; }
  }
  const float b2_on = p.prm[0][p.fc2_b + (p.bias_shared ? 0 : nc)];  // networks.py:120-134
  const float b2_tg = p.prm[1][p.fc2_b + (p.bias_shared ? 0 : nc)];
  int a0 = 0; float r_t = 0.f, d_t = 0.f, wt = 1.f;
  if (p.mode == 1) {   // uniform branch; same addresses for every lane
    a0 = (int)p.a_tm1[b]; r_t = (float)p.r_t[b]; d_t = (float)p.d_t[b];
    wt = p.weights ? p.weights[b] : 1.0f;
  }
  __builtin_amdgcn_sched_barrier(0);

  // (2) h1 = relu(slab sum + b1), in fc_epilogue_kernel's order
#pragma unroll
  for (int g = 0; g < 3; ++g) {
    if (g < p.G) {
      float v[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < MAXS; ++s) v[s & 3] += s < p.S ? x[g][s] : 0.f;
      float h = ((v[0] + v[1]) + (v[2] + v[3])) + b1[g];
      h = h > 0.f ? h : 0.f;
      p.h1[(long)(g * p.B + b) * KH + t] = h;
      s_h[g][t] = h;
    }
  }
  __syncthreads();
  // (3) out[g][n] = sum_k h1[g][k] W2[k][n] + b2: 16 k-slices, folded in slice order
#pragma unroll
  for (int g = 0; g < 3; ++g) {
    if (g < p.G) {
      float acc = 0.f;
      const float* h = &s_h[g][ks * KPS];
#pragma unroll
      for (int k = 0; k < KPS; ++k) acc = __builtin_fmaf(h[k], g == 1 ? w_tg[k] : w_on[k], acc);
      s_red[ks][g][n] = acc;
    }
  }
  if (p.mode == 1 && p.dh1 && n == a0) {
#pragma unroll
    for (int k = 0; k < KPS; ++k) s_w[ks * KPS + k] = w_on[k];
  }
  __syncthreads();
  if (t < 3 * NMAX) {
    const int g = t >> 5;
    if (g < p.G && n < p.N) {
      float v = s_red[0][g][n];
#pragma unroll
      for (int k2 = 1; k2 < KS; ++k2) v += s_red[k2][g][n];
      v += g == 1 ? b2_tg : b2_on;
      s_out[g][n] = v;
      p.out[(long)(g * p.B + b) * p.ld2 + n] = v;
    }
  }
  if (p.mode == 0) return;
  __syncthreads();
  if (p.mode == 1 && p.dh1) {
    // the second layer's input gradient on the spot: dout has ONE non-zero per sample
    // (-g at a_tm1), so dh1[b][k] = relu'(h1) * (-g) * W2[k][a_tm1] (what FcDgradOp
    // computes from the dout row, the other terms being exact zeros)
    if (t == 0) {
      const float* qs = s_out[p.sel_group];
      int a_star = 0;
      float best = qs[0];
      for (int a = 1; a < p.N; ++a)
        if (qs[a] > best) { best = qs[a]; a_star = a; }
      const float td = (r_t + d_t * s_out[p.tgt_group][a_star]) - s_out[0][a0];
      float g = td * wt / (float)p.B;
      s_g = fminf(fmaxf(g, -p.bound), p.bound);
    }
    __syncthreads();
    const float v = (-s_g) * s_w[t];
    p.dh1[(long)b * KH + t] = s_h[0][t] > 0.f ? v : 0.f;
  }
  if (t != 0) return;
  if (p.mode == 1) {  // td_loss_kernel's arithmetic for sample b
    const float* q0 = s_out[0];
    const float* qs = s_out[p.sel_group];
    const float* qt = s_out[p.tgt_group];
    int a_star = 0;
    float best = qs[0];
    for (int a = 1; a < p.N; ++a)
      if (qs[a] > best) { best = qs[a]; a_star = a; }
    const float td = (r_t + d_t * qt[a_star]) - q0[a0];
    float g = td * wt / (float)p.B;
    g = fminf(fmaxf(g, -p.bound), p.bound);
    float* d = p.dout + (long)b * p.ld2;
    for (int a = 0; a < p.N; ++a) d[a] = (a == a0) ? -g : 0.f;
    p.td_out[b] = td;
    if (p.prio_out) p.prio_out[b] = fabsf(td);
    if (p.dout_rowsum) p.dout_rowsum[b] = -g;
  } else {            // dense_q_values_kernel's outputs for row b
    int arg = 0;
    float best = s_out[0][0];
    for (int a = 0; a < p.N; ++a) {
      const float q = s_out[0][a];
      if (p.q_values) p.q_values[b * p.N + a] = q;
      if (q > best) { best = q; arg = a; }
    }
    if (p.greedy) p.greedy[b] = arg;
    if (p.vmax) p.vmax[b] = best;
  }
}

// q_values of a plain Q head + greedy action + max (dqn/agent.py:121-131)
__global__ void dense_q_values_kernel(const float* __restrict__ out, int ld, int B, int A,
                                      float* __restrict__ q_out,
                                      int32_t* __restrict__ greedy_out,
                                      float* __restrict__ vmax_out) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const float* q = out + (long)b * ld;
  int arg = 0;
  float best = q[0];
  for (int a = 0; a < A; ++a) {
    q_out[b * A + a] = q[a];
    if (q[a] > best) { best = q[a]; arg = a; }
  }
  if (greedy_out) greedy_out[b] = arg;
  if (vmax_out) vmax_out[b] = best;
}

}  // namespace
