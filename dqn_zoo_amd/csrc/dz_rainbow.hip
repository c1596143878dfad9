// Rainbow learner step on one MI355X (ref: rainbow/agent.py:85-121).
//
// Launch sequence (all on the caller's stream, no host synchronisation):
//   forward : conv1 conv2 conv3 (3 applies batched as groups) -> fc1 (noisy,
//             adv1|val1 fused, split-K) -> epilogue -> fc2 (noisy adv2, val2)
//             -> epilogue -> head/loss (dueling + softmax + double-Q selector +
//             Cramer projection + cross-entropy + dlogits + priorities)
//   backward: fc2 wgrad/dgrad, fc1 wgrad/dgrad, conv3 wgrad/dgrad, conv2
//             wgrad/dgrad, conv1 wgrad, bias column sums
//   update  : sum of squares -> global norm/clip scale -> Adam
// Roofline notes per kernel are in DESIGN.md.
#include "dz_fc_stream.h"

namespace {

constexpr int kFlat = 3136;   // 7*7*64 torso features
constexpr int kHid = 512;
constexpr int kG = 3;         // applies: online(s_tm1), online(s_t), target(s_t)
constexpr int kS_fc1 = 7;     // grid split-K factors
constexpr int kS_fc2 = 4;
constexpr int kS_dh1 = 5;
constexpr int kS_dfeat = 8;
constexpr int kS_cw1 = 50, kS_cw2 = 27, kS_cw3 = 14;
constexpr int kNormBlocks = 512;

inline int64_t align4(int64_t v) { return (v + 3) & ~(int64_t)3; }

// Run-time tuning knobs (dz_set_tuning): kernel variant and split factors, used
// by tools/tune.py to sweep configurations in ONE GPU session.
constexpr int kMaxSplitFc1 = 32;
int g_fc1_variant = 9;   // 8/9 = weight-streaming kernels (dz_fc_stream.h)
int g_fc1_splits = 32;
int g_fc1_dgrad_stream = 0;  // measured: tile-GEMM 26 us vs streaming 35 us
int g_fc1_blocked_experiment = 0;
int g_overlap = 1;           // weight gradients on an auxiliary stream
hipStream_t g_aux_stream = nullptr;
hipEvent_t g_ev[5];

int ensure_aux() {
  if (g_aux_stream) return DZ_OK;
  DZ_HIP_CHECK(hipStreamCreateWithFlags(&g_aux_stream, hipStreamNonBlocking));
  for (int i = 0; i < 5; ++i)
    DZ_HIP_CHECK(hipEventCreateWithFlags(&g_ev[i], hipEventDisableTiming));
  return DZ_OK;
}

// conv geometries (networks.py:194-198)
//                      U8  H   W   C  KS S  OH  OW  CO
//                                                       WM WN WK KT
using Conv1Fwd = ConvFwdOp<1, 84, 84, 4, 8, 4, 20, 20, 32, 4, 1, 1, 4>;
using Conv2Fwd = ConvFwdOp<0, 20, 20, 32, 4, 2, 9, 9, 64, 1, 2, 2, 4>;
using Conv3Fwd = ConvFwdOp<0, 9, 9, 64, 3, 1, 7, 7, 64, 1, 2, 2, 3>;
using Conv1Wg = ConvWgradOp<1, 84, 84, 4, 8, 4, 20, 20, 32, 2, 1, 2, 2>;
using Conv2Wg = ConvWgradOp<0, 20, 20, 32, 4, 2, 9, 9, 64, 2, 2, 1, 2>;
using Conv3Wg = ConvWgradOp<0, 9, 9, 64, 3, 1, 7, 7, 64, 2, 2, 1, 2>;
using Conv2Dg = ConvDgradOp<20, 20, 32, 4, 2, 9, 9, 64, 1, 1, 4, 1>;
using Conv3Dg = ConvDgradOp<9, 9, 64, 3, 1, 7, 7, 64, 1, 1, 4, 3>;
using FcFwd = FcFwdOp<1, 2, 2, 4>;
using FcDg = FcDgradOp<1, 2, 2, 4>;
using FcWg = FcWgradOp<2, 2, 1, 2>;

// ---- small kernels ----------------------------------------------------------

// out[r][c] = act( sum_s part[s][r][c] + b_mu[c] + b_sig[c]*eps_out[g][c] )
// block = 64 columns x 4 waves striding over the split slabs (LDS combine).
__global__ __launch_bounds__(256) void fc_epilogue_kernel(
    const float* __restrict__ part, int S, int rows, int cols, int ld,
    int rows_per_group, const float* p0, const float* p1, const float* p2, long b_mu,
    long b_sig, const float* n0, const float* n1, const float* n2, int eps_out, int relu,
    float* __restrict__ out) {
  __shared__ float red[4][64];
  const int l = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + l;
  const int r = blockIdx.y;
  float v = 0.f;
  if (c < cols)
    for (int s = w; s < S; s += 4) v += part[((long)s * rows + r) * ld + c];
  red[w][l] = v;
  __syncthreads();
  if (w != 0 || c >= cols) return;
  v = (red[0][l] + red[1][l]) + (red[2][l] + red[3][l]);
  const int g = r / rows_per_group;
  const float* prm = g == 0 ? p0 : (g == 1 ? p1 : p2);
  const float* nz = g == 0 ? n0 : (g == 1 ? n1 : n2);
  if (b_mu >= 0) v += prm[b_mu + c];
  if (b_sig >= 0) v += prm[b_sig + c] * nz[eps_out + c];
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
  float v = 0.f;
  if (i < n)
    for (int s = w; s < S; s += 4) v += part[(long)s * n + i];
  red[w][l] = v;
  __syncthreads();
  if (w == 0 && i < n) {
    v = (red[0][l] + red[1][l]) + (red[2][l] + red[3][l]);
    if (mask && !(mask[i] > 0.f)) v = 0.f;
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
  float v = 0.f;
  if (c < jb.cols) {
#pragma unroll 4
    for (int r = w; r < jb.rows; r += 4) v += jb.m[(long)r * jb.ld + c];
  }
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
struct FinalizeJobs {
  ReduceJob r[3];
  unsigned r_end[3];      // exclusive prefix of 64-wide tiles
  ColsumJob c[2];
  unsigned c_tiles[2];    // tiles per colsum job
};
__global__ __launch_bounds__(256) void finalize_grads_kernel(FinalizeJobs J) {
  __shared__ float red[4][64];
  const int l = threadIdx.x & 63, w = threadIdx.x >> 6;
  unsigned b = blockIdx.x;
  float v = 0.f;
  if (b < J.r_end[2]) {
    const int j = b < J.r_end[0] ? 0 : (b < J.r_end[1] ? 1 : 2);
    const ReduceJob jb = J.r[j];
    const long i = (long)(b - (j ? J.r_end[j - 1] : 0)) * 64 + l;
    if (i < jb.n)
      for (int s = w; s < jb.S; s += 4) v += jb.part[(long)s * jb.n + i];
    red[w][l] = v;
    __syncthreads();
    if (w == 0 && i < jb.n)
      jb.out[i] = (red[0][l] + red[1][l]) + (red[2][l] + red[3][l]);
    return;
  }
  b -= J.r_end[2];
  const int j = b < J.c_tiles[0] ? 0 : 1;
  const ColsumJob jb = J.c[j];
  const int c = (int)(b - (j ? J.c_tiles[0] : 0)) * 64 + l;
  if (c < jb.cols) {
#pragma unroll 4
    for (int r = w; r < jb.rows; r += 4) v += jb.m[(long)r * jb.ld + c];
  }
  red[w][l] = v;
  __syncthreads();
  if (w == 0 && c < jb.cols) {
    const float s = (red[0][l] + red[1][l]) + (red[2][l] + red[3][l]);
    if (jb.out) jb.out[c] = s;
    if (jb.out_scaled) jb.out_scaled[c] = s * jb.scale[c];
  }
}

__device__ __forceinline__ float wave_max(float v) {
  for (int o = 32; o >= 1; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}
__device__ __forceinline__ float wave_sum(float v) {
  for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// One wave per sample; lane k owns atom k (K <= 64).
// ref: networks.py:254-258 (dueling, softmax, expectation),
//      rainbow/agent.py:97-109 + rlax.categorical_double_q_learning,
//      rainbow/agent.py:194 (priorities).
__global__ __launch_bounds__(64) void rainbow_head_loss_kernel(
    const float* __restrict__ fc2_out, int ld, int val_off, int B, int A, int K,
    const int64_t* __restrict__ a_tm1, const double* __restrict__ r_t,
    const double* __restrict__ d_t, const float* __restrict__ weights,
    const float* __restrict__ support, float* __restrict__ dout2,
    float* __restrict__ losses, float* __restrict__ priorities,
    float* __restrict__ q_sel_out, float* __restrict__ target_out) {
  __shared__ float s_p[64];
  __shared__ float s_z[64];
  const int b = blockIdx.x, k = threadIdx.x;
  const bool on = k < K;
  const int NA = val_off;  // value-head columns start at the padded offset
  const float z = on ? support[k] : 0.f;
  const float invA = 1.0f / (float)A;

  // ---- group 1: q_values of online(s_t) -> argmax (double-Q selector) ----
  const float* o1 = fc2_out + (long)(1 * B + b) * ld;
  float mean_adv = 0.f;
  for (int a = 0; a < A; ++a) mean_adv += on ? o1[a * K + k] : 0.f;
  mean_adv /= (float)A;
  const float v1 = on ? o1[NA + k] : 0.f;
  float best_q = -__builtin_inff();
  int a_star = 0;
  for (int a = 0; a < A; ++a) {
    const float lg = on ? (v1 + o1[a * K + k] - mean_adv) : -__builtin_inff();
    const float mx = wave_max(lg);
    const float e = on ? expf(lg - mx) : 0.f;
    const float sm = wave_sum(e);
    const float q = wave_sum((e / sm) * z);
    if (k == 0 && q_sel_out) q_sel_out[b * A + a] = q;
    if (q > best_q) { best_q = q; a_star = a; }  // first maximum, as jnp.argmax
  }
  // ---- group 2: target distribution of the selected action ----
  const float* o2 = fc2_out + (long)(2 * B + b) * ld;
  float mean2 = 0.f;
  for (int a = 0; a < A; ++a) mean2 += on ? o2[a * K + k] : 0.f;
  mean2 /= (float)A;
  const float lg2 = on ? (o2[NA + k] + o2[a_star * K + k] - mean2) : -__builtin_inff();
  const float mx2 = wave_max(lg2);
  const float e2 = on ? expf(lg2 - mx2) : 0.f;
  const float p_t = e2 / wave_sum(e2);
  // ---- Cramer projection of (r + g z, p_t) onto the support ----
  const float r = (float)r_t[b], g = (float)d_t[b];  // f64 -> f32 at the jit boundary
  const float vmin = support[0], vmax = support[K - 1];
  float zp = r + g * z;
  zp = fminf(fmaxf(zp, vmin), vmax);
  s_p[k] = on ? p_t : 0.f;
  s_z[k] = zp;
  __syncthreads();
  float m = 0.f;
  if (on) {
    const float zq = z;
    const float dpos = (k + 1 < K ? support[k + 1] : support[0]) - zq;
    const float dneg = zq - (k > 0 ? support[k - 1] : support[K - 1]);
    const float rpos = dpos > 0.f ? 1.0f / dpos : 0.f;
    const float rneg = dneg > 0.f ? 1.0f / dneg : 0.f;
    for (int j = 0; j < K; ++j) {
      const float delta = s_z[j] - zq;
      const float dh = delta >= 0.f ? delta * rpos : -(delta * rneg);
      const float c = fminf(fmaxf(1.0f - dh, 0.f), 1.f);
      m += c * s_p[j];
    }
  }
  if (target_out && on) target_out[b * K + k] = m;
  // ---- group 0: cross-entropy with log_softmax(logits_tm1[a_tm1]) ----
  const int a0 = (int)a_tm1[b];
  const float* o0 = fc2_out + (long)(0 * B + b) * ld;
  float mean0 = 0.f;
  for (int a = 0; a < A; ++a) mean0 += on ? o0[a * K + k] : 0.f;
  mean0 /= (float)A;
  const float lg0 = on ? (o0[NA + k] + o0[a0 * K + k] - mean0) : -__builtin_inff();
  const float mx0 = wave_max(lg0);
  const float sh = lg0 - mx0;
  const float e0 = on ? expf(sh) : 0.f;
  const float se0 = wave_sum(e0);
  const float lsm = sh - logf(se0);
  const float loss = -wave_sum(on ? m * lsm : 0.f);
  const float msum = wave_sum(m);
  // d loss / d logits_tm1[a0][k], scaled by w/B (loss = mean(losses*w))
  const float gk = on ? ((e0 / se0) * msum - m) * (weights[b] / (float)B) : 0.f;
  if (on) {
    float* d = dout2 + (long)b * ld;
    for (int a = 0; a < A; ++a)  // dadv[a][k] = G[a][k] - mean_a G[.][k]
      d[a * K + k] = (a == a0 ? gk : 0.f) - gk * invA;
    d[NA + k] = gk;              // dval[k] = sum_a G[a][k]
  }
  if (k == 0) {
    losses[b] = loss;
    priorities[b] = fminf(fmaxf(fabsf(loss), 0.f), 100.f);
  }
}

__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ g,
                                                    long n, float* __restrict__ part) {
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
  if (threadIdx.x == 0) part[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// One block: global norm, clip decision, Adam bias corrections, mean loss.
// (A single-launch "last workgroup finishes" form with agent-scope fences was
// measured slower than this second tiny launch: 20 us vs 9 + 8 us.)
__global__ __launch_bounds__(256) void opt_scalars_kernel(
    const float* __restrict__ part, int nparts, int32_t* count, float b1, float b2,
    float max_norm, const float* __restrict__ losses, const float* __restrict__ weights,
    int B, float* __restrict__ sc) {
  __shared__ float red[4];
  float s = 0.f;
  for (int i = threadIdx.x; i < nparts; i += 256) s += part[i];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    const float gn = sqrtf((red[0] + red[1]) + (red[2] + red[3]));
    const int c = *count + 1;  // optax: count_inc = count + 1
    *count = c;
    sc[DZ_SC_GNORM] = gn;
    sc[DZ_SC_BC1] = 1.0f - powf(b1, (float)c);
    sc[DZ_SC_BC2] = 1.0f - powf(b2, (float)c);
    sc[DZ_SC_CLIP] = (max_norm > 0.f && !(gn < max_norm)) ? 0.f : 1.f;
    float l = 0.f;
    for (int i = 0; i < B; ++i) l += losses[i] * weights[i];
    sc[DZ_SC_LOSS] = l / (float)B;
  }
}

// optax.clip_by_global_norm then optax.adam, then apply_updates.
__global__ __launch_bounds__(256) void adam_kernel(
    float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
    float* __restrict__ v, long n4, const float* __restrict__ sc, float lr, float b1,
    float b2, float eps, float max_norm) {
  const float gn = sc[DZ_SC_GNORM], bc1 = sc[DZ_SC_BC1], bc2 = sc[DZ_SC_BC2];
  const bool pass = sc[DZ_SC_CLIP] != 0.f;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    float4 gv = ((const float4*)g)[i];
    float4 mv = ((float4*)m)[i], vv = ((float4*)v)[i], pv = ((float4*)p)[i];
    float* G = (float*)&gv; float* M = (float*)&mv; float* V = (float*)&vv;
    float* P = (float*)&pv;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float gj = pass ? G[j] : (G[j] / gn) * max_norm;
      M[j] = (1.0f - b1) * gj + b1 * M[j];
      V[j] = (1.0f - b2) * (gj * gj) + b2 * V[j];
      const float upd = (M[j] / bc1) / (sqrtf(V[j] / bc2) + eps);
      P[j] = P[j] + (-lr) * upd;
    }
    ((float4*)m)[i] = mv; ((float4*)v)[i] = vv; ((float4*)p)[i] = pv;
  }
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
__global__ void noise_fill_kernel(float* __restrict__ out, long n, uint64_t seed,
                                  uint64_t counter, const int32_t* __restrict__ step) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (step) counter += (uint64_t)(*step) * (uint64_t)n;  // per-step stream offset
  const uint64_t h = mix64(mix64(seed) ^ mix64(counter + (uint64_t)i));
  // jax.random.truncated_normal: sqrt2 * erfinv(U(erf(lo/sqrt2), erf(hi/sqrt2)))
  const float u01 = ((float)(h >> 40) + 0.5f) * (1.0f / 16777216.0f);
  const float e = 0.9544997361036416f;  // erf(2/sqrt(2))
  const float u = (2.0f * u01 - 1.0f) * e;
  float x = 1.4142135623730951f * erfinvf(u);
  x = fminf(fmaxf(x, -2.0f), 2.0f);
  const float s = sqrtf(fabsf(x));
  out[i] = x < 0.f ? -s : (x > 0.f ? s : 0.f);  // sign(x) * sqrt|x| (networks.py:144)
}

// q_values[b][a] = sum_k softmax(v + adv_a - mean_a adv)[k] * support[k]
// (ref: networks.py:254-258); also the greedy action and its value
// (ref: rainbow/agent.py:125-131, first maximum).
__global__ __launch_bounds__(64) void rainbow_q_values_kernel(
    const float* __restrict__ fc2_out, int ld, int val_off, int A, int K,
    const float* __restrict__ support, float* __restrict__ q_out,
    int32_t* __restrict__ greedy_out, float* __restrict__ vmax_out) {
  const int b = blockIdx.x, k = threadIdx.x;
  const bool on = k < K;
  const float z = on ? support[k] : 0.f;
  const float* o = fc2_out + (long)b * ld;
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

}  // namespace


// The network apply for G groups (ref: networks.py:224-253): conv torso, fused
// noisy fc1 (adv1|val1), noisy fc2 (adv2, val2) into ws_fc2_out rows [G*B].
struct FwdHeads { FcHead fc1h[2]; FcHead fc2h[2]; };
static int rainbow_forward(const dz_rainbow_layout_t& L, const FwdHeads& H, int G, int B,
                           const float* const* prm, const float* const* nz,
                           const uint8_t* const* in, float* ws, hipStream_t s) {
  int rc = DZ_OK;
  const int NA = L.num_actions * L.num_atoms;
  const int ld2 = L.adv2_ld + L.val2_ld;
  const FcHead* fc1h = H.fc1h;
  const FcHead* fc2h = H.fc2h;
  (void)NA;
    {  // conv1: uint8 states -> act1, u8->f32 /255 fused into the A-tile load
    ConvFwdParams p;
    for (int g = 0; g < G; ++g) p.in[g] = in[g];
    for (int g = 0; g < G; ++g) {
      p.in_img_base[g] = 0; p.w[g] = prm[g] + L.conv_w[0]; p.bias[g] = prm[g] + L.conv_b[0];
    }
    p.out = ws + L.ws_act1; p.B = B; p.G = G;
    rc = dz_launch_gemm<Conv1Fwd>(p, dim3(1, G * Conv1Fwd::tiles_per_group(B)), s);
    if (rc) return rc;
    DZ_PROF(s, "conv1_fwd");
  }
  {  // conv2
    ConvFwdParams p;
    for (int g = 0; g < G; ++g) {
      p.in[g] = ws + L.ws_act1; p.in_img_base[g] = g * B; p.w[g] = prm[g] + L.conv_w[1]; p.bias[g] = prm[g] + L.conv_b[1];
    }
    p.out = ws + L.ws_act2; p.B = B; p.G = G;
    rc = dz_launch_gemm<Conv2Fwd>(p, dim3(1, G * Conv2Fwd::tiles_per_group(B)), s);
    if (rc) return rc;
    DZ_PROF(s, "conv2_fwd");
  }
  {  // conv3 (+ flatten: NHWC rows are already (h,w,c) order)
    ConvFwdParams p;
    for (int g = 0; g < G; ++g) {
      p.in[g] = ws + L.ws_act2; p.in_img_base[g] = g * B; p.w[g] = prm[g] + L.conv_w[2]; p.bias[g] = prm[g] + L.conv_b[2];
    }
    p.out = ws + L.ws_feat; p.B = B; p.G = G;
    rc = dz_launch_gemm<Conv3Fwd>(p, dim3(1, G * Conv3Fwd::tiles_per_group(B)), s);
    if (rc) return rc;
    DZ_PROF(s, "conv3_fwd");
  }
  {  // fc1: noisy adv1 | val1, split-K partials
    FcFwdParams p;
    p.x = ws + L.ws_feat; p.ldx = kFlat; p.M = B; p.G = G; p.NH = 2; p.S = kS_fc1;
    p.noisy = 1;
    for (int g = 0; g < G; ++g) { p.params[g] = prm[g]; p.noise[g] = nz[g]; }
    p.head[0] = fc1h[0]; p.head[1] = fc1h[1];
    p.part = ws + L.ws_fc1_part; p.ldo = 1024;
    p.S = g_fc1_splits;
    const dim3 gz(1, (B + 31) / 32, G * 2 * g_fc1_splits);
    switch (g_fc1_variant) {
      default:
      case 0: rc = dz_launch_gemm<FcFwdOp<1, 2, 2, 4>>(p, dim3(8, gz.y, gz.z), s); break;
      case 1: rc = dz_launch_gemm<FcFwdOp<1, 2, 2, 2>>(p, dim3(8, gz.y, gz.z), s); break;
      case 2: rc = dz_launch_gemm<FcFwdOp<1, 2, 2, 1>>(p, dim3(8, gz.y, gz.z), s); break;
      case 3: rc = dz_launch_gemm<FcFwdOp<1, 4, 1, 4>>(p, dim3(4, gz.y, gz.z), s); break;
      case 4: rc = dz_launch_gemm<FcFwdOp<1, 4, 1, 2>>(p, dim3(4, gz.y, gz.z), s); break;
      case 5: rc = dz_launch_gemm<FcFwdOp<1, 1, 4, 2>>(p, dim3(16, gz.y, gz.z), s); break;
      case 6: rc = dz_launch_gemm<FcFwdOp<1, 1, 4, 1>>(p, dim3(16, gz.y, gz.z), s); break;
      case 7: rc = dz_launch_gemm<FcFwdOp<1, 4, 1, 1>>(p, dim3(4, gz.y, gz.z), s); break;
      case 8: {
        DZ_REQUIRE(B <= 32);
        FcStreamFwdParams q;
        q.x = p.x; q.ldx = p.ldx; q.M = B; q.G = G; q.NH = 2; q.S = g_fc1_splits;
        q.noisy = 1;
        for (int g = 0; g < G; ++g) { q.params[g] = prm[g]; q.noise[g] = nz[g]; }
        q.head[0] = fc1h[0]; q.head[1] = fc1h[1];
        q.part = p.part; q.ldo = p.ldo;
        hipLaunchKernelGGL(dz_fc_stream_fwd, dim3(8, G * g_fc1_splits), dim3(256), 0,
                           s, q);
        DZ_LAUNCH_CHECK();
        rc = DZ_OK;
        break;
      }
      case 9: {
        DZ_REQUIRE(B <= 32);
        FcStreamFwd2Params q;
        q.x = p.x; q.ldx = p.ldx; q.M = B; q.G = G; q.NH = 2; q.S = g_fc1_splits;
        q.noisy = 1;
        for (int g = 0; g < G; ++g) { q.params[g] = prm[g]; q.noise[g] = nz[g]; }
        q.head[0] = fc1h[0]; q.head[1] = fc1h[1];
        q.part = p.part; q.ldo = p.ldo;
        const int rtotal = 2 * kFlat;
        q.rows_per_split = ((rtotal + g_fc1_splits - 1) / g_fc1_splits + 3) & ~3;
        DZ_REQUIRE(q.rows_per_split <= DZ_FC2_MAX_ROWS);
        q.blocked = g_fc1_blocked_experiment;
        hipLaunchKernelGGL(dz_fc_stream_fwd2, dim3(8, G * g_fc1_splits), dim3(256),
                           (size_t)q.rows_per_split * 32 * sizeof(float), s, q);
        DZ_LAUNCH_CHECK();
        rc = DZ_OK;
        break;
      }
    }
    if (rc) return rc;
    DZ_PROF(s, "fc1_fwd");
    hipLaunchKernelGGL(fc_epilogue_kernel, dim3(16, G * B), dim3(256), 0, s,
                       ws + L.ws_fc1_part, g_fc1_splits, G * B, 1024, 1024, B,
                       prm[0], prm[1], prm[2], (long)L.fc1_mu_b, (long)L.fc1_sig_b,
                       nz[0], nz[1], nz[2], (int)L.n_fc1_out, 1, ws + L.ws_h1);
    DZ_LAUNCH_CHECK();
    DZ_PROF(s, "fc1_epilogue");
  }
  {  // fc2: noisy adv2 (no mu bias) and val2 (no mu bias)
    FcFwdParams p;
    p.x = ws + L.ws_h1; p.ldx = 1024; p.M = B; p.G = G; p.NH = 2; p.S = kS_fc2;
    p.noisy = 1;
    for (int g = 0; g < G; ++g) { p.params[g] = prm[g]; p.noise[g] = nz[g]; }
    p.head[0] = fc2h[0]; p.head[1] = fc2h[1];
    p.part = ws + L.ws_fc2_part; p.ldo = ld2;
    rc = dz_launch_gemm<FcFwd>(p, dim3((NA + FcFwd::BN - 1) / FcFwd::BN, (B + 31) / 32,
                                      G * 2 * kS_fc2), s);
    if (rc) return rc;
    DZ_PROF(s, "fc2_fwd");
    hipLaunchKernelGGL(fc_epilogue_kernel, dim3((ld2 + 63) / 64, G * B), dim3(256),
                       0, s, ws + L.ws_fc2_part, kS_fc2, G * B, ld2, ld2, B,
                       prm[0], prm[1], prm[2], (long)-1, (long)L.fc2_sig_b, nz[0],
                       nz[1], nz[2], (int)L.n_fc2_out, 0, ws + L.ws_fc2_out);
    DZ_LAUNCH_CHECK();
    DZ_PROF(s, "fc2_epilogue");
  }
  return rc;
}

static void make_heads(const dz_rainbow_layout_t& L, FwdHeads& H) {
  const int A = L.num_actions, K = L.num_atoms;
  const int NA = A * K, NAp = L.adv2_ld;
  FcHead* fc1h = H.fc1h;
  FcHead* fc2h = H.fc2h;
  for (int h = 0; h < 2; ++h) {
    fc1h[h].w_mu = L.fc1_mu_w + 512 * h; fc1h[h].w_sig = L.fc1_sig_w + 512 * h;
    fc1h[h].ldw = L.fc1_ld; fc1h[h].N = 512; fc1h[h].K = kFlat; fc1h[h].x_off = 0;
    fc1h[h].eps_in = (int)(h == 0 ? L.n_adv1_in : L.n_val1_in);
    fc1h[h].eps_out = (int)L.n_fc1_out + 512 * h; fc1h[h].out_off = 512 * h;
  }
  fc2h[0].w_mu = L.adv2_mu_w; fc2h[0].w_sig = L.adv2_sig_w; fc2h[0].ldw = L.adv2_ld;
  fc2h[0].N = NA; fc2h[0].K = kHid; fc2h[0].x_off = 0;
  fc2h[0].eps_in = (int)L.n_adv2_in; fc2h[0].eps_out = (int)L.n_fc2_out;
  fc2h[0].out_off = 0;
  fc2h[1].w_mu = L.val2_mu_w; fc2h[1].w_sig = L.val2_sig_w; fc2h[1].ldw = L.val2_ld;
  fc2h[1].N = K; fc2h[1].K = kHid; fc2h[1].x_off = 512;
  fc2h[1].eps_in = (int)L.n_val2_in; fc2h[1].eps_out = (int)L.n_fc2_out + NAp;
  fc2h[1].out_off = NAp;

}

// ---- layout -----------------------------------------------------------------
extern "C" int dz_rainbow_layout(int A, int K, int B, dz_rainbow_layout_t* L) {
  DZ_REQUIRE(L && A > 0 && K > 0 && K <= 64 && B > 0 && B <= 1024);
  const int NA = A * K;
  L->num_actions = A; L->num_atoms = K; L->batch = B; L->groups = kG;
  int64_t o = 0;
  const int64_t cw[3] = {256 * 32, 512 * 64, 576 * 64};
  const int64_t cb[3] = {32, 64, 64};
  for (int i = 0; i < 3; ++i) {
    L->conv_w[i] = o; o = align4(o + cw[i]);
    L->conv_b[i] = o; o = align4(o + cb[i]);
  }
  // fc1 rows are padded by 32 floats: a 4096-byte row pitch maps every row of a
  // column tile to the same HBM channel group (measured 3x slower streaming)
  L->fc1_ld = 1024 + 32;
  L->fc1_mu_w = o; o = align4(o + (int64_t)kFlat * L->fc1_ld);
  L->fc1_mu_b = o; o = align4(o + 1024);
  L->fc1_sig_w = o; o = align4(o + (int64_t)kFlat * L->fc1_ld);
  L->fc1_sig_b = o; o = align4(o + 1024);
  // fc2 matrices use a leading dimension padded to 4 floats so that every row
  // is 16-byte aligned (pad columns are zero and receive zero gradients)
  L->adv2_ld = (int32_t)align4(NA); L->val2_ld = (int32_t)align4(K);
  L->adv2_mu_w = o; o = align4(o + (int64_t)kHid * L->adv2_ld);
  L->adv2_sig_w = o; o = align4(o + (int64_t)kHid * L->adv2_ld);
  L->val2_mu_w = o; o = align4(o + (int64_t)kHid * L->val2_ld);
  L->val2_sig_w = o; o = align4(o + (int64_t)kHid * L->val2_ld);
  L->fc2_sig_b = o; o = align4(o + L->adv2_ld + L->val2_ld);
  L->param_count = o;
  L->param_count_ref = 77984 + 2 * ((int64_t)kFlat * 512 * 2 + 1024) +
                       ((int64_t)kHid * NA * 2 + NA) + ((int64_t)kHid * K * 2 + K);
  // noise block of one apply
  L->n_adv1_in = 0; L->n_val1_in = kFlat; L->n_fc1_out = 2 * kFlat;
  L->n_adv2_in = 2 * kFlat + 1024; L->n_val2_in = L->n_adv2_in + kHid;
  L->n_fc2_out = L->n_val2_in + kHid;
  L->noise_stride = align4(L->n_fc2_out + L->adv2_ld + L->val2_ld);
  // workspace
  const int64_t GB = (int64_t)kG * B;
  const int64_t ld2 = L->adv2_ld + L->val2_ld;  // padded fc2 column space
  int64_t w = 0;
  auto take = [&](int64_t n) { int64_t r = w; w = align4(w + n); return r; };
  L->ws_act1 = take(GB * 400 * 32);
  L->ws_act2 = take(GB * 81 * 64);
  L->ws_feat = take(GB * kFlat);
  L->ws_fc1_part = take((int64_t)kMaxSplitFc1 * GB * 1024);
  L->ws_h1 = take(GB * 1024);
  L->ws_fc2_part = take((int64_t)kS_fc2 * GB * ld2);
  L->ws_fc2_out = take(GB * ld2);
  L->ws_dout2 = take((int64_t)B * ld2);
  L->ws_dh1 = take((int64_t)B * 1024);
  L->ws_dfeat_part = take((int64_t)kS_dfeat * B * kFlat);
  L->ws_dfeat = take((int64_t)B * kFlat);
  L->ws_dact2 = take((int64_t)B * 81 * 64);
  L->ws_dact1 = take((int64_t)B * 400 * 32);
  const int64_t wp = (int64_t)kS_cw1 * Conv1Wg::KROWS * 32 +
                     (int64_t)kS_cw2 * Conv2Wg::KROWS * 64 +
                     (int64_t)kS_cw3 * Conv3Wg::KROWS * 64;  // one slab per conv
  L->ws_wgrad_part = take(wp);
  L->ws_norm_part = take(kNormBlocks);
  L->ws_colsum_part = take(4);
  L->ws_scalars = take(16);
  L->ws_q_sel = take((int64_t)B * A);
  L->ws_target_probs = take((int64_t)B * K);
  L->ws_count = w;
  return DZ_OK;
}

// ---- the step ---------------------------------------------------------------
extern "C" int dz_rainbow_learn(const dz_rainbow_args_t* a, int phases,
                                dz_stream_t stream) {
  DZ_REQUIRE(a && a->online && a->target && a->ws && a->noise && a->support);
  DZ_REQUIRE(a->s_tm1 && a->s_t && a->a_tm1 && a->r_t && a->discount_t && a->weights);
  DZ_REQUIRE(a->losses && a->priorities);
  dz_rainbow_layout_t L;
  int rc = dz_rainbow_layout(a->num_actions, a->num_atoms, a->batch, &L);
  if (rc != DZ_OK) return rc;
  hipStream_t s = dz_s(stream);
  const int B = a->batch, A = a->num_actions, K = a->num_atoms;
  const int NA = A * K, NAp = L.adv2_ld, ld2 = L.adv2_ld + L.val2_ld;
  float* ws = a->ws;
  const float* prm[kG] = {a->online, a->online, a->target};
  const float* nz[kG] = {a->noise, a->noise + L.noise_stride,
                         a->noise + 2 * L.noise_stride};

  FwdHeads H;
  make_heads(L, H);
  const FcHead* fc1h = H.fc1h;
  const FcHead* fc2h = H.fc2h;

  if (g_dz_prof_on) dz_prof_begin(s);
  if ((phases & DZ_PHASE_FORWARD) && a->resample_noise) {
    DZ_REQUIRE(a->adam_count);
    const long n = 3 * L.noise_stride;
    hipLaunchKernelGGL(noise_fill_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                       s, const_cast<float*>(a->noise), n, a->noise_seed,
                       (uint64_t)0x5eed, a->adam_count);
    DZ_LAUNCH_CHECK();
    DZ_PROF(s, "noise");
  }
  if (phases & DZ_PHASE_FORWARD) {
    {
      const uint8_t* in[kG] = {a->s_tm1, a->s_t, a->s_t};
      rc = rainbow_forward(L, H, kG, B, prm, nz, in, ws, s);
      if (rc) return rc;
    }
    hipLaunchKernelGGL(rainbow_head_loss_kernel, dim3(B), dim3(64), 0, s,
                       ws + L.ws_fc2_out, ld2, NAp, B, A, K, a->a_tm1, a->r_t, a->discount_t,
                       a->weights, a->support, ws + L.ws_dout2, a->losses, a->priorities,
                       ws + L.ws_q_sel, ws + L.ws_target_probs);
    DZ_LAUNCH_CHECK();
      DZ_PROF(s, "head_loss");
  }

  if (phases & DZ_PHASE_BACKWARD) {
    DZ_REQUIRE(a->grad);
    float* grad = a->grad;
    // Every layer's weight gradient and input gradient are independent, so each
    // pair is ONE launch (dz_mfma_gemm2/3: horizontal fusion); the conv partial
    // reductions and the bias column sums are one launch at the end.
    float* part1 = ws + L.ws_wgrad_part;
    float* part2 = part1 + (long)kS_cw1 * Conv1Wg::KROWS * 32;
    float* part3 = part2 + (long)kS_cw2 * Conv2Wg::KROWS * 64;
    {  // fc2: weight gradients (both heads) + input gradient of each head
      FcWgradParams w;
      w.x = ws + L.ws_h1; w.ldx = 1024; w.dy = ws + L.ws_dout2; w.ldy = ld2; w.M = B;
      w.NH = 2; w.noisy = 1; w.noise = nz[0]; w.head[0] = fc2h[0]; w.head[1] = fc2h[1];
      w.grad = grad;
      FcDgradParams d[2];
      for (int h = 0; h < 2; ++h) {
        d[h].dy = ws + L.ws_dout2; d[h].ldy = ld2; d[h].M = B; d[h].NH = 1;
        d[h].S = kS_dh1; d[h].noisy = 1; d[h].params = a->online; d[h].noise = nz[0];
        d[h].head[0] = fc2h[h]; d[h].head[1] = fc2h[h];
        d[h].part = ws + L.ws_dfeat_part; d[h].ldo = 1024; d[h].K = kHid;
        d[h].x_off = 512 * h;
      }
      const dim3 gd(kHid / FcDg::BN, (B + 31) / 32, kS_dh1);
      rc = dz_launch_gemm3<FcWg, FcDg, FcDg>(
          w, dim3((NA + FcWg::BN - 1) / FcWg::BN, kHid / FcWg::BM, 2), d[0], gd, d[1], gd, s);
      if (rc) return rc;
      DZ_PROF(s, "fc2_wgrad+dgrad");
      hipLaunchKernelGGL(reduce_parts_kernel, dim3((B * 1024 + 63) / 64), dim3(256), 0,
                         s, ws + L.ws_dfeat_part, kS_dh1, (long)B * 1024, ws + L.ws_h1,
                         ws + L.ws_dh1);
      DZ_LAUNCH_CHECK();
      DZ_PROF(s, "dh1_reduce");
    }
    {  // fc1: weight gradients + input gradient (adv1 + val1 paths) -> dfeat
      FcWgradParams w;
      w.x = ws + L.ws_feat; w.ldx = kFlat; w.dy = ws + L.ws_dh1; w.ldy = 1024; w.M = B;
      w.NH = 2; w.noisy = 1; w.noise = nz[0]; w.head[0] = fc1h[0]; w.head[1] = fc1h[1];
      w.grad = grad;
      FcDgradParams d;
      d.dy = ws + L.ws_dh1; d.ldy = 1024; d.M = B; d.NH = 2; d.S = kS_dfeat; d.noisy = 1;
      d.params = a->online; d.noise = nz[0]; d.head[0] = fc1h[0]; d.head[1] = fc1h[1];
      d.part = ws + L.ws_dfeat_part; d.ldo = kFlat; d.K = kFlat; d.x_off = 0;
      // (measured: fusing these two HBM-heavy contractions is slower, 48 us vs
      // 15 + 23 us back to back, so this pair stays two launches)
      rc = dz_launch_gemm<FcWg>(w, dim3(512 / FcWg::BN, kFlat / FcWg::BM, 2), s);
      if (rc) return rc;
      DZ_PROF(s, "fc1_wgrad");
      rc = dz_launch_gemm<FcDg>(d, dim3(kFlat / FcDg::BN, (B + 31) / 32, kS_dfeat), s);
      if (rc) return rc;
      DZ_PROF(s, "fc1_dgrad");
      hipLaunchKernelGGL(reduce_parts_kernel, dim3((B * kFlat + 63) / 64), dim3(256), 0,
                         s, ws + L.ws_dfeat_part, kS_dfeat, (long)B * kFlat,
                         ws + L.ws_feat, ws + L.ws_dfeat);
      DZ_LAUNCH_CHECK();
      DZ_PROF(s, "dfeat_reduce");
    }
    {  // conv3: weight+bias gradient partials + input gradient (relu(conv2) mask)
      ConvWgradParams w;
      w.in = ws + L.ws_act2; w.dy = ws + L.ws_dfeat; w.part = part3; w.B = B; w.S = kS_cw3;
      ConvDgradParams d;
      d.dy = ws + L.ws_dfeat; d.w = a->online + L.conv_w[2]; d.act = ws + L.ws_act2;
      d.dx = ws + L.ws_dact2; d.B = B;
      rc = dz_launch_gemm2<Conv3Wg, Conv3Dg>(
          w, dim3(64 / Conv3Wg::BN, Conv3Wg::MT, kS_cw3), d,
          dim3(64 / Conv3Dg::BN, Conv3Dg::tiles(B), 1), s);
      if (rc) return rc;
      DZ_PROF(s, "conv3_wgrad+dgrad");
    }
    {  // conv2
      ConvWgradParams w;
      w.in = ws + L.ws_act1; w.dy = ws + L.ws_dact2; w.part = part2; w.B = B; w.S = kS_cw2;
      ConvDgradParams d;
      d.dy = ws + L.ws_dact2; d.w = a->online + L.conv_w[1]; d.act = ws + L.ws_act1;
      d.dx = ws + L.ws_dact1; d.B = B;
      rc = dz_launch_gemm2<Conv2Wg, Conv2Dg>(
          w, dim3(64 / Conv2Wg::BN, Conv2Wg::MT, kS_cw2), d,
          dim3(32 / Conv2Dg::BN, Conv2Dg::tiles(B), 4), s);
      if (rc) return rc;
      DZ_PROF(s, "conv2_wgrad+dgrad");
    }
    {  // conv1 weight+bias gradient partials straight from the uint8 states
      ConvWgradParams p;
      p.in = a->s_tm1; p.dy = ws + L.ws_dact1; p.part = part1; p.B = B; p.S = kS_cw1;
      rc = dz_launch_gemm<Conv1Wg>(p, dim3(32 / Conv1Wg::BN, Conv1Wg::MT, kS_cw1), s);
      if (rc) return rc;
      DZ_PROF(s, "conv1_wgrad");
    }
    {  // reduce the three conv partial slabs; linear-layer bias gradients
      FinalizeJobs J;
      J.r[0] = {part1, kS_cw1, (long)Conv1Wg::KROWS * 32, grad + L.conv_w[0]};
      J.r[1] = {part2, kS_cw2, (long)Conv2Wg::KROWS * 64, grad + L.conv_w[1]};
      J.r[2] = {part3, kS_cw3, (long)Conv3Wg::KROWS * 64, grad + L.conv_w[2]};
      unsigned acc = 0;
      for (int j = 0; j < 3; ++j) { acc += (unsigned)((J.r[j].n + 63) / 64); J.r_end[j] = acc; }
      J.c[0] = {ws + L.ws_dh1, B, 1024, 1024, grad + L.fc1_mu_b, nz[0] + L.n_fc1_out,
                grad + L.fc1_sig_b};
      J.c[1] = {ws + L.ws_dout2, B, ld2, ld2, nullptr, nz[0] + L.n_fc2_out,
                grad + L.fc2_sig_b};
      J.c_tiles[0] = 16; J.c_tiles[1] = (unsigned)((ld2 + 63) / 64);
      hipLaunchKernelGGL(finalize_grads_kernel, dim3(acc + J.c_tiles[0] + J.c_tiles[1]),
                         dim3(256), 0, s, J);
      DZ_LAUNCH_CHECK();
      DZ_PROF(s, "finalize_grads");
    }
  }

  if (phases & DZ_PHASE_OPTIMIZER) {
    DZ_REQUIRE(a->grad && a->adam_m && a->adam_v && a->adam_count);
    float* sc = ws + L.ws_scalars;
    hipLaunchKernelGGL(sumsq_kernel, dim3(kNormBlocks), dim3(256), 0, s, a->grad,
                       (long)L.param_count, ws + L.ws_norm_part);
    DZ_LAUNCH_CHECK();
    DZ_PROF(s, "grad_sumsq");
    hipLaunchKernelGGL(opt_scalars_kernel, dim3(1), dim3(256), 0, s,
                       ws + L.ws_norm_part, kNormBlocks, a->adam_count, a->b1, a->b2,
                       a->max_norm, a->losses, a->weights, B, sc);
    DZ_LAUNCH_CHECK();
    DZ_PROF(s, "opt_scalars");
    hipLaunchKernelGGL(adam_kernel, dim3(2048), dim3(256), 0, s, a->online, a->grad,
                       a->adam_m, a->adam_v, (long)(L.param_count >> 2), sc, a->lr,
                       a->b1, a->b2, a->eps, a->max_norm);
    DZ_LAUNCH_CHECK();
      DZ_PROF(s, "adam");
  }
  return DZ_OK;
}

extern "C" int dz_rainbow_graph_capture(const dz_rainbow_args_t* args, int phases,
                                        dz_stream_t stream, void** graph_exec_out) {
  DZ_REQUIRE(args && graph_exec_out && stream && !g_dz_prof_on);
  int rc = ensure_aux();  // no resource creation inside the capture
  if (rc) return rc;
  hipStream_t s = dz_s(stream);
  hipGraph_t graph = nullptr;
  DZ_HIP_CHECK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
  rc = dz_rainbow_learn(args, phases, stream);
  hipError_t e = hipStreamEndCapture(s, &graph);
  if (rc) { if (graph) (void)hipGraphDestroy(graph); return rc; }
  DZ_HIP_CHECK(e);
  hipGraphExec_t exec = nullptr;
  e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
  (void)hipGraphDestroy(graph);
  DZ_HIP_CHECK(e);
  *graph_exec_out = (void*)exec;
  return DZ_OK;
}

extern "C" int dz_graph_launch(void* graph_exec, dz_stream_t stream) {
  DZ_REQUIRE(graph_exec);
  DZ_HIP_CHECK(hipGraphLaunch((hipGraphExec_t)graph_exec, dz_s(stream)));
  return DZ_OK;
}

extern "C" int dz_graph_destroy(void* graph_exec) {
  DZ_REQUIRE(graph_exec);
  DZ_HIP_CHECK(hipGraphExecDestroy((hipGraphExec_t)graph_exec));
  return DZ_OK;
}

extern "C" int dz_rainbow_apply(int num_actions, int num_atoms, int batch,
                                const float* params, const uint8_t* states,
                                const float* noise, const float* support, float* ws,
                                float* q_values_out, int32_t* greedy_out,
                                float* vmax_out, dz_stream_t stream) {
  DZ_REQUIRE(params && states && noise && support && ws && q_values_out);
  dz_rainbow_layout_t L;
  int rc = dz_rainbow_layout(num_actions, num_atoms, batch, &L);
  if (rc != DZ_OK) return rc;
  hipStream_t s = dz_s(stream);
  FwdHeads H;
  make_heads(L, H);
  const float* prm[kG] = {params, params, params};
  const float* nz[kG] = {noise, noise, noise};
  const uint8_t* in[kG] = {states, states, states};
  const bool prof = g_dz_prof_on;
  g_dz_prof_on = false;  // marks belong to dz_rainbow_learn
  rc = rainbow_forward(L, H, 1, batch, prm, nz, in, ws, s);
  g_dz_prof_on = prof;
  if (rc) return rc;
  hipLaunchKernelGGL(rainbow_q_values_kernel, dim3(batch), dim3(64), 0, s,
                     ws + L.ws_fc2_out, L.adv2_ld + L.val2_ld, L.adv2_ld, num_actions,
                     num_atoms, support, q_values_out, greedy_out, vmax_out);
  DZ_LAUNCH_CHECK();
  return DZ_OK;
}

extern "C" int dz_set_tuning(int key, int value) {
  switch (key) {
    case 0: g_fc1_variant = value; return DZ_OK;
    case 1: DZ_REQUIRE(value >= 1 && value <= kMaxSplitFc1); g_fc1_splits = value; return DZ_OK;
    case 2: g_fc1_dgrad_stream = value; return DZ_OK;
    case 3: g_fc1_blocked_experiment = value; return DZ_OK;
    case 4: g_overlap = value; return DZ_OK;
    default: return DZ_ERR_INVALID_ARG;
  }
}

extern "C" int dz_noise_fill(float* noise, int64_t count, uint64_t seed,
                             uint64_t counter, dz_stream_t stream) {
  DZ_REQUIRE(noise && count > 0);
  hipLaunchKernelGGL(noise_fill_kernel, dim3((unsigned)((count + 255) / 256)),
                     dim3(256), 0, dz_s(stream), noise, (long)count, seed, counter,
                     (const int32_t*)nullptr);
  DZ_LAUNCH_CHECK();
  return DZ_OK;
}

extern "C" int dz_param_copy(float* dst, const float* src, int64_t count,
                             dz_stream_t stream) {
  DZ_REQUIRE(dst && src && count > 0 && (count & 3) == 0);
  hipLaunchKernelGGL(copy_kernel, dim3(1024), dim3(256), 0, dz_s(stream), dst, src,
                     (long)(count >> 2));
  DZ_LAUNCH_CHECK();
  return DZ_OK;
}
