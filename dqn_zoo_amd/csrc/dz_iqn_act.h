// The IQN actor's decision for ONE observation as ONE launch (round 5; the Rainbow and dense-head
// forms are dz_act_one.h, whose torso role and seam words this kernel shares).
// (ref: iqn/agent.py:234-247 select_action: N = tau_samples_policy fresh draws tau_j ~ U[0, 1),
//  q = mean_j network(s, tau_j), epsilon-greedy on q; networks.py:264-292 iqn_atari_network:
//  cos(pi i tau) -> linear(3136) -> ReLU, times the torso features, -> linear(512) -> ReLU ->
//  linear(A).)
//
// At one observation the value head still sees N rows (N = tau_samples_policy = 64 by default,
// iqn/run_atari.py:98; up to 64 here, two 32-row MFMA blocks per fc1 workgroup): the multi-launch apply
// is a tau draw, three convolutions, the cosine table, three GEMM launches, a copy and the
// q-value kernel -- ten launches, ~70 us, almost all of it launch floors.  Here:
//
//   torso  (25 workgroups)  act_torso_block (dz_act_one.h): conv1 -> conv2 -> conv3, features as a seam
//   fc1    (112)            28 K-splits x 4 column groups of 128.  A workgroup requests its 112 x
//                           128 slice of W1 lane-wise in MFMA B-operand layout (56 registers) at
//                           launch, draws the N taus itself (a pure function of the stream
//                           position: the draws of dz_uniform_fill), builds the cosine table and
//                           its 112 columns of the tau embedding relu(cos W_emb + b_emb) in LDS
//                           while the torso runs, then multiplies by the feature slice (seam) and
//                           chains 56 v_mfma_f32_32x32x2_f32 per wave (rows = taus); the [N][128]
//                           partial tile goes out transposed through LDS as 16-byte seam stores
//   tail   (N)              one workgroup per tau: folds the 28 slabs of its row (+ b1, ReLU), the
//                           second layer with W2 in registers from the start, a ticket; the last
//                           workgroup averages over the taus and stores every q-value as one
//                           8-byte {value, marker} word into the pinned slot the host polls
//
// Liveness, bounded spins and the failure protocol: dz_act_one.h (DZ_ACT_FAILED_MARKER).
#pragma once

#include "dz_act_one.h"

namespace {

constexpr int kIqnActMaxTaus = 64, kIqnActMaxLatent = 64;
constexpr int kIqnActPartLd = kIqnActMaxTaus * kHid;                     // one K-split's slab: [64][512]
constexpr int kIqnActSeamWords = act_seam_words(kIqnActPartLd);
constexpr int kIqnActFc1Blocks = kActFc1Splits * 4;

struct IqnActParams : ActTorso {
  int latent, N, A, ld2;                 // N <= 64 taus, A <= 32 actions
  long emb_w, emb_b, fc1_b, fc2_w, fc2_b;   // (fc1_mu_w / fc1_ld of ActTorso: the [3136][512] matrix)
  uint64_t tau_seed, tau_counter;        // tau_j = the draw dz_uniform_fill makes at counter + j
  float* taus_out;                       // [N] (nullable): the draws, written out for tests
  float* out;                            // [N][ld2] per-tau head outputs (ticket-ordered)
  unsigned long long* pairs_out;         // [A] {float q, float marker}
};

__device__ __forceinline__ float iqn_tau_at(uint64_t seed, uint64_t pos) {   // uniform_fill_kernel's draw
  const uint64_t h = mix64(mix64(seed) ^ mix64(pos));
  return (float)(h >> 41) * (1.0f / 8388608.0f);
}

// LDS of the fc1 role (NT = 32 NB taus): cos [NT][65] | emb / hin [112][NT + 1] -> later the
// [NT][132] output tile
constexpr int kIqnCosLd = kIqnActMaxLatent + 1, kIqnTileLd = 132;
constexpr int iqn_act_lds_floats(int nt) {
  return (nt * kIqnCosLd + kActFc1Rows * (nt + 1)) > nt * kIqnTileLd ? (nt * kIqnCosLd + kActFc1Rows * (nt + 1))
                                                                     : nt * kIqnTileLd;
}
constexpr int kIqnActLdsFloats = iqn_act_lds_floats(kIqnActMaxTaus) > kActLdsFloats ? iqn_act_lds_floats(kIqnActMaxTaus)
                                                                                     : kActLdsFloats;
static_assert(iqn_act_lds_floats(32) <= kActLdsFloats, "up to 32 taus the torso role's LDS block covers the fc1 role");

// NB = 32-row blocks of taus (1: N <= 32, 2: N <= 64)
template <int NB>
__device__ __forceinline__ void iqn_act_fc1_block(const IqnActParams& p, int fb, float* lds) {
  constexpr int NT = 32 * NB, kIqnHinLd = NT + 1;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, l31 = lane & 31;
  const int split = fb >> 2, cg = fb & 3;
  const int k0 = kActFc1Rows * split, n0 = 128 * cg + 32 * wave;
  float* s_cos = lds;                               // [NT][65]
  float* s_hin = lds + NT * kIqnCosLd;              // [112][NT + 1]: emb, then emb * feat
  ACT_STAMP(0);
  const unsigned gen = *act_line(p.sync, 3);
  float* const set = act_set(p, gen);
  const float* const feat = set + kActOffFeat;
  {  // the NEXT decision's set (last read one decision ago) goes back to all-zero bits
    const __amdgpu_buffer_rsrc_t nr = act_rsrc(act_set(p, gen + 1u));
    for (int c = fb * 1024 + 4 * tid; c < p.set_floats; c += kIqnActFc1Blocks * 1024)
      act_store4(nr, (unsigned)c * 4u, make_float4(0.f, 0.f, 0.f, 0.f));
  }
  // (1) this wave's 32 columns of W1's K slice, lane-wise in B-operand layout:
  //     wb[u] = W1[k0 + 2 u + half][n0 + l31]
  float wb[kActFc1Rows / 2];
  {
    const float* w1 = p.prm + p.fc1_mu_w + (long)(k0 + half) * p.fc1_ld + n0 + l31;
#pragma unroll
    for (int u = 0; u < kActFc1Rows / 2; ++u) wb[u] = w1[(long)(2 * u) * p.fc1_ld];
  }
  __builtin_amdgcn_sched_barrier(0);
  // (2) the taus and their cosine table cos(pi (i + 1) tau_j)   (iqn_cos_kernel's arithmetic)
  if (tid < NT) {
    const float tau = iqn_tau_at(p.tau_seed, p.tau_counter + (uint64_t)min(tid, p.N - 1));
    if (fb == 0 && tid < p.N && p.taus_out) p.taus_out[tid] = tau;
    lds[NT * kIqnCosLd + tid] = tau;   // (parked in the hin block until the table is built)
  }
  __syncthreads();
  for (int e = tid; e < NT * p.latent; e += 256) {
    const int j = e / p.latent, i = e - j * p.latent;
    const float tau = lds[NT * kIqnCosLd + j];
    s_cos[j * kIqnCosLd + i] = cosf((float)(i + 1) * 3.14159274101257324f * tau);
  }
  __syncthreads();
  // (3) the tau embedding of this K slice: emb[j][k] = relu(sum_i cos[j][i] W_emb[i][k] + b_emb[k]),
  //     i ascending; thread = (k, half of the taus)
  const int kk = tid & 127, jg = tid >> 7;
  const bool kon = kk < kActFc1Rows;
  const int kc = min(kk, kActFc1Rows - 1);
  constexpr int TPT = 16 * NB;   // taus per thread
  float acc[TPT];
#pragma unroll
  for (int jj = 0; jj < TPT; ++jj) acc[jj] = 0.f;
  {
    const float* we = p.prm + p.emb_w + k0 + kc;
    for (int i0 = 0; i0 < p.latent; i0 += 8) {
      float wv[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) wv[q] = we[(long)(i0 + q) * kFlat];
#pragma unroll
      for (int q = 0; q < 8; ++q)
#pragma unroll
        for (int jj = 0; jj < TPT; ++jj)
          acc[jj] = __builtin_fmaf(s_cos[(TPT * jg + jj) * kIqnCosLd + i0 + q], wv[q], acc[jj]);
    }
  }
  const float be = p.prm[p.emb_b + k0 + kc];
  ACT_STAMP(1);
  // (4) the feature slice: conv3's outputs are their own flags
  float f = 0.f;
  {
    int round = 0;
    bool miss, give_up;
    act_watch(feat + k0 + kActFc1Rows - 1, p.spin_limit);
    do {
      f = act_load(feat + k0 + kc);
      miss = act_missing(f);
    } while (act_again(miss, round++, act_line(p.sync, 5), &give_up, p.spin_limit));
    if (give_up) return;
  }
  ACT_STAMP(2);
  if (kon) {
#pragma unroll
    for (int jj = 0; jj < TPT; ++jj) {
      const int j = TPT * jg + jj;
      const float e = acc[jj] + be;
      s_hin[kk * kIqnHinLd + j] = j < p.N ? (e > 0.f ? e : 0.f) * f : 0.f;   // networks.py:281-285
    }
  }
  __syncthreads();
  // (5) partial[j][n] = sum_k hin[j][k] W1[k][n] over the K slice: rows = taus
  f32x16 pacc[NB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb)
#pragma unroll
    for (int i = 0; i < 16; ++i) pacc[nb][i] = 0.f;
#pragma unroll
  for (int u = 0; u < kActFc1Rows / 2; ++u)
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
      pacc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(s_hin[(2 * u + half) * kIqnHinLd + 32 * nb + l31], wb[u],
                                                      pacc[nb], 0, 0, 0);
  __syncthreads();   // (hin is dead: the output tile takes its place)
  ACT_STAMP(3);
  float* T = lds;    // [NT][132]
#pragma unroll
  for (int nb = 0; nb < NB; ++nb)
#pragma unroll
    for (int i = 0; i < 16; ++i) T[(32 * nb + dz_acc_row(i, lane)) * kIqnTileLd + 32 * wave + l31] = pacc[nb][i];
  __syncthreads();
  const __amdgpu_buffer_rsrc_t sr = act_rsrc(set + kActOffPart);
#pragma unroll
  for (int q = 0; q < 4 * NB; ++q) {
    const int idx = tid + 256 * q, j = idx >> 5, c4 = idx & 31;
    if (j < p.N)
      act_store4(sr, (unsigned)(split * p.part_ld + j * kHid + 128 * cg + 4 * c4) * 4u,
                 act_mark4(*(const float4*)(T + j * kIqnTileLd + 4 * c4)));
  }
  ACT_STAMP(4);
}

// One workgroup per tau j: h1[j] = relu(slab sum + b1), out[j] = h1[j] W2 + b2; the last one
// (ticket) stores q = mean_j out[j] (iqn_q_values_kernel's order: one wave, lane j holds row j).
__device__ __forceinline__ void iqn_act_tail_block(const IqnActParams& p, int j, float* lds) {
  float* s_h = lds;            // [512]
  float* s_red = lds + 512;    // [8][32]
  int* s_last = (int*)(lds + 768);
  const int tid = threadIdx.x;
  ACT_STAMP(0);
  const int n = tid & 31, ks = tid >> 5, nc = min(n, p.A - 1);
  float w[64];
  {
    const float* w2 = p.prm + p.fc2_w + (long)(ks * 64) * p.ld2 + nc;
#pragma unroll
    for (int k = 0; k < 64; ++k) {   // (running pointer kept opaque, see act_tail_block)
      w[k] = *w2;
      w2 += p.ld2;
      asm volatile("" : "+v"(w2));
    }
  }
  const float b1a = p.prm[p.fc1_b + tid], b1b = p.prm[p.fc1_b + 256 + tid];
  const float b2 = p.prm[p.fc2_b + nc];
  __builtin_amdgcn_sched_barrier(0);
  const unsigned gen = *act_line(p.sync, 3);
  const unsigned sticky = *act_line(p.sync, 5);
  ACT_STAMP(1);
  float x[2][kActFc1Splits];
  bool failed;
  {
    const float* part = act_set(p, gen) + kActOffPart + j * kHid + tid;
    int round = 0;
    bool miss, give_up;
    act_watch(part + (long)(kActFc1Splits - 1) * p.part_ld, p.spin_limit);
    do {
      const float* pp = part;
#pragma unroll
      for (int s = 0; s < kActFc1Splits; ++s) {
        asm volatile("" : "+v"(pp));
        x[0][s] = act_load(pp); x[1][s] = act_load(pp + 256);
        pp += p.part_ld;
      }
      unsigned all = 1u;
#pragma unroll
      for (int e = 0; e < 2; ++e)
#pragma unroll
        for (int s = 0; s < kActFc1Splits; ++s) all &= __builtin_bit_cast(unsigned, x[e][s]) != 0u ? 1u : 0u;
      miss = all == 0u;
    } while (act_again(miss, round++, act_line(p.sync, 5), &give_up, p.spin_limit));
    failed = give_up || sticky != 0u;
  }
  ACT_STAMP(2);
  if (!failed) {
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      float v = 0.f;
#pragma unroll
      for (int s = 0; s < kActFc1Splits; ++s) v += x[e][s];
      const float h = v + (e ? b1b : b1a);
      s_h[tid + 256 * e] = h > 0.f ? h : 0.f;
    }
    __syncthreads();
    {
      float acc = 0.f;
#pragma unroll
      for (int k = 0; k < 64; ++k) acc = __builtin_fmaf(s_h[ks * 64 + k], w[k], acc);
      s_red[ks * 32 + n] = acc;
    }
    __syncthreads();
    if (tid < 32 && n < p.A) {
      const float o = (((s_red[n] + s_red[32 + n]) + (s_red[64 + n] + s_red[96 + n])) +
                       ((s_red[128 + n] + s_red[160 + n]) + (s_red[192 + n] + s_red[224 + n]))) + b2;
      act_store(p.out + j * p.ld2 + n, o);
    }
  } else if (tid == 0) {
    __hip_atomic_store(act_line(p.sync, 5), 1u, DZ_ACT_RLX);
  }
  ACT_STAMP(3);
  // ticket: the last tau's workgroup finishes the decision
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid == 0) {
    int last = __hip_atomic_fetch_add(act_line(p.sync, 4), 1u, DZ_ACT_RLX) == (unsigned)p.N - 1;
    if (last && __hip_atomic_load(act_line(p.sync, 5), DZ_ACT_RLX) != 0u) last = 2;
    *s_last = last;
  }
  __syncthreads();
  const int last = *s_last;
  if (!last) return;
  if (tid == 0) {   // re-armed for the next decision (every poller has passed)
    __hip_atomic_store(act_line(p.sync, 4), 0u, DZ_ACT_RLX);
    __hip_atomic_store(act_line(p.sync, 3), gen + 1u, DZ_ACT_RLX);
  }
  if (tid >= 64) return;
  for (int a = 0; a < p.A; ++a) {
    float q;
    if (last == 2) {
      q = __builtin_nanf("");
    } else {
      const float y = act_load(p.out + min(tid, p.N - 1) * p.ld2 + a);
      q = wave_sum(tid < p.N ? y : 0.f) / (float)p.N;
    }
    if (tid == 0)
      p.pairs_out[a] = (unsigned long long)__builtin_bit_cast(unsigned, q) |
                       ((unsigned long long)__builtin_bit_cast(unsigned, last == 2 ? kActFailedMarker : 1.0f) << 32);
  }
  ACT_STAMP(4);
}

__global__ __launch_bounds__(256, 2) void iqn_act_one_kernel(IqnActParams p) {
  __shared__ __attribute__((aligned(16))) float lds[kIqnActLdsFloats];
  const int b = blockIdx.x;
  if (b < kActTorsoBlocks) act_torso_block(p, b, lds);
  else if (b < kActTorsoBlocks + kIqnActFc1Blocks) {
    if (p.N <= 32) iqn_act_fc1_block<1>(p, b - kActTorsoBlocks, lds);   // (launch-uniform)
    else iqn_act_fc1_block<2>(p, b - kActTorsoBlocks, lds);
  } else iqn_act_tail_block(p, b - kActTorsoBlocks - kIqnActFc1Blocks, lds);
}

}  // namespace
