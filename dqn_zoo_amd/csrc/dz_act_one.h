// The actor's decision for ONE observation as ONE launch: Rainbow (rainbow_act_one_kernel) and the
// dense-head agents (dense_act_one_kernel: DQN, double-Q, prioritized, C51, QR-DQN).
// (ref: rainbow/agent.py:125-133,171-179 select_action -> network.apply on a batch of one;
//  networks.py:186-204 torso, :137-178 noisy linear, :229-261 dueling C51 head, :206-221 / :295-363
//  the dense heads; dqn/agent.py:121-131.)
//
// At one image the network is 15 MFLOP of convolution, a 25.7 MB weight stream (Rainbow's fc1 mu
// and sigma) and 1.5 MB for the second layer -- five launches of 7-9 us each spent 41 us on it,
// almost all of it launch floors and cold round trips in series.  Here the workgroups of one launch
// take three roles and hand results to each other through write-through (sc1) stores and sc1 loads
// (no L2 write-back or invalidate anywhere), and THE DATA IS ITS OWN FLAG: every intermediate lives
// in a buffer that is all-zero bits before its producers write it, producers store -0.0f for a
// zero, and a consumer re-reads the values it needs until none of them is +0.0f -- first ONE thread
// watching ONE word, then whole-workgroup rounds.  No arrival counter, no drained `vmcnt`, no
// fence, no second round trip for the payload (the first version of this kernel did counter + flag
// + load: 23.0 us in-kernel against 21.5; EXPERIMENTS.md).  Two sets of buffers alternate by the
// parity of a generation word; the set of the NEXT apply is cleared by fc1 workgroups of this one.
//
//   torso  (25 workgroups)  conv1 -> conv2 -> conv3.  A workgroup owns 16 output pixels (a 4 x 4
//                           square x 32 channels in conv1: 1.6 KB of the observation over PCIe per
//                           workgroup; 16 consecutive pixels x 16 channels in conv2 / conv3), its 4
//                           waves split K; the input patch is copied once into LDS, A fragments
//                           are read from the patch (step order chosen so that one 4- or 16-byte
//                           LDS read feeds four MFMAs), B fragments (weights) were requested
//                           lane-wise straight into MFMA operand registers for ALL three layers
//                           before the first wait; v_mfma_f32_16x16x4_f32.
//   fc1    (224 / 112)      28 K-splits x 8 (Rainbow) or 4 (dense) column groups of 128: every
//                           thread requests its 14 x 4 weights (mu and sigma) at launch, draws the
//                           eps it needs itself (dz_noise_at is a pure function of the stream
//                           position), forms W_eff in registers and only then waits for the
//                           torso's feature vector: the weight stream runs UNDER the convolutions.
//   tail   (12 / 1..114)    the second layer's 32-column tiles (weights in registers from the
//                           start), fold the 28 fc1 slabs.  Rainbow: a ticket, the last workgroup
//                           emits q-values, greedy action and value -- the pair as ONE 8-byte
//                           store into the pinned slot the host polls.  Dense: every head output
//                           as one 8-byte {value, marker} store.
//
// LIVENESS.  Dependencies only point from lower to higher block ids.  The launch relies on ONE
// property of the dispatcher that HIP does not promise but every CDNA part has (and
// MI355X_MICROARCH.md "Workgroup dispatch" describes): workgroups are handed to the XCDs round-robin
// by linear id and every XCD starts its share in ascending id order.  Then the lowest-id unfinished
// workgroup is either resident or the next one its XCD starts, whatever else occupies the chip (a
// learner step on another stream delays the decision, it cannot block it): no co-residency of the
// roles is needed for progress, only for speed.  If that property ever failed, a consumer would
// spin on a producer that cannot start -- so EVERY spin is bounded (spin_limit rounds): a stuck
// seam sets the sticky word (line 5), the workgroups still take their tickets (the counters
// re-arm), and the decision comes back as action kActFailedAction with a NaN value / outputs
// marked kActFailedMarker -- which the host-side readers turn into an exception after clearing
// the seam area (learner.py) -- instead of hanging the device or returning a wrong action.
#pragma once

#include "dz_qnet_kernels.h"
#include "dz_seam.h"
#include "dz_sumtree_dev.h"

namespace {

typedef float act_f4 __attribute__((ext_vector_type(4)));

constexpr int kActTorsoBlocks = 25;   // conv1 pixel tiles (400 / 16)
constexpr int kActConv2Blocks = 24;   // 6 pixel tiles x 4 channel tiles
constexpr int kActConv3Blocks = 16;   // 4 pixel tiles x 4 channel tiles
constexpr int kActFc1Splits = 28, kActFc1Rows = 112, kActFc1Blocks = kActFc1Splits * 8;
constexpr int kActFailedAction = -2;  // DZ_ACT_FAILED: the "action" of a decision whose seams timed out
constexpr float kActFailedMarker = 2.0f;   // dense heads: the marker of an output of such a decision
// one set of intermediates: act1 | act2 | feat | 28 fc1 slabs, padded to whole 256-float chunks
constexpr int kActOffA1 = 0, kActOffA2 = 400 * 32, kActOffFeat = kActOffA2 + 81 * 64,
              kActOffPart = kActOffFeat + kFlat;
constexpr int act_set_floats(int part_ld) { return (kActOffPart + kActFc1Splits * part_ld + 255) & ~255; }
constexpr int act_seam_words(int part_ld) { return 64 * 8 + 2 * act_set_floats(part_ld); }   // 8 lines + two sets
constexpr int kActSeamWords = act_seam_words(1024), kDenseActSeamWords = act_seam_words(512);
constexpr int kActLdsFloats = 8 * 20 * 36 + 4 * 256;   // largest patch (conv2) + partial tiles
static_assert(kActFc1Splits * kActFc1Rows == kFlat, "fc1 K-splits");

// (tools builds, -DDZ_ACT_STAMPS) per-workgroup wall-clock stamps in the idle dfeat slab buffer
#ifdef DZ_ACT_STAMPS
#define ACT_STAMP(i) do { if (threadIdx.x == 0) p.dbg[blockIdx.x * 16 + (i)] = (long long)wall_clock64(); } while (0)
#else
#define ACT_STAMP(i) do {} while (0)
#endif

// What every role needs: the torso's inputs and the seam area.
struct ActTorso {
  const uint8_t* obs;                 // [84][84][4], device or pinned device-mapped host memory
  const float* prm;
  long conv_w[3], conv_b[3];
  // Seam area (dz_*_layout_t::ws_act_seams; zero in a fresh workspace, owned by these kernels):
  // line 3 generation, line 4 tail tickets, line 5 sticky failure (lines are 256 bytes apart), then
  // the two sets of intermediates: act1 | act2 | feat | 28 fc1 slabs of part_ld columns.
  unsigned* sync;
  int set_floats;                     // floats per set (a multiple of 256)
  int ncg, part_ld;                   // fc1: column groups of 128, slab row length (128 ncg)
  int spin_limit;                     // polling rounds per seam before giving up (g_dz_act_spin_limit;
                                      // dz_act_debug_spin_limit lowers it to force the failure path)
  long fc1_mu_w; int fc1_ld;
  long long* dbg = nullptr;
};

struct ActOneParams : ActTorso {      // Rainbow: noisy fc1 (adv | val), dueling C51 head
  static constexpr bool NOISY = true;
  long fc1_sig_w, fc1_mu_b, fc1_sig_b;
  float* noise; int n_noise;          // the apply's noise block, also written out (tests, state)
  uint64_t seed, counter; const int32_t* step;
  int n_eps_in[2]; int n_fc1_out;
  FcHead head[2];                     // adv2, val2
  long fc2_sig_b; int n_fc2_out;
  int ld2, val_off, A, K;
  const float* support;
  float* fc2_out; float* q_out; int32_t* greedy_out; float* vmax_out;
  int tiles0, tiles;
  int32_t* bump;
};

struct DenseActParams : ActTorso {    // dense heads: linear(512) + ReLU + linear(N)
  static constexpr bool NOISY = false;
  long fc1_b, fc2_w, fc2_b; int ld2, N, bias_shared;
  int tiles;                          // tail workgroups: 32 output columns each
  unsigned long long* pairs_out;      // [N] {float output, float 1.0f}: one 8-byte store each
};

// ---- seams (primitives: dz_seam.h) ------------------------------------------------------------
__device__ __forceinline__ unsigned* act_line(unsigned* sync, int i) { return sync + 64 * i; }
__device__ __forceinline__ float* act_set(const ActTorso& p, unsigned gen) {
  return (float*)(p.sync + 64 * 8) + (gen & 1u) * p.set_floats;
}
// ---- torso -----------------------------------------------------------------------------------
// 16x16x4 operand maps: A lane l = A[l & 15][l >> 4], B lane l = B[l >> 4][l & 15],
// D lane l reg r = D[4 (l >> 4) + r][l & 15].  The k's of one MFMA step are ANY four k's as long
// as both operands agree, so steps are chosen such that one 4-byte (conv1) or 16-byte (conv2/3)
// LDS read per lane feeds four steps.
__device__ __forceinline__ void act_partial_to_lds(float* red, int wave, int lane, const act_f4& a,
                                                   const act_f4& b) {
  float* d = red + wave * 256 + (4 * (lane >> 4)) * 16 + (lane & 15);
  d[0] = a[0] + b[0]; d[16] = a[1] + b[1]; d[32] = a[2] + b[2]; d[48] = a[3] + b[3];
}

__device__ __forceinline__ void act_torso_block(const ActTorso& p, int blk, float* lds) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n = lane & 15, kq = lane >> 4;
  float* red = lds + 8 * 20 * 36;
  const float* W1 = p.prm + p.conv_w[0];
  const float* W2 = p.prm + p.conv_w[1];
  const float* W3 = p.prm + p.conv_w[2];
  // conv1: wave = (channel tile, K half); step (tl, c): tap 4 (8 kh1 + tl) + kq, channel c
  const int ct1 = wave & 1, kh1 = wave >> 1;
  const int pt2 = min(blk >> 2, 5), ct2 = blk & 3;   // conv2: wave = kernel row
  const int pt3 = min(blk >> 2, 3), ct3 = blk & 3;   // conv3: wave = 9 of the 36 (tap, 16-channel) pairs

  ACT_STAMP(0);
  const unsigned gen = *act_line(p.sync, 3);
  float* const set = act_set(p, gen);
  float* const act1 = set + kActOffA1; float* const act2 = set + kActOffA2;
  float* const feat = set + kActOffFeat;
  unsigned* const fail = act_line(p.sync, 5);
  bool give_up;
  // conv1's patch first (vector loads return in order).  conv1's tile is a 4 x 4 SQUARE of output
  // pixels (5 x 5 tiles): its input patch is 20 x 20 pixels x 4 bytes = 1.6 KB, 40 KB over PCIe for
  // the whole image instead of the 100 KB that 16 consecutive pixels (12 full rows) cost
  const int ty1 = blk / 5, tx1 = blk % 5;
  unsigned raw[2];
  {
    const unsigned* src = (const unsigned*)p.obs;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int idx = min(tid + 256 * i, 399);
      raw[i] = src[(16 * ty1 + idx / 20) * 84 + 16 * tx1 + idx % 20];
    }
  }
  __builtin_amdgcn_sched_barrier(0);
  // ---- every weight this workgroup will ever need, requested now -------------------------------
  float b1[32], b2[32], b3[36];
#pragma unroll
  for (int tl = 0; tl < 8; ++tl)
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int tap = 4 * (8 * kh1 + tl) + kq;
      b1[4 * tl + c] = W1[(tap * 4 + c) * 32 + 16 * ct1 + n];
    }
#pragma unroll
  for (int kw = 0; kw < 4; ++kw)
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int k = (4 * wave + kw) * 32 + 16 * g + 4 * kq + j;
        b2[(kw * 2 + g) * 4 + j] = W2[k * 64 + 16 * ct2 + n];
      }
#pragma unroll
  for (int i = 0; i < 9; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int q = 9 * wave + i;
      const int k = (q >> 2) * 64 + 16 * (q & 3) + 4 * kq + j;
      b3[4 * i + j] = W3[k * 64 + 16 * ct3 + n];
    }
  const float bias1a = p.prm[p.conv_b[0] + (tid & 15)], bias1b = p.prm[p.conv_b[0] + 16 + (tid & 15)];
  const float bias2 = p.prm[p.conv_b[1] + 16 * ct2 + (tid & 15)];
  const float bias3 = p.prm[p.conv_b[2] + 16 * ct3 + (tid & 15)];

  // ---- conv1: output pixels (4 ty1 + m / 4, 4 tx1 + m % 4), all 32 channels ------------------------
  {
    unsigned* patch = (unsigned*)lds;   // [20][20] dwords (the 64 lanes of an A read hit 64 banks)
    patch[tid] = raw[0];
    if (tid + 256 < 400) patch[tid + 256] = raw[1];
    __syncthreads();
    ACT_STAMP(1);
    unsigned a[8];
#pragma unroll
    for (int tl = 0; tl < 8; ++tl) {
      const int T = 8 * kh1 + tl;
      a[tl] = patch[(4 * (n >> 2) + (T >> 1)) * 20 + 4 * (n & 3) + 4 * (T & 1) + kq];
    }
    act_f4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int tl = 0; tl < 8; ++tl) {
      const float4 v = dz_u8x4_to_unit(a[tl]);
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(v.x, b1[4 * tl + 0], acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(v.y, b1[4 * tl + 1], acc1, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(v.z, b1[4 * tl + 2], acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(v.w, b1[4 * tl + 3], acc1, 0, 0, 0);
    }
    act_partial_to_lds(red, wave, lane, acc0, acc1);
    __syncthreads();
    {   // wave (ct, kh): red[ct + 2 kh]; output element (m, channel 16 ct + n')
      const int m = tid >> 4;
      const int pix = (4 * ty1 + (m >> 2)) * 20 + 4 * tx1 + (m & 3);
      const float va = red[0 * 256 + tid] + red[2 * 256 + tid] + bias1a;
      const float vb = red[1 * 256 + tid] + red[3 * 256 + tid] + bias1b;
      act_store(act1 + pix * 32 + (tid & 15), va > 0.f ? va : -0.f);
      act_store(act1 + pix * 32 + 16 + (tid & 15), vb > 0.f ? vb : -0.f);
    }
  }
  ACT_STAMP(2);
  if (blk >= kActConv2Blocks) return;
  __syncthreads();   // (the partial tiles and the patch are read; LDS is reused below)

  // ---- conv2: pixels 16 pt2 .., channels 16 ct2 .. ------------------------------------------------
  {
    const int p0 = 16 * pt2, oy0 = p0 / 9;
    float2 v[10];   // 8 input rows x 20 pixels x 32 channels, LDS pixel pitch 36
    int round = 0;
    bool miss;
    act_watch(act1 + (min(2 * oy0 + 7, 19) * 20 + 19) * 32 + 31, p.spin_limit);
    do {            // conv1's outputs are their own flags
      miss = false;
#pragma unroll
      for (int i = 0; i < 10; ++i) {
        const int f = 2 * (tid + 256 * i);
        const int r = f / 640, rem = f % 640;
        v[i] = act_load2(act1 + (min(2 * oy0 + r, 19) * 20 + rem / 32) * 32 + (rem & 31));
      }
#pragma unroll
      for (int i = 0; i < 10; ++i) miss = miss || act_missing(v[i].x) || act_missing(v[i].y);
    } while (act_again(miss, round++, fail, &give_up, p.spin_limit));
    if (give_up) return;
    ACT_STAMP(3);
#pragma unroll
    for (int i = 0; i < 10; ++i) {
      const int f = 2 * (tid + 256 * i);
      const int r = f / 640, rem = f % 640;
      *(float2*)(lds + (r * 20 + rem / 32) * 36 + (rem & 31)) = v[i];
    }
    __syncthreads();
    ACT_STAMP(4);
    const int pix = min(p0 + n, 80), oy = pix / 9, ox = pix % 9;
    float4 a[8];
#pragma unroll
    for (int kw = 0; kw < 4; ++kw)
#pragma unroll
      for (int g = 0; g < 2; ++g)
        a[kw * 2 + g] = *(const float4*)(lds + ((2 * (oy - oy0) + wave) * 20 + 2 * ox + kw) * 36 +
                                         16 * g + 4 * kq);
    act_f4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].x, b2[4 * i + 0], acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].y, b2[4 * i + 1], acc1, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].z, b2[4 * i + 2], acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].w, b2[4 * i + 3], acc1, 0, 0, 0);
    }
    act_partial_to_lds(red, wave, lane, acc0, acc1);
    __syncthreads();
    {
      const int m = tid >> 4;
      const float s = ((red[tid] + red[256 + tid]) + (red[512 + tid] + red[768 + tid])) + bias2;
      if (p0 + m < 81) act_store(act2 + (p0 + m) * 64 + 16 * ct2 + (tid & 15), s > 0.f ? s : -0.f);
    }
  }
  ACT_STAMP(5);
  if (blk >= kActConv3Blocks) return;
  __syncthreads();

  // ---- conv3: pixels 16 pt3 .., channels 16 ct3 .. ------------------------------------------------
  {
    const int p0 = 16 * pt3, oy0 = p0 / 7;
    float2 v[7];    // 6 input rows x 9 pixels x 64 channels, LDS pixel pitch 68
    int round = 0;
    bool miss;
    act_watch(act2 + (min(oy0 + 5, 8) * 9 + 8) * 64 + 63, p.spin_limit);
    do {
      miss = false;
#pragma unroll
      for (int i = 0; i < 7; ++i) {
        const int f = min(2 * (tid + 256 * i), 3454);
        const int r = f / 576, rem = f % 576;
        v[i] = act_load2(act2 + (min(oy0 + r, 8) * 9 + rem / 64) * 64 + (rem & 63));
      }
#pragma unroll
      for (int i = 0; i < 7; ++i) miss = miss || act_missing(v[i].x) || act_missing(v[i].y);
    } while (act_again(miss, round++, fail, &give_up, p.spin_limit));
    if (give_up) return;
    ACT_STAMP(6);
#pragma unroll
    for (int i = 0; i < 7; ++i) {
      const int f = 2 * (tid + 256 * i);
      const int r = f / 576, rem = f % 576;
      if (f < 3456) *(float2*)(lds + (r * 9 + rem / 64) * 68 + (rem & 63)) = v[i];
    }
    __syncthreads();
    ACT_STAMP(7);
    const int pix = min(p0 + n, 48), oy = pix / 7, ox = pix % 7;
    float4 a[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) {
      const int q = 9 * wave + i, tap = q >> 2, g = q & 3;
      a[i] = *(const float4*)(lds + (((oy - oy0) + tap / 3) * 9 + ox + tap % 3) * 68 + 16 * g + 4 * kq);
    }
    act_f4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 9; ++i) {
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].x, b3[4 * i + 0], acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].y, b3[4 * i + 1], acc1, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].z, b3[4 * i + 2], acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].w, b3[4 * i + 3], acc1, 0, 0, 0);
    }
    act_partial_to_lds(red, wave, lane, acc0, acc1);
    __syncthreads();
    {
      const int m = tid >> 4;
      const float s = ((red[tid] + red[256 + tid]) + (red[512 + tid] + red[768 + tid])) + bias3;
      if (p0 + m < 49) act_store(feat + (p0 + m) * 64 + 16 * ct3 + (tid & 15), s > 0.f ? s : -0.f);
    }
  }
  ACT_STAMP(8);
}

// ---- fc1 -------------------------------------------------------------------------------------
// P = ActOneParams (noisy, 8 column groups) or DenseActParams (plain, 4 column groups)
template <class P>
__device__ __forceinline__ void act_fc1_block(const P& p, int fb, int nfc1, float* lds) {
  const int tid = threadIdx.x;
  const int split = fb / p.ncg, cg = fb % p.ncg;
  const int cq = tid & 31, kg = tid >> 5;
  const int col = 128 * cg + 4 * cq, k0 = kActFc1Rows * split + 14 * kg;
  ACT_STAMP(0);
  const unsigned gen = *act_line(p.sync, 3);
  float* const set = act_set(p, gen);
  const float* const feat = set + kActOffFeat;
  // the NEXT apply's set (last read one apply ago) goes back to all-zero bits, 256 floats at a time
  {
    float* const nxt = act_set(p, gen + 1u);
    for (int c = fb * 256; c < p.set_floats; c += nfc1 * 256) act_store(nxt + c + tid, 0.f);
  }
  float4 w[14];
  if constexpr (P::NOISY) {
    float4 sg[14];
    const float* wm = p.prm + p.fc1_mu_w + (long)k0 * p.fc1_ld + col;
    const float* ws = p.prm + p.fc1_sig_w + (long)k0 * p.fc1_ld + col;
#pragma unroll
    for (int j = 0; j < 14; ++j) {
      w[j] = dz_ld4(wm + (long)j * p.fc1_ld);
      sg[j] = dz_ld4(ws + (long)j * p.fc1_ld);
    }
    __builtin_amdgcn_sched_barrier(0);
    // this apply's position in the actor's noise stream; the block is also written out
    const uint64_t base = p.counter + (uint64_t)(*p.step) * (uint64_t)p.n_noise;
    {
      const int i = fb * 256 + tid;
      if (i < p.n_noise) p.noise[i] = dz_noise_at(p.seed, base + (uint64_t)i);
    }
    float eo[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) eo[c] = dz_noise_at(p.seed, base + (uint64_t)(p.n_fc1_out + col + c));
    const int e_in = cg >= 4 ? p.n_eps_in[1] : p.n_eps_in[0];
#pragma unroll
    for (int j = 0; j < 14; ++j) {
      const float ei = dz_noise_at(p.seed, base + (uint64_t)(e_in + k0 + j));
      // W_eff = Wmu + Wsig * (eps_in[k] * eps_out[n])   (networks.py:168-176)
      w[j].x = __builtin_fmaf(sg[j].x, ei * eo[0], w[j].x);
      w[j].y = __builtin_fmaf(sg[j].y, ei * eo[1], w[j].y);
      w[j].z = __builtin_fmaf(sg[j].z, ei * eo[2], w[j].z);
      w[j].w = __builtin_fmaf(sg[j].w, ei * eo[3], w[j].w);
    }
  } else {
    const float* wm = p.prm + p.fc1_mu_w + (long)k0 * p.fc1_ld + col;
#pragma unroll
    for (int j = 0; j < 14; ++j) w[j] = dz_ld4(wm + (long)j * p.fc1_ld);
    __builtin_amdgcn_sched_barrier(0);
  }
  ACT_STAMP(1);
  float x[14];
  {
    int round = 0;
    bool miss, give_up;
    act_watch(feat + k0 + 13, p.spin_limit);
    do {            // conv3's outputs are their own flags
      miss = false;
#pragma unroll
      for (int j = 0; j < 14; ++j) x[j] = act_load(feat + k0 + j);
#pragma unroll
      for (int j = 0; j < 14; ++j) miss = miss || act_missing(x[j]);
    } while (act_again(miss, round++, act_line(p.sync, 5), &give_up, p.spin_limit));
    if (give_up) return;
  }
  ACT_STAMP(2);
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int j = 0; j < 14; ++j) {
    acc.x = __builtin_fmaf(x[j], w[j].x, acc.x); acc.y = __builtin_fmaf(x[j], w[j].y, acc.y);
    acc.z = __builtin_fmaf(x[j], w[j].z, acc.z); acc.w = __builtin_fmaf(x[j], w[j].w, acc.w);
  }
  *(float4*)(lds + kg * 128 + 4 * cq) = acc;
  __syncthreads();
  ACT_STAMP(3);
  if (tid < 128) {
    const float s = ((lds[tid] + lds[128 + tid]) + (lds[256 + tid] + lds[384 + tid])) +
                    ((lds[512 + tid] + lds[640 + tid]) + (lds[768 + tid] + lds[896 + tid]));
    act_store(set + kActOffPart + split * p.part_ld + 128 * cg + tid, act_mark(s));
  }
  ACT_STAMP(4);
}

// ---- tail ------------------------------------------------------------------------------------
// As rainbow_act_tail_kernel (one row), with the noise drawn here and the slabs read after the
// fc1 counter is full.
__device__ __forceinline__ void act_tail_block(const ActOneParams& p, int tile, float* lds) {
  float* s_h1 = lds;            // [512]
  float* s_ein = lds + 512;     // [512]
  float* s_red = lds + 1024;    // [8][32]
  float* s_row = lds + 1280;    // [ld2 <= 1024]
  float* s_q = lds + 2304;      // [64]
  float* s_best = lds + 2368;
  int* s_arg = (int*)(lds + 2369);
  int* s_last = (int*)(lds + 2370);
  const int tid = threadIdx.x;
  ACT_STAMP(0);
  const int hsel = tile >= p.tiles0 ? 1 : 0;
  const FcHead hd = dz_pick_head(p.head, hsel);
  const int c = tid & 31, kg = tid >> 5;
  const int col = (tile - (hsel ? p.tiles0 : 0)) * 32 + c;   // within the head
  const int colc = min(col, hd.ldw - 1);
  float m[64], g[64];
  {
    const float* wmu = p.prm + hd.w_mu + (long)(kg * 64) * hd.ldw + colc;
    const float* wsg = p.prm + hd.w_sig + (long)(kg * 64) * hd.ldw + colc;
#pragma unroll
    for (int j = 0; j < 64; ++j) {   // (running pointers kept opaque: 128 precomputed addresses = 256 registers)
      m[j] = *wmu; g[j] = *wsg;
      wmu += hd.ldw; wsg += hd.ldw;
      asm volatile("" : "+v"(wmu), "+v"(wsg));
    }
  }
  float bm[2], bs[2];
#pragma unroll
  for (int e = 0; e < 2; ++e) {
    const int cc = hd.x_off + tid + 256 * e;
    bm[e] = p.prm[p.fc1_mu_b + cc]; bs[e] = p.prm[p.fc1_sig_b + cc];
  }
  const float sbw = p.prm[p.fc2_sig_b + hd.out_off + colc];
  __builtin_amdgcn_sched_barrier(0);
  const uint64_t base = p.counter + (uint64_t)(*p.step) * (uint64_t)p.n_noise;
  float be[2], ei[2];
#pragma unroll
  for (int e = 0; e < 2; ++e) {
    be[e] = dz_noise_at(p.seed, base + (uint64_t)(p.n_fc1_out + hd.x_off + tid + 256 * e));
    ei[e] = dz_noise_at(p.seed, base + (uint64_t)(hd.eps_in + tid + 256 * e));
  }
  const float eo = dz_noise_at(p.seed, base + (uint64_t)(hd.eps_out + colc));
  const float sb = sbw * dz_noise_at(p.seed, base + (uint64_t)(p.n_fc2_out + hd.out_off + colc));
  // W_eff of this thread's 64 k's, formed while the torso runs (64 registers instead of 128)
  s_ein[tid] = ei[0]; s_ein[tid + 256] = ei[1];
  __syncthreads();
#pragma unroll
  for (int j = 0; j < 64; ++j) {
    m[j] = __builtin_fmaf(g[j], s_ein[kg * 64 + j] * eo, m[j]);
    asm volatile("" : "+v"(m[j]));   // formed HERE (the compiler otherwise sinks it below the wait and keeps g alive)
  }
  ACT_STAMP(1);
  // ---- 1. h1 (this head's half): 28 slabs in slab order; the slab values are their own flags ------
  float x[2][kActFc1Splits];
  bool failed;
  {
    const float* part = act_set(p, *act_line(p.sync, 3)) + kActOffPart + hd.x_off + tid;
    int round = 0;
    bool miss, give_up;
    act_watch(part + (kActFc1Splits - 1) * 1024, p.spin_limit);
    do {
      miss = false;
      // (running pointer kept opaque: the slabs are 4 KB apart, beyond the immediate offset, and
      // 56 addresses computed ahead of the loads cost 112 registers -- the kernel spilled)
      const float* pp = part;
#pragma unroll
      for (int j = 0; j < kActFc1Splits; ++j) {
        asm volatile("" : "+v"(pp));
        x[0][j] = act_load(pp); x[1][j] = act_load(pp + 256);
        pp += 1024;
      }
#pragma unroll
      for (int e = 0; e < 2; ++e)
#pragma unroll
        for (int j = 0; j < kActFc1Splits; ++j) miss = miss || act_missing(x[e][j]);
    } while (act_again(miss, round++, act_line(p.sync, 5), &give_up, p.spin_limit));
    // A stuck seam (workgroup-uniform): the sticky word is set; this workgroup still takes its
    // ticket below, so that the ticket counter and the generation re-arm and the last workgroup
    // can tell the host (the decision comes back as kActFailedAction / NaN, never as an action).
    failed = give_up;
  }
  ACT_STAMP(2);
  if (!failed) {
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      float v = 0.f;
#pragma unroll
      for (int j = 0; j < kActFc1Splits; ++j) v += x[e][j];
      const float h = v + bm[e] + bs[e] * be[e];
      s_h1[tid + 256 * e] = h > 0.f ? h : 0.f;
    }
    __syncthreads();
    ACT_STAMP(3);
    // ---- 2. this workgroup's 32 output columns ----------------------------------------------------
    {
      float acc = 0.f;
#pragma unroll
      for (int j = 0; j < 64; ++j) {
        const int k = kg * 64 + j;
        acc = __builtin_fmaf(s_h1[k], m[j], acc);
      }
      s_red[kg * 32 + c] = acc;
    }
    __syncthreads();
    if (tid < 32 && col < hd.ldw) {
      const float o = (((s_red[c] + s_red[32 + c]) + (s_red[64 + c] + s_red[96 + c])) +
                       ((s_red[128 + c] + s_red[160 + c]) + (s_red[192 + c] + s_red[224 + c]))) + sb;
      act_store(p.fc2_out + hd.out_off + col, col < hd.N ? o : 0.f);
    }
  }
  // ---- 3. ticket: the last workgroup finishes the row ---------------------------------------------
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid == 0) {
    int last = __hip_atomic_fetch_add(act_line(p.sync, 4), 1u, DZ_ACT_RLX) == (unsigned)p.tiles - 1;
    // (a workgroup that gave up stored the sticky word before its ticket: the last one sees it)
    if (last && __hip_atomic_load(act_line(p.sync, 5), DZ_ACT_RLX) != 0u) last = 2;
    *s_last = last;
  }
  __syncthreads();
  ACT_STAMP(4);
  const int last = *s_last;
  if (!last) return;
  if (last == 1)
    for (int i = tid; i < p.ld2; i += 256) s_row[i] = act_load(p.fc2_out + i);
  __syncthreads();
  if (tid == 0) {
    // re-armed for the next apply (ordered by the kernel boundary; every poller has passed)
    const unsigned gen = *act_line(p.sync, 3);
    __hip_atomic_store(act_line(p.sync, 4), 0u, DZ_ACT_RLX);
    __hip_atomic_store(act_line(p.sync, 3), gen + 1u, DZ_ACT_RLX);
    if (p.bump) *p.bump = *p.bump + 1;
  }
  if (last == 2) {
    // visible to the host: the decision is NOT an action.  The sticky word stays set until the
    // host clears the seam area (RainbowLearner raises and resets), so every decision until
    // then comes back like this one.
    if (tid == 0) {
      const float nan = __builtin_nanf("");
      for (int a = 0; a < p.A; ++a) p.q_out[a] = nan;
      if (p.greedy_out && p.vmax_out == (float*)(p.greedy_out + 1) && ((uintptr_t)p.greedy_out & 7) == 0) {
        *(volatile unsigned long long*)p.greedy_out =
            (unsigned long long)(unsigned)kActFailedAction |
            ((unsigned long long)__builtin_bit_cast(unsigned, nan) << 32);
      } else {
        if (p.greedy_out) *p.greedy_out = kActFailedAction;
        if (p.vmax_out) *p.vmax_out = nan;
      }
    }
    return;
  }
  dz_q_from_row(s_row, p.A, p.K, p.val_off, p.support, p.q_out, p.greedy_out, p.vmax_out, s_q,
                s_best, s_arg);
  ACT_STAMP(5);
}

__global__ __launch_bounds__(256, 2) void rainbow_act_one_kernel(ActOneParams p) {
  __shared__ __attribute__((aligned(16))) float lds[kActLdsFloats];
  const int b = blockIdx.x;
  if (b < kActTorsoBlocks) act_torso_block(p, b, lds);
  else if (b < kActTorsoBlocks + kActFc1Blocks) act_fc1_block(p, b - kActTorsoBlocks, kActFc1Blocks, lds);
  else act_tail_block(p, b - kActTorsoBlocks - kActFc1Blocks, lds);
}

// ---- dense heads (DQN, double-Q, prioritized: N = A; C51: N = 51 A; QR-DQN: N = 201 A) ------------
// A workgroup per 32 output columns: h1 = relu(slab sum + b1) (every workgroup folds the 28 slabs
// itself), out = h1 W2 + b2 (ref: networks.py:206-221, 120-134, 295-363), each output stored with a
// marker as one 8-byte word into the pinned slot the host polls; the last workgroup (ticket)
// advances the generation.
__device__ __forceinline__ void dense_act_tail_block(const DenseActParams& p, int tile, float* lds) {
  float* s_h = lds;            // [512]
  float* s_red = lds + 512;    // [8][32]
  const int tid = threadIdx.x;
  ACT_STAMP(0);
  const int n = tid & 31, ks = tid >> 5, col = 32 * tile + n, nc = min(col, p.N - 1);
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
  const float b2 = p.prm[p.fc2_b + (p.bias_shared ? 0 : nc)];
  __builtin_amdgcn_sched_barrier(0);
  const unsigned gen = *act_line(p.sync, 3);
  const unsigned sticky = *act_line(p.sync, 5);   // set by an earlier decision and not yet cleared by the host
  ACT_STAMP(1);
  float x[2][kActFc1Splits];
  bool failed;
  {
    const float* part = act_set(p, gen) + kActOffPart + tid;
    int round = 0;
    bool miss, give_up;
    act_watch(part + (kActFc1Splits - 1) * 512, p.spin_limit);
    do {
      miss = false;
      const float* pp = part;
#pragma unroll
      for (int j = 0; j < kActFc1Splits; ++j) {
        asm volatile("" : "+v"(pp));
        x[0][j] = act_load(pp); x[1][j] = act_load(pp + 256);
        pp += 512;
      }
#pragma unroll
      for (int e = 0; e < 2; ++e)
#pragma unroll
        for (int j = 0; j < kActFc1Splits; ++j) miss = miss || act_missing(x[e][j]);
    } while (act_again(miss, round++, act_line(p.sync, 5), &give_up, p.spin_limit));
    failed = give_up || sticky != 0u;
  }
  ACT_STAMP(2);
  if (failed) {
    // visible to the host: the outputs are not numbers and carry the FAILED marker (2.0f); the
    // ticket below still re-arms the counter and the generation
    if (tid < 32 && col < p.N)
      p.pairs_out[col] = 0x7fc00000ull |
                         ((unsigned long long)__builtin_bit_cast(unsigned, kActFailedMarker) << 32);
  } else {
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      float v = 0.f;
#pragma unroll
      for (int j = 0; j < kActFc1Splits; ++j) v += x[e][j];
      const float h = v + (e ? b1b : b1a);
      s_h[tid + 256 * e] = h > 0.f ? h : 0.f;
    }
    __syncthreads();
    ACT_STAMP(3);
    {
      float acc = 0.f;
#pragma unroll
      for (int k = 0; k < 64; ++k) acc = __builtin_fmaf(s_h[ks * 64 + k], w[k], acc);
      s_red[ks * 32 + n] = acc;
    }
    __syncthreads();
    if (tid < 32 && col < p.N) {
      const float q = (((s_red[n] + s_red[32 + n]) + (s_red[64 + n] + s_red[96 + n])) +
                       ((s_red[128 + n] + s_red[160 + n]) + (s_red[192 + n] + s_red[224 + n]))) + b2;
      p.pairs_out[col] = (unsigned long long)__builtin_bit_cast(unsigned, q) | (0x3f800000ull << 32);
    }
  }
  // every tail workgroup has read the generation and the slabs once it takes its ticket: the last
  // one re-arms the ticket and switches the next apply to the other set
  if (tid == 0 &&
      __hip_atomic_fetch_add(act_line(p.sync, 4), 1u, DZ_ACT_RLX) == (unsigned)p.tiles - 1) {
    __hip_atomic_store(act_line(p.sync, 4), 0u, DZ_ACT_RLX);
    __hip_atomic_store(act_line(p.sync, 3), gen + 1u, DZ_ACT_RLX);
  }
  ACT_STAMP(4);
}

constexpr int kDenseActFc1Blocks = kActFc1Splits * 4;
__global__ __launch_bounds__(256, 2) void dense_act_one_kernel(DenseActParams p) {
  __shared__ __attribute__((aligned(16))) float lds[kActLdsFloats];
  const int b = blockIdx.x;
  if (b < kActTorsoBlocks) act_torso_block(p, b, lds);
  else if (b < kActTorsoBlocks + kDenseActFc1Blocks)
    act_fc1_block(p, b - kActTorsoBlocks, kDenseActFc1Blocks, lds);
  else dense_act_tail_block(p, b - kActTorsoBlocks - kDenseActFc1Blocks, lds);
}

}  // namespace
