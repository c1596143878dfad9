// Learner step of the dense-head agents (DQN, double-Q, prioritized, C51,
// QR-DQN) on one MI355X: the same conv / linear / optimiser kernels as the
// Rainbow step (dz_qnet_kernels.h) around a plain (non-noisy) two-layer head.
//   groups: g0 = online(s_tm1) [gradient], g1 = target(s_t), g2 = online(s_t)
//   (double-Q selector only).
#include "dz_torso.h"
#include "dz_row_dgrad.h"
#include "dz_act_one.h"

namespace {
constexpr int kS_dfc1 = 32;   // fc1 forward k-splits (100 rows each: dz_fc_stream_fwd3<0, 50>)
constexpr int kMaxS_ddh1 = 32; // fc2 input-gradient k-splits grow with the head width (QR: 3618 outputs)
// fc1's input gradient as a row-owning weight stream (dz_row_dgrad.h) in front of the
// layer's weight-gradient contraction, for the learners whose forward phase already left
// dh1 finished (the narrow Q heads): no split over the reduction, no slabs, no reduce
// launch.  256 workgroups x 12-13 rows, two jobs per row (512 columns), two row groups:
// with the stream's 218 VGPRs a CU holds two workgroups, so the 392 one-stage
// weight-gradient workgroups pass through the second slot while the 256 streams run.
constexpr int kDenseDgBlocks = 256;
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2)))
void dense_fc1_bwd_rows_kernel(FcWgradParams w, dim3 gw, RowDgrad q) {
  constexpr int SM = DzGemmSmem<FcWg>::ELEMS > kRdLdsFloats ? DzGemmSmem<FcWg>::ELEMS : kRdLdsFloats;
  __shared__ __attribute__((aligned(16))) float smem[SM];
  if (blockIdx.x < (unsigned)q.nblocks) row_dgrad_block<2, 0, true, 4, false>(q, blockIdx.x, smem);
  else dz_gemm_body<FcWg>(w, dz_unflatten(blockIdx.x - q.nblocks, gw), smem);
}
// The second layer's input gradient the same way (wide heads: C51, QR-DQN; up to 16 chunks of
// 256 outputs): 128 workgroups x 4 rows in front of that layer's weight-gradient
// contraction; dh1 leaves the launch finished (summed over the outputs, ReLU-masked), so
// nothing has to fold slabs and fc1's stream above can read it.
constexpr int kDenseDg2Blocks = 128;
template <int NJ0>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2)))
void dense_fc2_bwd_rows_kernel(FcWgradParams w, dim3 gw, RowDgrad q) {
  constexpr int SM = DzGemmSmem<FcWg>::ELEMS > kRdLdsFloats ? DzGemmSmem<FcWg>::ELEMS : kRdLdsFloats;
  __shared__ __attribute__((aligned(16))) float smem[SM];
  if (blockIdx.x < (unsigned)q.nblocks) row_dgrad_block<NJ0, 0, true, 4, false>(q, blockIdx.x, smem);
  else dz_gemm_body<FcWg>(w, dz_unflatten(blockIdx.x - q.nblocks, gw), smem);
}
template <int NJ0>
static inline void launch_dense_fc2_rows(const FcWgradParams& w, dim3 gw, const RowDgrad& q, hipStream_t s) {
  hipLaunchKernelGGL(dense_fc2_bwd_rows_kernel<NJ0>, dim3(q.nblocks + dz_count(gw)), dim3(256), 0, s, w, gw, q);
}
constexpr int kS_ddfeat = 8;  // fc1 input-gradient k-splits (FcDgradOp<1,2,2,1>): 8 x 4 stages (16 x 2: +1.5 us)
}

extern "C" int dz_dense_layout(int N, int shared_bias, int B, int G,
                               dz_dense_layout_t* L) {
  DZ_REQUIRE(L && N > 0 && B > 0 && B <= 1024 && (G == 1 || G == 2 || G == 3));
  L->num_outputs = N; L->shared_bias = shared_bias; L->batch = B; L->groups = G;
  L->fc1_ld = 512 + 32;  // padded pitch (see dz_rainbow_layout)
  L->fc2_ld = (int32_t)align4(N);
  int64_t o = 0;
  const int64_t cw[3] = {256 * 32, 512 * 64, 576 * 64};
  const int64_t cb[3] = {32, 64, 64};
  for (int i = 0; i < 3; ++i) {
    L->conv_w[i] = o; o = align4(o + cw[i]);
    L->conv_b[i] = o; o = align4(o + cb[i]);
  }
  L->fc1_w = o; o = align4(o + (int64_t)kFlat * L->fc1_ld);
  L->fc1_b = o; o = align4(o + kHid);
  L->fc2_w = o; o = align4(o + (int64_t)kHid * L->fc2_ld);
  L->fc2_b = o; o = align4(o + (shared_bias ? 1 : N));
  L->param_count = o;
  L->param_count_ref = 77984 + (int64_t)kFlat * kHid + kHid + (int64_t)kHid * N +
                       (shared_bias ? 1 : N);
  const int64_t GB = (int64_t)G * B, ld2 = L->fc2_ld;
  int64_t w = 0;
  auto take = [&](int64_t n) { int64_t r = w; w = align4(w + n); return r; };
  L->ws_act1 = take(GB * 400 * 32);
  L->ws_act2 = take(GB * 81 * 64);
  L->ws_feat = take(GB * kFlat);
  L->ws_fc1_part = take((int64_t)kS_dfc1 * GB * kHid);
  L->ws_h1 = take(GB * kHid);
  L->ws_fc2_part = take((int64_t)kS_fc2 * GB * ld2);
  L->ws_out = take(GB * ld2);
  L->ws_dout = take((int64_t)B * ld2);
  L->ws_dh1 = take((int64_t)B * kHid);
  int64_t dp = (int64_t)kS_ddfeat * B * kFlat;
  if ((int64_t)kMaxS_ddh1 * B * kHid > dp) dp = (int64_t)kMaxS_ddh1 * B * kHid;
  L->ws_dfeat_part = take(dp);
  L->ws_dfeat = take((int64_t)B * kFlat);
  L->ws_dact2 = take((int64_t)B * 81 * 64);
  L->ws_dact1 = take((int64_t)B * 400 * 32);
  L->ws_wgrad_part = take(torso_wgrad_part_elems());
  L->ws_norm_part = take(kNormFinal + kNormSlots);   // fused-norm partials + per-wave slots
  L->ws_scalars = take(16);
  L->ws_zeros = take(kFlat + 1024);   // stands in for the (absent) noise vectors
  L->ws_act_seams = take(kDenseActSeamWords);
  L->ws_count = w;
  return DZ_OK;
}

static void dense_heads(const dz_dense_layout_t& L, FcHead& h1, FcHead& h2) {
  h1.w_mu = L.fc1_w; h1.w_sig = L.fc1_w; h1.ldw = L.fc1_ld; h1.N = kHid; h1.K = kFlat;
  h1.x_off = 0; h1.eps_in = 0; h1.eps_out = 0; h1.out_off = 0;
  h2.w_mu = L.fc2_w; h2.w_sig = L.fc2_w; h2.ldw = L.fc2_ld; h2.N = L.num_outputs;
  h2.K = kHid; h2.x_off = 0; h2.eps_in = 0; h2.eps_out = 0; h2.out_off = 0;
}

// torso + head for G groups; outputs in ws_out rows [G*B][fc2_ld].
// `head`: narrow Q heads (num_outputs <= 32) finish in ONE launch after the fc1 weight
// stream (dense_head_kernel: fc1 epilogue + second layer + TD loss or q-values); the
// caller presets the mode-specific fields, the common ones are filled in here.
static inline bool dense_head_fused(const dz_dense_layout_t& L) { return L.num_outputs <= 32; }
static int dense_forward(const dz_dense_layout_t& L, int G, int B, const float* const* prm,
                         const uint8_t* const* in, float* ws, hipStream_t s,
                         DenseHeadParams* head = nullptr, bool skip_fc2_epilogue = false) {
  int rc;
  const float* zeros = ws + L.ws_zeros;
  FcHead h1, h2;
  dense_heads(L, h1, h2);
  const TorsoBufs T = {L.conv_w, L.conv_b, ws + L.ws_act1, ws + L.ws_act2, ws + L.ws_feat};
  rc = torso_forward(T, G, B, prm, in, s);
  if (rc) return rc;
  const float* p3[3] = {prm[0], prm[G > 1 ? 1 : 0], prm[G > 2 ? 2 : 0]};
  {  // fc1 (3136 -> 512): weight-streaming kernel when the batch fits one tile
    if (B <= 32) {
      // one weight stream per parameter set (online is shared by the double-Q
      // selector apply), 4 strips x 32 splits x sets workgroups
      FcStreamFwd3Params q;
      q.x = ws + L.ws_feat; q.ldx = kFlat; q.M = B; q.noisy = 0; q.G = G;
      const float* zn[3] = {zeros, zeros, zeros};
      const int ns = dz_fc3_assign_sets(q, G, p3, zn);
      DZ_REQUIRE(ns > 0);
      q.head[0] = h1; q.head[1] = h1;
      q.part = ws + L.ws_fc1_part; q.ldo = kHid;
      q.rows_per_split = ((kFlat + kS_dfc1 - 1) / kS_dfc1 + 3) & ~3;
      DZ_REQUIRE(q.rows_per_split <= 100);
      q.xcd_order = 0;
      hipLaunchKernelGGL((dz_fc_stream_fwd3<0, 50>), dim3(kHid / 128, kS_dfc1, ns), dim3(256),
                         (size_t)q.rows_per_split * (2 * 32 + 2) * sizeof(float), s, q);
      DZ_LAUNCH_CHECK();
    } else {
      FcFwdParams p;
      p.x = ws + L.ws_feat; p.ldx = kFlat; p.M = B; p.G = G; p.NH = 1; p.S = kS_dfc1;
      p.noisy = 0;
      for (int g = 0; g < 3; ++g) { p.params[g] = p3[g]; p.noise[g] = zeros; }
      p.head[0] = h1; p.head[1] = h1;
      p.part = ws + L.ws_fc1_part; p.ldo = kHid;
      rc = dz_launch_gemm<FcFwdOp<1, 2, 2, 4, 0>>(p, dim3(kHid / FcFwd::BN, (B + 31) / 32, G * kS_dfc1), s);
      if (rc) return rc;
    }
    DZ_PROF(s, "fc1_fwd");
    if (head && dense_head_fused(L)) {
      DenseHeadParams& q = *head;
      DZ_REQUIRE(G < 3 || p3[2] == p3[0]);  // the kernel reads groups 0 and 2 from one set
      q.part = ws + L.ws_fc1_part; q.S = kS_dfc1; q.rows = G * B; q.B = B; q.G = G;
      for (int g = 0; g < 3; ++g) q.prm[g] = p3[g];
      q.fc1_b = L.fc1_b; q.fc2_w = L.fc2_w; q.fc2_b = L.fc2_b; q.ld2 = L.fc2_ld;
      q.N = L.num_outputs; q.bias_shared = L.shared_bias;
      q.h1 = ws + L.ws_h1; q.out = ws + L.ws_out;
      hipLaunchKernelGGL(dense_head_kernel, dim3(B), dim3(512), 0, s, q);
      DZ_LAUNCH_CHECK();
      DZ_PROF(s, q.mode == 1 ? "head+loss" : "head");
      return DZ_OK;
    }
    hipLaunchKernelGGL(fc_epilogue_kernel, dim3(kHid / 64, G * B), dim3(256), 0, s,
                       ws + L.ws_fc1_part, kS_dfc1, G * B, kHid, kHid, B, p3[0], p3[1],
                       p3[2], (long)L.fc1_b, (long)-1, zeros, zeros, zeros, 0, 1,
                       ws + L.ws_h1, 0);
    DZ_LAUNCH_CHECK();
    DZ_PROF(s, "fc1_epilogue");
  }
  {  // fc2 (512 -> num_outputs), vector or shared scalar bias
    const int ld2 = L.fc2_ld, N = L.num_outputs;
    FcFwdParams p;
    p.x = ws + L.ws_h1; p.ldx = kHid; p.M = B; p.G = G; p.NH = 1; p.S = kS_fc2;
    p.noisy = 0;
    for (int g = 0; g < 3; ++g) { p.params[g] = p3[g]; p.noise[g] = zeros; }
    p.head[0] = h2; p.head[1] = h2;
    p.part = ws + L.ws_fc2_part; p.ldo = ld2;
    rc = dz_launch_gemm<FcFwdOp<1, 2, 2, 4, 0>>(p, dim3((N + FcFwd::BN - 1) / FcFwd::BN, (B + 31) / 32,
                                      G * kS_fc2), s);
    if (rc) return rc;
    DZ_PROF(s, "fc2_fwd");
    if (skip_fc2_epilogue) return DZ_OK;  // the loss kernel folds the partial slabs itself
    hipLaunchKernelGGL(fc_epilogue_kernel, dim3((N + 63) / 64, G * B), dim3(256), 0, s,
                       ws + L.ws_fc2_part, kS_fc2, G * B, N, ld2, B, p3[0], p3[1], p3[2],
                       (long)L.fc2_b, (long)-1, zeros, zeros, zeros, 0, 0,
                       ws + L.ws_out, L.shared_bias);
    DZ_LAUNCH_CHECK();
    DZ_PROF(s, "fc2_epilogue");
  }
  return DZ_OK;
}

extern "C" int dz_dense_learn(const dz_dense_args_t* a, int phases, dz_stream_t stream) {
  DZ_REQUIRE(a && a->online && a->target && a->ws && a->s_tm1 && a->s_t && a->a_tm1 &&
             a->r_t && a->discount_t && a->losses);
  DZ_REQUIRE(a->loss >= DZ_LOSS_Q && a->loss <= DZ_LOSS_QUANTILE);
  const int G = a->loss == DZ_LOSS_DOUBLE_Q ? 3 : 2;
  const int B = a->batch, A = a->num_actions, N = a->num_outputs;
  if (a->loss == DZ_LOSS_CATEGORICAL)
    DZ_REQUIRE(a->aux && a->num_atoms > 0 && a->num_atoms <= 64 && A <= 256 &&
               N == A * a->num_atoms);
  else if (a->loss == DZ_LOSS_QUANTILE)
    DZ_REQUIRE(a->aux && a->num_atoms > 0 && a->num_atoms <= 256 && A <= 256 &&
               N == A * a->num_atoms);
  else
    DZ_REQUIRE(N == A);
  dz_dense_layout_t L;
  int rc = dz_dense_layout(N, a->shared_bias, B, G, &L);
  if (rc) return rc;
  hipStream_t s = dz_s(stream);
  float* ws = a->ws;
  const int ld2 = L.fc2_ld;
  const float* zeros = ws + L.ws_zeros;
  const float* prm[3] = {a->online, a->target, a->online};
  const uint8_t* in[3] = {a->s_tm1, a->s_t, a->s_t};
  FcHead h1, h2;
  dense_heads(L, h1, h2);
  if (g_dz_prof_on) dz_prof_begin(s);

  if (phases & DZ_PHASE_FORWARD) {
    float* out = ws + L.ws_out;
    float* dout = ws + L.ws_dout;
    const bool q_loss = a->loss == DZ_LOSS_Q || a->loss == DZ_LOSS_DOUBLE_Q;
    if (q_loss && dense_head_fused(L)) {
      // one value per action: fc1 epilogue + second layer + TD loss + dh1 in one launch
      DenseHeadParams hp = {};
      hp.mode = 1; hp.sel_group = a->loss == DZ_LOSS_DOUBLE_Q ? 2 : 1; hp.tgt_group = 1;
      hp.a_tm1 = a->a_tm1; hp.r_t = a->r_t; hp.d_t = a->discount_t; hp.weights = a->weights;
      hp.bound = a->grad_error_bound; hp.dout = dout; hp.td_out = a->losses;
      hp.prio_out = a->priorities;
      // shared scalar bias: its gradient is the sum of all of dout = sum_b rowsum[b]
      // (B values for finalize instead of a serial walk over B*ld2); the fc2 slab
      // buffer is idle on this path
      hp.dout_rowsum = ws + L.ws_fc2_part;
      hp.dh1 = ws + L.ws_dh1;   // second-layer input gradient (see the backward phase)
      rc = dense_forward(L, G, B, prm, in, ws, s, &hp);
      if (rc) return rc;
    } else {
      // C51: the fc2 split-K slabs (+ bias) are folded by the loss kernel (as in
      // dz_rainbow.hip) instead of an epilogue launch
      const bool fold_fc2 = a->loss == DZ_LOSS_CATEGORICAL && !a->shared_bias &&
                            (size_t)3 * ld2 * sizeof(float) <= 48 * 1024 && kS_fc2 <= 8;
      rc = dense_forward(L, G, B, prm, in, ws, s, nullptr, fold_fc2);
      if (rc) return rc;
      switch (a->loss) {
        case DZ_LOSS_Q:
        case DZ_LOSS_DOUBLE_Q:   // more than 32 actions: separate launches
          hipLaunchKernelGGL(td_loss_kernel, dim3((B + 63) / 64), dim3(64), 0, s, out, ld2, B,
                             A, a->loss == DZ_LOSS_DOUBLE_Q ? 2 : 1, 1, a->a_tm1, a->r_t,
                             a->discount_t, a->weights, a->grad_error_bound, dout, a->losses,
                             a->priorities);
          break;
        case DZ_LOSS_CATEGORICAL: {
          DZ_REQUIRE(a->weights);  // c51 passes all-ones weights
          float* prio = a->priorities ? a->priorities : (ws + L.ws_dfeat_part);
          if (fold_fc2) {
            HeadPre pre = {};
            pre.part = ws + L.ws_fc2_part; pre.S = kS_fc2; pre.rows = G * B; pre.groups = G;
            for (int g = 0; g < 3; ++g) { pre.prm[g] = prm[g < G ? g : 0]; pre.nz[g] = zeros; }
            pre.b_sig = L.fc2_b; pre.eps_out = 0; pre.plain_bias = 1;
            hipLaunchKernelGGL(rainbow_head_loss_kernel<1>, dim3(B), dim3(256),
                               (size_t)3 * ld2 * sizeof(float), s, out, ld2, 0, B, A,
                               a->num_atoms, 0, 1, 1, a->a_tm1, a->r_t, a->discount_t,
                               a->weights, a->aux, dout, a->losses, prio, (float*)nullptr,
                               (float*)nullptr, pre);
          } else {
            hipLaunchKernelGGL(rainbow_head_loss_kernel<0>, dim3(B), dim3(256), 0, s, out, ld2,
                               0, B, A, a->num_atoms, 0, 1, 1, a->a_tm1, a->r_t, a->discount_t,
                               a->weights, a->aux, dout, a->losses, prio, (float*)nullptr,
                               (float*)nullptr, HeadPre{});
          }
          break;
        }
        case DZ_LOSS_QUANTILE:
          hipLaunchKernelGGL(quantile_loss_kernel, dim3(B), dim3(1024), 0, s, out, ld2, B, A,
                             a->num_atoms, 1, 1, a->aux, a->a_tm1, a->r_t, a->discount_t,
                             a->huber, dout, a->losses);
          break;
      }
      DZ_LAUNCH_CHECK();
      DZ_PROF(s, "loss");
    }
  }

  bool rms_in_finalize = false;
  bool next_sample_done = false;
  if (a->next_sample) {
    DZ_REQUIRE((phases & DZ_PHASE_BACKWARD) && (phases & DZ_PHASE_OPTIMIZER));
    // a prioritized sample must follow THIS step's write-back into the same tree
    // (prioritized/agent.py:187-206)
    DZ_REQUIRE(!a->next_sample->args.node || a->prio_node == a->next_sample->args.node);
  }
  bool fc1_onfly = false;   // fc1's weight gradient is formed inside the RMSProp launch
  bool norm_fused = false;  // this call's backward phase left the norm partials (Adam)
  int n_final = 0;   // fused-norm partials left by this call's backward phase (Adam)
  if (phases & DZ_PHASE_BACKWARD) {
    DZ_REQUIRE(a->grad);
    float* grad = a->grad;
    int s_dh1 = (N + 127) / 128;  // one split per 128 outputs of reduction depth
    s_dh1 = s_dh1 < kS_dh1 ? kS_dh1 : (s_dh1 > kMaxS_ddh1 ? kMaxS_ddh1 : s_dh1);
    // narrow Q heads: the fused forward kernel already left dh1, and the second layer's
    // weight gradient (512 x N, B terms each) is a job of the finalize launch
    const bool q_fused = (a->loss == DZ_LOSS_Q || a->loss == DZ_LOSS_DOUBLE_Q) &&
                         dense_head_fused(L);
    // Adam (clip by global norm): the weight-gradient kernels leave per-wave sums of squares
    // in sq_slots (fc2 first, then fc1), finalize folds them and adds its own, so that
    // ws_norm_part[0..n_final) is the partial list for adam_kernel -- no sumsq launch over
    // the whole gradient (as in dz_rainbow.hip)
    // (heads too wide for the slot array -- beyond ~21k outputs -- keep the separate
    // sumsq launch over the stored gradient)
    float* sq_final = ws + L.ws_norm_part;
    float* sq_slots = sq_final + kNormFinal;
    const int fc2_nx = (N + FcWg::BN - 1) / FcWg::BN, fc2_ny = kHid / FcWg::BM;
    const int fc2_slots = q_fused ? 0 : fc2_nx * fc2_ny * 4;
    const int fc1_slots = (kHid / FcWg::BN) * (kFlat / FcWg::BM) * 4;
    const bool fused_norm = a->optimizer == DZ_OPT_ADAM && fc2_slots + fc1_slots <= kNormSlots;
    norm_fused = fused_norm;
    {  // fc2: weight gradient + input gradient -> dh1 (relu(h1) mask)
      FcWgradParams w;
      w.x = ws + L.ws_h1; w.ldx = kHid; w.dy = ws + L.ws_dout; w.ldy = ld2; w.M = B;
      w.NH = 1; w.noisy = 0; w.noise = zeros; w.head[0] = h2; w.head[1] = h2;
      w.grad = grad;
      if (fused_norm) { w.sumsq = sq_slots; w.sq_nx = fc2_nx; w.sq_ny = fc2_ny; }
      FcDgradParams d;
      d.dy = ws + L.ws_dout; d.ldy = ld2; d.M = B; d.NH = 1; d.S = s_dh1; d.noisy = 0;
      d.params = a->online; d.noise = zeros; d.head[0] = h2; d.head[1] = h2;
      // the usual head (s_dh1 == kS_dh1): the slabs go to the idle fc1 forward slab
      // buffer and are summed + ReLU-masked by the loaders of the fc1 backward launch
      // (DyParts, as in dz_rainbow.hip): no dh1 reduction launch
      // wide heads, one batch tile: the input gradient as a row-owning stream (dh1 finished)
      const int nj2 = (N + 255) / 256;
      const bool rows2 = !q_fused && B <= 32 && nj2 <= 16;
      const bool fold = !rows2 && !q_fused && s_dh1 == kS_dh1 &&
                        (int64_t)kS_dh1 * B * kHid <= (int64_t)kS_dfc1 * G * B * kHid;
      d.part = fold ? ws + L.ws_fc1_part : ws + L.ws_dfeat_part;
      d.ldo = kHid; d.K = kHid; d.x_off = 0;
      if (rows2) {
        RowDgrad q = {};
        q.params = a->online; q.noise = nullptr; q.head[0] = h2; q.head[1] = h2;
        q.head[0].out_off = 0; q.head[1].N = 0;
        q.dy = ws + L.ws_dout; q.ldy = ld2; q.mask = ws + L.ws_h1; q.out = ws + L.ws_dh1;
        q.ldo = kHid; q.out_col[0] = 0; q.out_col[1] = 0; q.same_out = 1;
        q.M = B; q.K = kHid; q.nblocks = kDenseDg2Blocks;
        static_assert(kHid / kDenseDg2Blocks * 4 <= 32, "rows x jobs per workgroup");
        const dim3 gw2((N + FcWg::BN - 1) / FcWg::BN, kHid / FcWg::BM, 1);
        switch (nj2) {
          case 1: launch_dense_fc2_rows<1>(w, gw2, q, s); break;
          case 2: launch_dense_fc2_rows<2>(w, gw2, q, s); break;
          case 3: launch_dense_fc2_rows<3>(w, gw2, q, s); break;
          case 4: launch_dense_fc2_rows<4>(w, gw2, q, s); break;
          default:   // wider: run-time job count, as many rows per workgroup as the LDS block holds
            q.nblocks = (kHid + 32 / nj2 - 1) / (32 / nj2);
            DZ_REQUIRE(row_dgrad_max_rows(q) * nj2 <= 32);
            launch_dense_fc2_rows<0>(w, gw2, q, s);
            break;
        }
        DZ_LAUNCH_CHECK();
        DZ_PROF(s, "fc2_wgrad+dgrad");
      } else if (!q_fused) {
        rc = dz_launch_gemm2<FcWg, FcDgradOp<1, 2, 2, 4, 1, 1, 0>>(
            w, dim3((N + FcWg::BN - 1) / FcWg::BN, kHid / FcWg::BM, 1), d,
            dim3(kHid / FcDg::BN, (B + 31) / 32, s_dh1), s);
        if (rc) return rc;
        DZ_PROF(s, "fc2_wgrad+dgrad");
      }
      if (!fold && !q_fused && !rows2) {
        hipLaunchKernelGGL(reduce_parts_kernel, dim3((B * kHid + 63) / 64), dim3(256), 0, s,
                           ws + L.ws_dfeat_part, s_dh1, (long)B * kHid, ws + L.ws_h1,
                           ws + L.ws_dh1);
        DZ_LAUNCH_CHECK();
        DZ_PROF(s, "dh1_reduce");
      }
      // fc1
      FcWgradParams w1;
      w1.x = ws + L.ws_feat; w1.ldx = kFlat; w1.dy = ws + L.ws_dh1; w1.ldy = kHid; w1.M = B;
      w1.NH = 1; w1.noisy = 0; w1.noise = zeros; w1.head[0] = h1; w1.head[1] = h1;
      w1.grad = grad;
      if (fused_norm) {
        w1.sumsq = sq_slots + fc2_slots; w1.sq_nx = kHid / FcWg::BN; w1.sq_ny = kFlat / FcWg::BM;
      }
      FcDgradParams d1;
      d1.dy = ws + L.ws_dh1; d1.ldy = kHid; d1.M = B; d1.NH = 1; d1.S = kS_ddfeat; d1.noisy = 0;
      d1.params = a->online; d1.noise = zeros; d1.head[0] = h1; d1.head[1] = h1;
      d1.part = ws + L.ws_dfeat_part; d1.ldo = kFlat; d1.K = kFlat; d1.x_off = 0;
      // weight gradient and input gradient in ONE launch (as in dz_rainbow.hip)
      const dim3 gw(kHid / FcWg::BN, kFlat / FcWg::BM, 1), gdd(kFlat / 64, (B + 31) / 32, kS_ddfeat);
      const bool rows = (q_fused || rows2) && B <= 32;   // dh1 is finished: row-owning stream
      // fc1's weight gradient formed inside the RMSProp launch instead of stored (RmsOnFly):
      // this ONE call runs backward + optimiser, nobody asked for the full gradient vector
      fc1_onfly = rows && (phases & DZ_PHASE_OPTIMIZER) && a->optimizer != DZ_OPT_ADAM &&
                  !a->keep_all_grads;
      if (rows) {
        RowDgrad q = {};
        q.params = a->online; q.noise = nullptr; q.head[0] = h1; q.head[1] = h1;
        q.head[0].out_off = 0; q.head[1].N = 0;
        q.dy = ws + L.ws_dh1; q.ldy = kHid; q.mask = ws + L.ws_feat; q.out = ws + L.ws_dfeat;
        q.ldo = kFlat; q.out_col[0] = 0; q.out_col[1] = 0; q.same_out = 1;
        q.M = B; q.K = kFlat; q.nblocks = kDenseDgBlocks;
        static_assert(kHid == 512, "two 256-column jobs per row");
        static_assert((kFlat + kDenseDgBlocks - 1) / kDenseDgBlocks * 2 <= 32 &&
                      (kFlat + kDenseDgBlocks - 1) / kDenseDgBlocks * 32 <= 512, "rows x jobs per workgroup");
        hipLaunchKernelGGL(dense_fc1_bwd_rows_kernel,
                           dim3(kDenseDgBlocks + (fc1_onfly ? 0u : dz_count(gw))), dim3(256), 0, s,
                           w1, gw, q);
        DZ_LAUNCH_CHECK();
        rc = DZ_OK;
      } else if (fold) {
        w1.dyp.part = ws + L.ws_fc1_part; w1.dyp.stride = (long)B * kHid;
        w1.dyp.mask = ws + L.ws_h1; w1.dyp.out = ws + L.ws_dh1;
        d1.dyp = w1.dyp; d1.dyp.out = nullptr;
        rc = dz_launch_gemm2<FcWgradOp<2, 2, 1, 2, kS_dh1>, FcDgradOp<1, 2, 2, 1, 1, 1, 0, kS_dh1>>(
            w1, gw, d1, gdd, s);
      } else {
        rc = dz_launch_gemm2<FcWg, FcDgradOp<1, 2, 2, 1, 1, 1, 0>>(w1, gw, d1, gdd, s);
      }
      if (rc) return rc;
      DZ_PROF(s, "fc1_wgrad+dgrad");
      if (!rows) {
        hipLaunchKernelGGL(reduce_parts_kernel, dim3((B * kFlat + 63) / 64), dim3(256), 0, s,
                           ws + L.ws_dfeat_part, kS_ddfeat, (long)B * kFlat, ws + L.ws_feat,
                           ws + L.ws_dfeat);
        DZ_LAUNCH_CHECK();
        DZ_PROF(s, "dfeat_reduce");
      }
    }
    ReduceJob conv_jobs[3];
    const TorsoBufs T = {L.conv_w, L.conv_b, ws + L.ws_act1, ws + L.ws_act2, ws + L.ws_feat};
    PrioUpdateParams prio_q = {};
    if (a->prio_node) {
      DZ_REQUIRE(a->priorities && a->prio_ids && a->prio_status &&
                 dz_is_pow2(a->prio_cap_pow2) && a->prio_capacity > 0 &&
                 a->prio_capacity <= a->prio_cap_pow2 && a->prio_exponent >= 0.0 && B <= 256);
      prio_q = {a->prio_node, a->prio_cap_pow2, a->prio_capacity, 0, 0, a->prio_ids,
                a->priorities, 1, a->prio_exponent, B, a->prio_max_seen, a->prio_status, 0};
    }
    rc = torso_backward(T, B, a->online, a->s_tm1, ws + L.ws_dfeat, ws + L.ws_dact2,
                        ws + L.ws_dact1, ws + L.ws_wgrad_part, grad, conv_jobs, s,
                        a->prio_node ? &prio_q : nullptr);
    if (rc) return rc;
    {
      FinalizeJobs J;
      for (int j = 0; j < 3; ++j) J.r[j] = conv_jobs[j];
      unsigned acc = 0;
      for (int j = 0; j < 3; ++j) { acc += (unsigned)((J.r[j].n + 63) / 64); J.r_end[j] = acc; }
      J.c[0] = {ws + L.ws_dh1, B, kHid, kHid, grad + L.fc1_b, nullptr, nullptr};
      // (q_fused: the forward phase left the row sums)
      if (a->shared_bias && q_fused)  // one scalar: the sum of the per-sample row sums
        J.c[1] = {ws + L.ws_fc2_part, B, 1, 1, grad + L.fc2_b, nullptr, nullptr};
      else if (a->shared_bias)  // one scalar: sum over every element of dout
        J.c[1] = {ws + L.ws_dout, B * ld2, 1, 1, grad + L.fc2_b, nullptr, nullptr};
      else
        J.c[1] = {ws + L.ws_dout, B, N, ld2, grad + L.fc2_b, nullptr, nullptr};
      J.c_tiles[0] = kHid / 64;
      J.c_tiles[1] = (unsigned)(((a->shared_bias ? 1 : N) + 63) / 64);
      if (q_fused) {
        J.o_x = ws + L.ws_h1; J.o_dy = ws + L.ws_dout; J.o_out = grad + L.fc2_w;
        J.o_B = B; J.o_ld = ld2; J.o_tiles = (unsigned)(kHid * ld2 / 64);
      }
      // RMSProp needs no global norm: when this call also runs the optimiser, every
      // small gradient is applied where finalize produces it and the flat update of the
      // GEMM-written ranges (fc1 weights; fc2 weights on the wide-head path) shares the launch
      unsigned presum = 0;
      if (fused_norm) {
        presum = (unsigned)((fc2_slots + fc1_slots + 1023) / 1024);
        n_final = (int)(acc + J.c_tiles[0] + J.c_tiles[1] + J.o_tiles + presum);
        DZ_REQUIRE(n_final <= kNormFinal);
        J.sumsq = sq_final; J.presum_src = sq_slots; J.presum_n = fc2_slots + fc1_slots;
        J.bump_count = (phases & DZ_PHASE_OPTIMIZER) ? a->opt_count : nullptr;
      }
      rms_in_finalize = (phases & DZ_PHASE_OPTIMIZER) && a->optimizer != DZ_OPT_ADAM;
      if (rms_in_finalize) {
        DZ_REQUIRE(a->opt_m && a->opt_v);
        J.rms.p = a->online; J.rms.mu = a->opt_m; J.rms.nu = a->opt_v; J.rms.grad = grad;
        J.rms.lr = a->lr; J.rms.decay = a->decay_or_b1; J.rms.eps = a->eps;
        J.rms.lo4[0] = L.fc1_w >> 2; J.rms.n4[0] = fc1_onfly ? 0 : ((int64_t)kFlat * L.fc1_ld) >> 2;
        if (!q_fused) { J.rms.lo4[1] = L.fc2_w >> 2; J.rms.n4[1] = ((int64_t)kHid * ld2) >> 2; }
        J.rms.flat_blocks = J.rms.n4[0] + J.rms.n4[1] > 0 ? (fc1_onfly ? 256 : 1024) : 0;
        if (fc1_onfly) {
          J.of.feat = ws + L.ws_feat; J.of.dh1 = ws + L.ws_dh1; J.of.B = B;
          J.of.w = L.fc1_w; J.of.ld = L.fc1_ld; J.of.blocks = kRofBlocks;
          DZ_REQUIRE((L.fc1_w & 3) == 0 && (L.fc1_ld & 3) == 0);
        }
      }
      const unsigned fin_blocks = acc + J.c_tiles[0] + J.c_tiles[1] + J.o_tiles + J.of.blocks +
                                  J.rms.flat_blocks + presum;
      if (a->next_sample && rms_in_finalize) {
        // RMSProp lives in this launch: it also carries sample(k+1) + gather(k+1) (the
        // write-back, if any, rode in conv3's backward launch above)
        SampleGatherParams q;
        unsigned sgb = 0;
        rc = sample_gather_from_desc(a->next_sample, q, &sgb);
        if (rc) return rc;
        hipLaunchKernelGGL(finalize_grads_sg_kernel, dim3(sgb + fin_blocks), dim3(256), 0, s, J, q, sgb);
        next_sample_done = true;
      } else {
        hipLaunchKernelGGL(finalize_grads_kernel, dim3(fin_blocks), dim3(256), 0, s, J);
      }
      DZ_LAUNCH_CHECK();
      DZ_PROF(s, rms_in_finalize ? "finalize+rmsprop" : "finalize_grads");
    }
  }

  if (phases & DZ_PHASE_OPTIMIZER) {
    DZ_REQUIRE(a->grad && a->opt_m && a->opt_v && a->opt_count);
    float* sc = ws + L.ws_scalars;
    // loss scalar: Q/double-Q report 0.5 td^2 w in DZ_SC_LOSS via weights trick is
    // not needed by the reference (no loss statistic is logged); gnorm is.
    const float* wts = a->weights ? a->weights : zeros;
    if (a->optimizer == DZ_OPT_ADAM) {
      int nparts = n_final;
      if (!norm_fused) {  // optimiser alone / very wide head: norm from the stored gradient
        hipLaunchKernelGGL(sumsq_kernel, dim3(kNormBlocks), dim3(256), 0, s, a->grad,
                           (long)L.param_count, ws + L.ws_norm_part, a->opt_count);
        DZ_LAUNCH_CHECK();
        DZ_PROF(s, "grad_sumsq");
        nparts = kNormBlocks;
      }
      if (a->next_sample) {
        SampleGatherParams q;
        unsigned sgb = 0;
        rc = sample_gather_from_desc(a->next_sample, q, &sgb);
        if (rc) return rc;
        hipLaunchKernelGGL(adam_sg_kernel, dim3(sgb + 1536), dim3(256), 0, s, a->online, a->grad,
                           a->opt_m, a->opt_v, (long)(L.param_count >> 2),
                           ws + L.ws_norm_part, nparts, a->opt_count, a->losses, wts, B,
                           sc, a->lr, a->decay_or_b1, a->b2, a->eps, a->max_norm, DerivedGrad{}, q, sgb);
        next_sample_done = true;
      } else {
        hipLaunchKernelGGL(adam_kernel, dim3(2048), dim3(256), 0, s, a->online, a->grad,
                           a->opt_m, a->opt_v, (long)(L.param_count >> 2),
                           ws + L.ws_norm_part, nparts, a->opt_count, a->losses, wts, B,
                           sc, a->lr, a->decay_or_b1, a->b2, a->eps, a->max_norm);
      }
      DZ_LAUNCH_CHECK();
      DZ_PROF(s, "adam");
    } else if (!rms_in_finalize) {
      hipLaunchKernelGGL(rmsprop_kernel, dim3(1024), dim3(256), 0, s, a->online, a->grad,
                         a->opt_m, a->opt_v, (long)(L.param_count >> 2), a->lr,
                         a->decay_or_b1, a->eps);
      DZ_LAUNCH_CHECK();
      DZ_PROF(s, "rmsprop");
    }
  }
  DZ_REQUIRE(!a->next_sample || next_sample_done);
  return DZ_OK;
}

extern "C" int dz_dense_apply(int num_actions, int num_outputs, int shared_bias, int batch,
                              const float* params, const uint8_t* states, float* ws,
                              float* out, float* q_values_out, int32_t* greedy_out,
                              float* vmax_out, dz_stream_t stream) {
  DZ_REQUIRE(params && states && ws);
  dz_dense_layout_t L;
  int rc = dz_dense_layout(num_outputs, shared_bias, batch, 1, &L);
  if (rc) return rc;
  hipStream_t s = dz_s(stream);
  const float* prm[3] = {params, params, params};
  const uint8_t* in[3] = {states, states, states};
  const bool prof = g_dz_prof_on;
  g_dz_prof_on = false;
  DenseHeadParams hp = {};
  const bool fused = dense_head_fused(L);
  if (fused && q_values_out) {
    DZ_REQUIRE(num_outputs == num_actions);
    hp.mode = 2; hp.q_values = q_values_out; hp.greedy = greedy_out; hp.vmax = vmax_out;
  }
  rc = dense_forward(L, 1, batch, prm, in, ws, s, &hp);
  g_dz_prof_on = prof;
  if (rc) return rc;
  if (out)
    DZ_HIP_CHECK(hipMemcpy2DAsync(out, (size_t)num_outputs * sizeof(float), ws + L.ws_out,
                                  (size_t)L.fc2_ld * sizeof(float),
                                  (size_t)num_outputs * sizeof(float), batch,
                                  hipMemcpyDefault, s));  // `out` may be pinned host memory
  if (q_values_out && !fused) {
    DZ_REQUIRE(num_outputs == num_actions);
    hipLaunchKernelGGL(dense_q_values_kernel, dim3((batch + 63) / 64), dim3(64), 0, s,
                       ws + L.ws_out, L.fc2_ld, batch, num_actions, q_values_out,
                       greedy_out, vmax_out);
    DZ_LAUNCH_CHECK();
  }
  return DZ_OK;
}

// The dense-head actor's decision for ONE observation as ONE launch (dz_act_one.h), any head
// width (DQN / double-Q / prioritized: num_outputs = A; C51: 51 A; QR-DQN: 201 A;
// ref: dqn/agent.py:121-131, c51/agent.py:118-126, qrdqn/agent.py:121-129).  Every head output
// lands in `pairs_out` (pinned, device-mapped host memory or device memory) as ONE 8-byte word
// {float value, float 1.0f}: a host that cleared the words before the call reads them with plain
// loads until all markers are set, then forms the q-values as the reference's network does.
extern "C" int dz_dense_act(int num_outputs, int shared_bias, const float* params,
                            const uint8_t* state, float* ws, void* pairs_out,
                            dz_stream_t stream) {
  DZ_REQUIRE(params && state && ws && pairs_out && num_outputs > 0);
  DZ_REQUIRE(((uintptr_t)pairs_out & 7) == 0);
  dz_dense_layout_t L;
  int rc = dz_dense_layout(num_outputs, shared_bias, 1, 1, &L);
  if (rc) return rc;
  DenseActParams q;
  q.obs = state; q.prm = params;
  for (int i = 0; i < 3; ++i) { q.conv_w[i] = L.conv_w[i]; q.conv_b[i] = L.conv_b[i]; }
  q.sync = reinterpret_cast<unsigned*>(ws + L.ws_act_seams);   // zero in a fresh workspace
  q.set_floats = act_set_floats(512); q.ncg = 4; q.part_ld = 512;
  q.spin_limit = g_dz_act_spin_limit;
  q.fc1_mu_w = L.fc1_w; q.fc1_ld = L.fc1_ld;
#ifdef DZ_ACT_STAMPS
  q.dbg = reinterpret_cast<long long*>(ws + L.ws_dfeat_part);
#endif
  q.fc1_b = L.fc1_b; q.fc2_w = L.fc2_w; q.fc2_b = L.fc2_b; q.ld2 = L.fc2_ld;
  q.N = num_outputs; q.bias_shared = shared_bias;
  q.tiles = (num_outputs + 31) / 32;
  q.pairs_out = (unsigned long long*)pairs_out;
  hipLaunchKernelGGL(dense_act_one_kernel,
                     dim3((unsigned)(kActTorsoBlocks + kDenseActFc1Blocks + q.tiles)), dim3(256),
                     0, dz_s(stream), q);
  DZ_LAUNCH_CHECK();
  return DZ_OK;
}
