// Rainbow learner step on one MI355X (ref: rainbow/agent.py:85-121).
//
// Launch sequence of the one-call step at batch <= 32 (11 launches, all on the caller's
// stream, no host sync; DESIGN.md 4 has the table):
//   forward : conv1 (+ side blocks: the step's noise draw, the head launch's seam buffers back
//             to all-zero bits) -> conv2 -> conv3 (the 3 applies batched as groups) -> fc1
//             (noisy adv1|val1: one weight stream per parameter set, W_eff in registers,
//             split-K slabs)
//   head    : ONE multi-role launch (dz_head_chain.h): fold of fc1's slabs + bias + ReLU ->
//             noisy fc2 (adv2, val2) -> loss (dueling + softmax + double-Q selector + Cramer
//             projection + cross-entropy + dlogits + priorities) -> [fc2 input gradient as a
//             row-owning stream | fc2 weight gradient]; side blocks: Gram matrices of fc1's
//             input.  (dz_rainbow_args_t::separate_launches = 1, other shapes of the call:
//             epilogue -> fc2 -> head/loss -> fc2 backward as four launches)
//   backward: [Gram side blocks of dh1 | sum-tree priority write-back | fc1 input gradient on
//             the matrix pipe] -> [conv3 wgrad | conv3 dgrad] -> [conv2 wgrad | conv2 dgrad]
//             -> conv1 wgrad -> finalize (conv slabs, bias column sums, global-norm partials,
//             step count)
//   update  : Adam; forms fc1's mu and sigma weight gradients on the fly (dz_fc1_onfly.h)
//             and, in the loop over a static replay, carries the next step's sample+gather
// Other shapes of the call (split phases, batch > 32, keep_all_grads) store fc1's weight
// gradient: [fc1 wgrad | fc1 dgrad] -> reduce, and Adam streams the stored vector.
// Roofline notes per kernel are in DESIGN.md.
#include "dz_sumtree_dev.h"
#include "dz_torso.h"
#include "dz_fc1_onfly.h"
#include "dz_row_dgrad.h"
#include "dz_fc1_dgrad.h"
#include "dz_act_one.h"
#include "dz_head_chain.h"

namespace {

// Launch constants: the measured best on MI355X at B = 32 (the sweeps and the
// alternatives that lost are in EXPERIMENTS.md; the code that implemented them is gone).
constexpr int kFc1DgradSplits = 12;  // fc1 input-gradient k-splits of the STORED-gradient forms (split steps, batch > 32, keep_all_grads): 11 slabs of 6 single-chunk stages
constexpr int kFc2Splits = 8;        // fc2 forward k-splits
// The two noisy linear layers' input gradients in the one-call step (dz_fc1_onfly.h):
//   fc1: on the matrix pipe, weights transposed through LDS by LDS-DMA (dz_fc1_dgrad.h): 196
//        workgroups x 16 rows behind the sixteen Gram side blocks.  (Rounds 3-4 ran it as a
//        row-owning VALU stream, dz_row_dgrad.h: 12.3-13.9 us whatever the launch shape -- 224
//        ... 640 workgroups -- because its 170 vector instructions per row and wave, 64 of them
//        the multiply-adds, were the bound; 10.75 us now.)
//   fc2: row-owning stream (dz_row_dgrad.h), 128 workgroups x 4 rows in front of the fc2
//        weight-gradient contraction; dh1 leaves that launch finished (summed over n,
//        ReLU-masked) -- no slabs to fold.
constexpr int kDg2Blocks = 128;
// ... and as a role of the multi-role head launch (dz_head_chain.h), where the role's arithmetic sits
// on the launch's critical path behind the dlogit seam: 256 workgroups x 2 rows (same box, launch
// 27.5 -> 26.4 us; 64 x 8: 29.3)
#ifndef DZ_HC_D1_BLOCKS
#define DZ_HC_D1_BLOCKS 256
#endif
constexpr int kHcD1Blocks = DZ_HC_D1_BLOCKS;
// fc1's input gradient (dz_fc1_dgrad.h): one workgroup per 16 weight rows behind the side
// blocks (Gram norms of dh1; optionally the sum-tree priority write-back).
__global__ __launch_bounds__(256)
void fc1_dgrad_mfma_kernel(Fc1DgradMfma q, GramD gd, PrioUpdateParams prio) {
  __shared__ __attribute__((aligned(16))) float lds[kDmLdsFloats];
  static_assert(kDmLdsFloats >= 32 * GramDSide::kLd, "GramDSide's tile");
  static_assert(sizeof(lds) >= sizeof(WbScratch), "the write-back's LDS walk");
  unsigned bid = blockIdx.x;
  if (bid < (unsigned)GramDSide::kBlocks) { GramDSide::run(gd, bid, lds, (int)sizeof(lds)); return; }
  bid -= GramDSide::kBlocks;
  if (prio.node) {
    if (bid == 0) { PrioUpdateSideFast::run(prio, 0, lds, (int)sizeof(lds)); return; }
    bid -= 1;
  }
  fc1_dgrad_mfma_block(q, bid, lds);
}
template <int NJ0>   // 256-column chunks of the advantage head
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2)))
void fc2_bwd_rows_kernel(FcWgradParams w, dim3 gw, RowDgrad q) {
  constexpr int SM = DzGemmSmem<FcWg>::ELEMS > kRdLdsFloats ? DzGemmSmem<FcWg>::ELEMS : kRdLdsFloats;
  __shared__ __attribute__((aligned(16))) float smem[SM];
  if (blockIdx.x < (unsigned)q.nblocks) row_dgrad_block<NJ0, 1, false, 4>(q, blockIdx.x, smem);
  else dz_gemm_body<FcWg>(w, dz_unflatten(blockIdx.x - q.nblocks, gw), smem);
}
constexpr int kAdamBlocks = 2048;    // grid-stride Adam launch width (8 blocks per CU)
constexpr int kAdamBlocksSG = 1536;  // with ~550 sample+gather blocks in front: 6 per CU, so that the whole launch is co-resident

}  // namespace


// The network apply for G groups (ref: networks.py:224-253): conv torso, fused
// noisy fc1 (adv1|val1), noisy fc2 (adv2, val2) into ws_fc2_out rows [G*B].
struct FwdHeads { FcHead fc1h[2]; FcHead fc2h[2]; };
static int rainbow_forward(const dz_rainbow_layout_t& L, const FwdHeads& H, int G, int B,
                           const float* const* prm, const float* const* nz,
                           const uint8_t* const* in, float* ws, hipStream_t s,
                           const NoiseParams* resample = nullptr,
                           bool skip_fc2_epilogue = false, bool stop_after_fc1 = false,
                           const SeamClear* clr = nullptr) {
  int rc = DZ_OK;
  const int NA = L.num_actions * L.num_atoms;
  const int ld2 = L.adv2_ld + L.val2_ld;
  const FcHead* fc1h = H.fc1h;
  const FcHead* fc2h = H.fc2h;
  {
    const TorsoBufs T = {L.conv_w, L.conv_b, ws + L.ws_act1, ws + L.ws_act2, ws + L.ws_feat};
    long long* dbg = nullptr;
#ifdef DZ_GEMM_STAMPS
    if (G == 3) dbg = (long long*)(ws + L.ws_dfeat_part);
#endif
    rc = torso_forward(T, G, B, prm, in, s, resample, dbg, clr);
    if (rc) return rc;
  }
  {  // fc1: noisy adv1 | val1, split-K partials
    FcFwdParams p;
    p.x = ws + L.ws_feat; p.ldx = kFlat; p.M = B; p.G = G; p.NH = 2; p.S = kS_fc1;
    p.noisy = 1;
    for (int g = 0; g < G; ++g) { p.params[g] = prm[g]; p.noise[g] = nz[g]; }
    p.head[0] = fc1h[0]; p.head[1] = fc1h[1];
    p.part = ws + L.ws_fc1_part; p.ldo = 1024;
    p.S = kFc1Splits;
    if (B > 32) {
      // more than one batch tile: tile GEMM over the depth-2K form [x | x.eps_in] [Wmu ; Wsig.eps_out]
      rc = dz_launch_gemm<FcFwdOp<1, 2, 2, 4>>(p, dim3(8, (B + 31) / 32, G * 2 * kFc1Splits), s);
    } else {
      // one weight stream per parameter set, W_eff built in registers; workgroups in
      // XCD-aware order (+3 %)
      FcStreamFwd3Params q;
      q.x = p.x; q.ldx = p.ldx; q.M = B; q.noisy = 1; q.G = G;
      const int ns = dz_fc3_assign_sets(q, G, prm, nz);
      DZ_REQUIRE(ns > 0);
      q.head[0] = fc1h[0]; q.head[1] = fc1h[1];
      q.part = p.part; q.ldo = p.ldo;
      q.rows_per_split = ((kFlat + kFc1Splits - 1) / kFc1Splits + 3) & ~3;
      constexpr int kRps = ((kFlat + kFc1Splits - 1) / kFc1Splits + 3) & ~3;   // rows per split: 100 / 196
#ifndef DZ_FC1_CH
#define DZ_FC1_CH (kFc1Splits == 32 ? 5 : 7)
#endif
#ifndef DZ_FC1_DEPTH
#define DZ_FC1_DEPTH 3
#endif
      constexpr int kNl = (kRps / 2 + DZ_FC1_CH - 1) / DZ_FC1_CH * DZ_FC1_CH;   // k-pairs per lane
      q.xcd_order = (kFc1Splits * ns) % 8 == 0;
#ifndef DZ_FC1_DMA   // ring slots per wave of the LDS-DMA weight stream, 0 = the register pipeline
#define DZ_FC1_DMA 0
#endif
#if DZ_FC1_DMA
      constexpr int kNch = (kRps + 7) / 8;   // 8-row slots per split
      const size_t lds = ((size_t)4 * DZ_FC1_DMA * 512 + (size_t)q.rows_per_split * (2 * 32 + 2)) * sizeof(float);
      static bool once = [] {
        (void)hipFuncSetAttribute((const void*)dz_fc_stream_dma<1, kNch, DZ_FC1_DMA>,
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        return true;
      }();
      (void)once;
      hipLaunchKernelGGL((dz_fc_stream_dma<1, kNch, DZ_FC1_DMA>), dim3(8, kFc1Splits, ns), dim3(256), lds, s, q);
#else
      const size_t lds = (size_t)q.rows_per_split * (2 * 32 + 2) * sizeof(float);
      hipLaunchKernelGGL((dz_fc_stream_fwd3<1, kNl, DZ_FC1_CH, DZ_FC1_DEPTH>), dim3(8, kFc1Splits, ns), dim3(256), lds, s, q);
#endif
      DZ_LAUNCH_CHECK();
      rc = DZ_OK;
    }
    if (rc) return rc;
    DZ_PROF(s, "fc1_fwd");
    if (stop_after_fc1) return rc;  // the caller's kernel folds the slabs (actor tail)
    hipLaunchKernelGGL(fc_epilogue_kernel, dim3(16, G * B), dim3(256), 0, s,
                       ws + L.ws_fc1_part, kFc1Splits, G * B, 1024, 1024, B,
                       prm[0], prm[1], prm[2], (long)L.fc1_mu_b, (long)L.fc1_sig_b,
                       nz[0], nz[1], nz[2], (int)L.n_fc1_out, 1, ws + L.ws_h1);
    DZ_LAUNCH_CHECK();
    DZ_PROF(s, "fc1_epilogue");
  }
  {  // fc2: noisy adv2 (no mu bias) and val2 (no mu bias)
    FcFwdParams p;
    p.x = ws + L.ws_h1; p.ldx = 1024; p.M = B; p.G = G; p.NH = 2; p.S = kFc2Splits;
    p.noisy = 2;  // against W_eff (depth K): 11.2 vs 12.1 us for the two-GEMM form
    for (int g = 0; g < G; ++g) { p.params[g] = prm[g]; p.noise[g] = nz[g]; }
    p.head[0] = fc2h[0]; p.head[1] = fc2h[1];
    p.part = ws + L.ws_fc2_part; p.ldo = ld2;
    const dim3 g2((NA + FcFwd::BN - 1) / FcFwd::BN, (B + 31) / 32, G * 2 * kFc2Splits);
    rc = dz_launch_gemm<FcFwdOp<1, 2, 2, 4, 2>>(p, g2, s);
    if (rc) return rc;
    DZ_PROF(s, "fc2_fwd");
    if (skip_fc2_epilogue) return rc;  // the loss kernel folds the partial slabs itself
    hipLaunchKernelGGL(fc_epilogue_kernel, dim3((ld2 + 63) / 64, G * B), dim3(256),
                       0, s, ws + L.ws_fc2_part, kFc2Splits, G * B, ld2, ld2, B,
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
  DZ_REQUIRE(L && A > 0 && A <= 256 && K > 0 && K <= 64 && B > 0 && B <= 1024);
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
  L->ws_fc2_part = take((int64_t)kMaxS_fc2 * GB * ld2);
  L->ws_fc2_out = take(GB * ld2);
  L->ws_dout2 = take((int64_t)B * ld2);
  L->ws_dh1 = take((int64_t)B * 1024);
  L->ws_dfeat_part = take((int64_t)kMaxS_dfeat * B * kFlat);
  L->ws_dfeat = take((int64_t)B * kFlat);
  L->ws_dact2 = take((int64_t)B * 81 * 64);
  L->ws_dact1 = take((int64_t)B * 400 * 32);
  const int64_t wp = (int64_t)kS_cw1 * Conv1Wg::KROWS * 32 +
                     (int64_t)kS_cw2 * Conv2Wg::KROWS * 64 +
                     (int64_t)kS_cw3 * Conv3Wg::KROWS * 64;  // one slab per conv
  L->ws_wgrad_part = take(wp);
  L->ws_norm_part = take(kNormFinal + kNormSlots);
  L->ws_colsum_part = take(4);
  L->ws_scalars = take(16);
  L->ws_q_sel = take((int64_t)B * A);
  L->ws_target_probs = take((int64_t)B * K);
  L->ws_act_seams = take(kActSeamWords);
  L->ws_count = w;
  return DZ_OK;
}

// ---- the step ---------------------------------------------------------------
extern "C" int dz_rainbow_learn(const dz_rainbow_args_t* a, int phases,
                                dz_stream_t stream) {
  DZ_REQUIRE(a && a->online && a->target && a->ws && a->noise && a->support);
  DZ_REQUIRE(a->s_tm1 && a->s_t && a->a_tm1 && a->r_t && a->discount_t && a->weights);
  DZ_REQUIRE(a->losses && a->priorities);
  if (a->next_sample) {   // checked before anything is enqueued
    DZ_REQUIRE((phases & DZ_PHASE_BACKWARD) && (phases & DZ_PHASE_OPTIMIZER));
    // a prioritized sample must follow THIS step's write-back into the same tree
    // (rainbow/agent.py:181-198: update_priorities, then the next sample)
    DZ_REQUIRE(!a->next_sample->args.node || a->prio_node == a->next_sample->args.node);
  }
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
  const int Gf = kG;
  const bool fuse = (size_t)3 * ld2 * sizeof(float) <= 48 * 1024;
  const bool do_nets = (phases & (DZ_PHASE_FORWARD | DZ_PHASE_FWD_NETS)) != 0;
  const bool do_loss = (phases & (DZ_PHASE_FORWARD | DZ_PHASE_FWD_LOSS)) != 0;
  // the step's noise: blocks [0, Gf) of 3 at stream position *adam_count * 3 * stride
  NoiseParams nq = {const_cast<float*>(a->noise), Gf * (long)L.noise_stride, a->noise_seed,
                    (uint64_t)0x5eed, a->adam_count, 3 * (long)L.noise_stride, 0};
  if (do_nets && a->resample_noise) DZ_REQUIRE(a->adam_count);
  // fc1's weight gradient formed inside the optimiser (dz_fc1_onfly.h): when this ONE call
  // runs the loss, the backward pass and the optimiser, and nobody asked to keep the
  // gradient vector complete
  const bool onfly = do_loss && (phases & DZ_PHASE_BACKWARD) && (phases & DZ_PHASE_OPTIMIZER) &&
                     !a->keep_all_grads && B <= 32;
  // (Gram partials of the layer input: the head of the conv weight-gradient slab buffer,
  // idle between the previous step's finalize and this step's conv backward launches)
  double* gram_part = (double*)(ws + ((L.ws_wgrad_part + 7) & ~(int64_t)7));
  // The head chain -- fc1 epilogue, noisy fc2, loss, fc2 backward -- as ONE multi-role launch
  // (dz_head_chain.h) whenever this one call runs the whole step on the on-the-fly path; its seam
  // buffers (h1, four fc2 slabs, dlogits) are cleared by side blocks of the conv1 forward launch.
  const int nj0_rd = (NA + 255) / 256;
  // (a step whose head launch gave up on a seam is VOID: its finalize, optimiser and priority
  // write-back launches read the sticky word and change nothing -- ADVICE r5)
  const unsigned* chain_abort = reinterpret_cast<const unsigned*>(ws + L.ws_scalars + DZ_SC_CHAIN_FAIL);
  const bool chain = onfly && do_nets && !a->separate_launches && Gf == 3 &&
                     3 * ld2 <= kHcLdsFloats && nj0_rd >= 1 && nj0_rd <= 4 && K <= 256 &&
                     ((kHid + kHcD1Blocks - 1) / kHcD1Blocks) * (nj0_rd + 1) <= 32 &&
                     ((kHid + kHcD1Blocks - 1) / kHcD1Blocks) * 32 * 2 <= 512;
  if (do_nets) {
    const uint8_t* in[kG] = {a->s_tm1, a->s_t, a->s_t};
    SeamClear clr;
    if (chain) {
      clr.ptr[0] = ws + L.ws_h1; clr.n4[0] = (int)(((int64_t)Gf * B * 1024) >> 2);
      clr.ptr[1] = ws + L.ws_fc2_part; clr.n4[1] = (int)(((int64_t)kHcStages * Gf * B * ld2) >> 2);
      clr.ptr[2] = ws + L.ws_dout2; clr.n4[2] = (int)(((int64_t)B * ld2) >> 2);
      clr.blocks = (unsigned)((clr.n4[0] + clr.n4[1] + clr.n4[2] + 255) / 256);
    }
    rc = rainbow_forward(L, H, Gf, B, prm, nz, in, ws, s,
                         a->resample_noise ? &nq : nullptr, fuse, chain, chain ? &clr : nullptr);
    if (rc) return rc;
  }
  // the fc2 layer's gradient slots of the fused global norm (both forms of the head chain)
  const int fc2_slots_hc = ((NA + FcWg::BN - 1) / FcWg::BN) * (kHid / FcWg::BM) * 2 * 4;
  if (chain) {
    HeadChain q = {};
    q.fc1_part = ws + L.ws_fc1_part; q.rows = Gf * B; q.B = B; q.G = Gf;
    for (int g = 0; g < kG; ++g) { q.prm[g] = prm[g]; q.nz[g] = nz[g]; }
    q.fc1_mu_b = L.fc1_mu_b; q.fc1_sig_b = L.fc1_sig_b; q.n_fc1_out = (int)L.n_fc1_out;
    q.h1 = ws + L.ws_h1;
    q.fc2h[0] = fc2h[0]; q.fc2h[1] = fc2h[1];
    q.tiles0 = (NA + 63) / 64; q.tiles1 = (K + 63) / 64;
    q.fc2_part = ws + L.ws_fc2_part; q.ld2 = ld2;
    HeadPre pre = {};
    pre.part = ws + L.ws_fc2_part; pre.S = kHcStages; pre.rows = Gf * B;
    for (int g = 0; g < kG; ++g) { pre.prm[g] = prm[g]; pre.nz[g] = nz[g]; }
    pre.b_sig = L.fc2_sig_b; pre.eps_out = (int)L.n_fc2_out;
    q.loss = {ws + L.ws_fc2_out, ld2, NAp, B, A, K, 1, 1, 2, a->a_tm1, a->r_t, a->discount_t,
              a->weights, a->support, ws + L.ws_dout2, a->losses, a->priorities,
              ws + L.ws_q_sel, ws + L.ws_target_probs, pre};
    q.fail = reinterpret_cast<unsigned*>(ws + L.ws_scalars + DZ_SC_CHAIN_FAIL);
    q.status = a->prio_node ? a->prio_status : nullptr;
    q.limit = g_dz_act_spin_limit;
    // role D: the arguments of fc2_bwd_rows_kernel
    RowDgrad& r = q.rd;
    r.params = a->online; r.noise = nz[0]; r.head[0] = fc2h[0]; r.head[1] = fc2h[1];
    r.dy = ws + L.ws_dout2; r.ldy = ld2; r.mask = ws + L.ws_h1; r.out = ws + L.ws_dh1;
    r.ldo = 1024; r.out_col[0] = 0; r.out_col[1] = 512; r.same_out = 0;
    r.M = B; r.K = kHid; r.nblocks = kHcD1Blocks;
    r.fail = q.fail; r.limit = q.limit; r.poison = a->losses; r.watch_col = NAp + K - 1;
    FcWgradParams& w = q.wg;
    w.x = ws + L.ws_h1; w.ldx = 1024; w.dy = ws + L.ws_dout2; w.ldy = ld2; w.M = B;
    w.NH = 2; w.noisy = 1; w.noise = nz[0]; w.head[0] = fc2h[0]; w.head[1] = fc2h[1];
    w.grad = a->grad;
    w.sumsq = ws + L.ws_norm_part + kNormFinal;
    w.sq_nx = (NA + FcWg::BN - 1) / FcWg::BN; w.sq_ny = kHid / FcWg::BM;
    DZ_REQUIRE(fc2_slots_hc <= kNormSlots);
    q.gw = dim3((NA + FcWg::BN - 1) / FcWg::BN, kHid / FcWg::BM, 2);
    q.gram.x = ws + L.ws_feat; q.gram.M = B; q.gram.K = kFlat;
    q.gram.eps_in[0] = nz[0] + L.n_adv1_in; q.gram.eps_in[1] = nz[0] + L.n_val1_in;
    q.gram.part = gram_part;
    q.nA = (unsigned)(Gf * B * (1024 / kHcFoldCols));
    q.nB = (unsigned)(kHcStages * Gf * (q.tiles0 + q.tiles1));
    q.nC = (unsigned)B; q.nD1 = (unsigned)kHcD1Blocks; q.nD2 = dz_count(q.gw);
#ifdef DZ_HC_STAMPS
    q.dbg = reinterpret_cast<long long*>(ws + L.ws_dfeat_part); q.rd.dbg = q.dbg;
#endif
    const dim3 grid(q.nA + q.nB + q.nC + q.nD1 + q.nD2 + (unsigned)kGramXBlocks);
    switch (nj0_rd) {
      case 1: hipLaunchKernelGGL(rainbow_head_chain_kernel<1>, grid, dim3(256), 0, s, q); break;
      case 2: hipLaunchKernelGGL(rainbow_head_chain_kernel<2>, grid, dim3(256), 0, s, q); break;
      case 3: hipLaunchKernelGGL(rainbow_head_chain_kernel<3>, grid, dim3(256), 0, s, q); break;
      default: hipLaunchKernelGGL(rainbow_head_chain_kernel<4>, grid, dim3(256), 0, s, q); break;
    }
    DZ_LAUNCH_CHECK();
    DZ_PROF(s, "head_chain");
  }
  if (do_loss && !chain) {
    {
      GramX gx;
      if (onfly) {
        gx.x = ws + L.ws_feat; gx.M = B; gx.K = kFlat;
        gx.eps_in[0] = nz[0] + L.n_adv1_in; gx.eps_in[1] = nz[0] + L.n_val1_in;
        gx.part = gram_part;
      }
      const unsigned hb = (unsigned)B + (onfly ? (unsigned)kGramXBlocks : 0u);
      HeadPre pre = {};
      if (fuse) {
        pre.part = ws + L.ws_fc2_part; pre.S = kFc2Splits; pre.rows = Gf * B;
        for (int g = 0; g < kG; ++g) { pre.prm[g] = prm[g]; pre.nz[g] = nz[g]; }
        pre.b_sig = L.fc2_sig_b; pre.eps_out = (int)L.n_fc2_out;
        hipLaunchKernelGGL(rainbow_head_loss_kernel<1>, dim3(hb), dim3(256),
                           (size_t)3 * ld2 * sizeof(float), s, ws + L.ws_fc2_out, ld2, NAp, B,
                           A, K, 1, 1, 2, a->a_tm1, a->r_t, a->discount_t, a->weights,
                           a->support, ws + L.ws_dout2, a->losses, a->priorities,
                           ws + L.ws_q_sel, ws + L.ws_target_probs, pre, gx);
      } else {
        hipLaunchKernelGGL(rainbow_head_loss_kernel<0>, dim3(hb), dim3(256), 0, s,
                           ws + L.ws_fc2_out, ld2, NAp, B, A, K, 1, 1, 2, a->a_tm1, a->r_t,
                           a->discount_t, a->weights, a->support, ws + L.ws_dout2, a->losses,
                           a->priorities, ws + L.ws_q_sel, ws + L.ws_target_probs, pre, gx);
      }
    }
    DZ_LAUNCH_CHECK();
    DZ_PROF(s, "head_loss");
  }

  int n_final = 0;  // fused-norm partials left by this call's backward phase
  // fc1 sigma-weight gradient: derived inside Adam instead of stored and re-read,
  // when this one call both produces and consumes it
  const bool derive_sig = (phases & DZ_PHASE_BACKWARD) && (phases & DZ_PHASE_OPTIMIZER) &&
                          !a->keep_all_grads && !onfly;
  auto fc1_wgrad_params = [&](FcWgradParams& w) {
    w.x = ws + L.ws_feat; w.ldx = kFlat; w.dy = nullptr; w.ldy = 1024; w.M = B;
    w.NH = 2; w.noisy = 1; w.noise = nz[0]; w.head[0] = fc1h[0]; w.head[1] = fc1h[1];
    w.grad = a->grad;
    // dh1 = relu'(h1) * sum of the fc2 input-gradient slabs, formed in the loaders;
    // the first row tiles also write it out for the bias column sums (finalize)
    w.dyp.part = ws + L.ws_fc1_part; w.dyp.stride = (long)B * 1024; w.dyp.mask = ws + L.ws_h1;
    w.dyp.out = ws + L.ws_dh1;
  };
  // optional priority write-back (dz_rainbow_args_t::prio_*), carried by one of this
  // call's launches as an extra block
  bool prio_pending = a->prio_node != nullptr;
  ConvWgDeferred wg_defer;   // conv3's / conv2's weight gradients travel with conv1's (dz_torso.h)
  PrioUpdateParams prio_q = {};
  if (prio_pending) {
    DZ_REQUIRE(a->prio_ids && a->prio_status && dz_is_pow2(a->prio_cap_pow2) &&
               a->prio_capacity > 0 && a->prio_capacity <= a->prio_cap_pow2 &&
               a->prio_exponent >= 0.0 && B <= 256);
    prio_q = {a->prio_node, a->prio_cap_pow2, a->prio_capacity, 0, 0, a->prio_ids,
              a->priorities, 1, a->prio_exponent, B, a->prio_max_seen, a->prio_status, 0};
    prio_q.abort = chain_abort;
  }
  if (phases & DZ_PHASE_BACKWARD) {
    DZ_REQUIRE(a->grad);
    float* grad = a->grad;
    // Every layer's weight gradient and input gradient are independent, so each
    // pair is ONE launch (dz_mfma_gemm2/3: horizontal fusion); the conv partial
    // reductions and the bias column sums are one launch at the end.
    // fused global norm: the weight-gradient kernels leave per-wave sums of
    // squares in sq_slots (fc2 first, then fc1), finalize_grads folds them and
    // adds its own -> ws_norm_part[0..n_final) is the partial list for Adam.
    float* sq_final = ws + L.ws_norm_part;
    float* sq_slots = sq_final + kNormFinal;
    const int fc2_slots = ((NA + FcWg::BN - 1) / FcWg::BN) * (kHid / FcWg::BM) * 2 * 4;
    const int fc1_slots = (512 / FcWg::BN) * (kFlat / FcWg::BM) * 2 * 4;
    DZ_REQUIRE(fc2_slots + fc1_slots <= kNormSlots);
    // optional priority write-back (dz_rainbow_args_t::prio_*), carried by the conv3
    // backward launch
    float* part1 = ws + L.ws_wgrad_part;
    float* part2 = part1 + (long)kS_cw1 * Conv1Wg::KROWS * 32;
    float* part3 = part2 + (long)kS_cw2 * Conv2Wg::KROWS * 64;
    if (!chain) {  // fc2: weight gradients (both heads) + input gradient of each head
      FcWgradParams w;
      w.x = ws + L.ws_h1; w.ldx = 1024; w.dy = ws + L.ws_dout2; w.ldy = ld2; w.M = B;
      w.NH = 2; w.noisy = 1; w.noise = nz[0]; w.head[0] = fc2h[0]; w.head[1] = fc2h[1];
      w.grad = grad;
      w.sumsq = sq_slots; w.sq_nx = (NA + FcWg::BN - 1) / FcWg::BN; w.sq_ny = kHid / FcWg::BM;
      FcDgradParams d[2];
      constexpr int s_dh1 = kS_dh1;
      for (int h = 0; h < 2; ++h) {
        d[h].dy = ws + L.ws_dout2; d[h].ldy = ld2; d[h].M = B; d[h].NH = 1;
        // (the W_eff form of this small input gradient measured slower: 16.4 vs 14.1 us)
        d[h].S = s_dh1; d[h].noisy = 1; d[h].params = a->online; d[h].noise = nz[0];
        d[h].head[0] = fc2h[h]; d[h].head[1] = fc2h[h];
        d[h].ldo = 1024; d[h].K = kHid; d[h].x_off = 512 * h;
        // partial slabs of dh1: into the (now idle) fc1 forward slab buffer; they are
        // summed and ReLU-masked by the loaders of the fc1 backward launch (DyParts)
        d[h].part = ws + L.ws_fc1_part;
      }
      const dim3 gd(kHid / FcDg::BN, (B + 31) / 32, s_dh1);
      typedef FcDgradOp<1, 2, 2, 4, 1, 1, 1> FcDg1;  // noisy == 1 at compile time
      if (onfly) {
        // the input gradient as a row-owning stream over both heads' [512][N] matrices:
        // dh1 = relu'(h1) * (dout2_adv . Wadv^T | dout2_val . Wval^T), finished
        RowDgrad q = {};
        q.params = a->online; q.noise = nz[0]; q.head[0] = fc2h[0]; q.head[1] = fc2h[1];
        q.dy = ws + L.ws_dout2; q.ldy = ld2; q.mask = ws + L.ws_h1; q.out = ws + L.ws_dh1;
        q.ldo = 1024; q.out_col[0] = 0; q.out_col[1] = 512; q.same_out = 0;
        q.M = B; q.K = kHid; q.nblocks = kDg2Blocks;
        const int nj0 = (NA + 255) / 256;
        // (LDS: rows x jobs x 256 floats <= 32 KB; epilogue: two outputs per thread)
        DZ_REQUIRE(nj0 >= 1 && nj0 <= 4 && K <= 256 && row_dgrad_max_rows(q) * (nj0 + 1) <= 32 &&
                   row_dgrad_max_rows(q) * 32 * 2 <= 512);
        const dim3 gw((NA + FcWg::BN - 1) / FcWg::BN, kHid / FcWg::BM, 2);
        const dim3 grid(kDg2Blocks + dz_count(gw));
        switch (nj0) {
          case 1: hipLaunchKernelGGL(fc2_bwd_rows_kernel<1>, grid, dim3(256), 0, s, w, gw, q); break;
          case 2: hipLaunchKernelGGL(fc2_bwd_rows_kernel<2>, grid, dim3(256), 0, s, w, gw, q); break;
          case 3: hipLaunchKernelGGL(fc2_bwd_rows_kernel<3>, grid, dim3(256), 0, s, w, gw, q); break;
          default: hipLaunchKernelGGL(fc2_bwd_rows_kernel<4>, grid, dim3(256), 0, s, w, gw, q); break;
        }
        DZ_LAUNCH_CHECK();
        rc = DZ_OK;
      } else
      rc = dz_launch_gemm3<FcWg, FcDg1, FcDg1>(
          w, dim3((NA + FcWg::BN - 1) / FcWg::BN, kHid / FcWg::BM, 2), d[0], gd, d[1], gd, s);
      if (rc) return rc;
      DZ_PROF(s, "fc2_wgrad+dgrad");
    }
    {  // fc1: weight gradients + input gradient (adv1 + val1 paths) -> dfeat
      FcWgradParams w;
      fc1_wgrad_params(w);
      w.sumsq = sq_slots + fc2_slots; w.sq_nx = 512 / FcWg::BN; w.sq_ny = kFlat / FcWg::BM;
      w.skip_sig_store = derive_sig;
      // input gradient against W_eff (built in the B loader: depth N instead of the
      // two-GEMM form's 2N, 14.2 vs 17.2 us), single-chunk stages
      FcDgradParams d;
      d.dy = nullptr; d.ldy = 1024; d.M = B; d.NH = 2; d.S = kFc1DgradSplits; d.noisy = 2;
      d.dyp = w.dyp; d.dyp.out = nullptr;
      d.params = a->online; d.noise = nz[0]; d.head[0] = fc1h[0]; d.head[1] = fc1h[1];
      d.part = ws + L.ws_dfeat_part; d.ldo = kFlat; d.K = kFlat; d.x_off = 0;
      // ONE launch for the 25.7 MB weight read and the 12.9 MB gradient write, the
      // weight-gradient blocks first (15 us vs 14 + 14 back to back)
      // (compiled for 5 waves per SIMD: 1323 workgroups then find 1280 co-resident slots instead of 1024)
      if (onfly) {
        // no weight-gradient workgroups: sixteen side blocks leave the layer's squared
        // gradient norm in the fc1 slots (GramDSide); the input gradient runs on the matrix
        // pipe with the weights transposed through LDS (dz_fc1_dgrad.h: no slabs, no reduce
        // launch)
        GramD gdp;
        gdp.dh1 = ws + L.ws_dh1; gdp.M = B; gdp.eps_out = nz[0] + L.n_fc1_out;
        gdp.gx_part = gram_part; gdp.dot_out = sq_slots + fc2_slots;
        // the sum-tree priority write-back rides HERE in the one-call step (same-box A/B:
        // +0.8 us on this launch, +1.3 us on conv3's backward launch)
        // Round 6: the block's chain of three dependent trips to memory (ids -> path siblings ->
        // running maximum; then the walk) is ~9.5 us, longer than any launch of the backward chain: it
        // set THIS launch's duration (12.0 us with it, 9.9 without; conv3's 8.1 -> 12.1, the weight
        // gradients' 10.2 -> 14.4).  DZ_PRIO_HOST 3 cuts it in two (PrioUpdateParams::phase): checks,
        // maximum and the siblings' trip here, the walk and the stores in conv3's backward launch, the
        // hand-over in an idle workspace region.
#ifndef DZ_PRIO_HOST   // 0: whole block here, 1: in conv3's backward, 2: with the weight gradients, 3: split here + conv3's
#define DZ_PRIO_HOST 3
#endif
        const bool split_prio = DZ_PRIO_HOST == 3 && prio_pending && B <= 64 &&   // (one wave walks the batch)
                                a->prio_cap_pow2 <= ((int64_t)1 << 31) &&
                                (int64_t)kPrioScratchDoubles * 2 <= (int64_t)kMaxS_dfeat * B * kFlat;
        const bool carry_prio = prio_pending && (DZ_PRIO_HOST == 0 || split_prio);
        if (split_prio) {
          prio_q.phase = 1;
          // (the END of the idle split-K slab region of the stored-gradient forms)
          prio_q.scratch = reinterpret_cast<double*>(ws + L.ws_dfeat_part + (int64_t)kMaxS_dfeat * B * kFlat) - kPrioScratchDoubles;
        }
        Fc1DgradMfma m = {};
        m.params = a->online; m.noise = nz[0];
        m.w_mu = L.fc1_mu_w; m.w_sig = L.fc1_sig_w; m.ldw = L.fc1_ld;
        m.eps_in[0] = (int)L.n_adv1_in; m.eps_in[1] = (int)L.n_val1_in; m.eps_out = (int)L.n_fc1_out;
        m.dy = ws + L.ws_dh1; m.ldy = 1024; m.mask = ws + L.ws_feat; m.out = ws + L.ws_dfeat;
        m.ldo = kFlat; m.M = B; m.K = kFlat;
        static_assert(kFlat % 16 == 0, "16 weight rows per workgroup");
        hipLaunchKernelGGL(fc1_dgrad_mfma_kernel,
                           dim3(GramDSide::kBlocks + kFlat / 16 + (carry_prio ? 1 : 0)), dim3(256),
                           0, s, m, gdp, carry_prio ? prio_q : PrioUpdateParams{});
        if (carry_prio && !split_prio) prio_pending = false;
        if (split_prio) prio_q.phase = 2;   // (the second half stays pending: conv3's backward launch)
        DZ_LAUNCH_CHECK();
        rc = DZ_OK;
      } else {
        rc = dz_launch_gemm2_occ<FcWgradOp<2, 2, 1, 2, kS_dh1>, FcDgradOp<1, 2, 2, 1, 1, 1, 2, kS_dh1>, 5>(
            w, dim3(512 / FcWg::BN, kFlat / FcWg::BM, 2), d,
            dim3(kFlat / 64, (B + 31) / 32, kFc1DgradSplits), s);
      }
      if (rc) return rc;
      DZ_PROF(s, "fc1_dgrad+wgrad");
      if (!onfly) {
        hipLaunchKernelGGL(reduce_parts_kernel, dim3((B * kFlat + 63) / 64), dim3(256), 0,
                           s, ws + L.ws_dfeat_part, kFc1DgradSplits, (long)B * kFlat,
                           ws + L.ws_feat, ws + L.ws_dfeat);
        DZ_LAUNCH_CHECK();
        DZ_PROF(s, "dfeat_reduce");
      }
    }
    {  // conv3: weight+bias gradient partials + input gradient (relu(conv2) mask)
      ConvWgradParams w;
      w.in = ws + L.ws_act2; w.dy = ws + L.ws_dfeat; w.part = part3; w.B = B; w.S = kS_cw3;
      ConvDgradParams d;
      d.dy = ws + L.ws_dfeat; d.w = a->online + L.conv_w[2]; d.act = ws + L.ws_act2;
      d.dx = ws + L.ws_dact2; d.B = B;
      // (with the next step's sample riding in the optimiser launch the write-back must
      // be complete BEFORE that launch: it stays here)
      const bool prio_in_adam = (phases & DZ_PHASE_OPTIMIZER) != 0 && !a->next_sample && !onfly;
      if (prio_pending && !prio_in_adam && DZ_PRIO_HOST != 2) {
        // The sum-tree priority write-back rides in this launch as one extra block:
        // it needs only the loss kernel's priorities and nothing here reads the tree.
        // (Measured hosts: this launch hides it completely; inside the HBM-heavy fc1
        // launch its dependent loads stretch to 18 us, inside conv1's 8.5 us launch
        // it sticks out by 4 us.)
        rc = launch_conv3_bwd(w, d, B, s, &prio_q);
        prio_pending = false;
      } else {
        rc = launch_conv3_bwd(w, d, B, s, nullptr, &wg_defer);
      }
      if (rc) return rc;
      DZ_PROF(s, wg_defer.on3 ? "conv3_dgrad" : "conv3_wgrad+dgrad");
    }
    {  // conv2
      ConvWgradParams w;
      w.in = ws + L.ws_act1; w.dy = ws + L.ws_dact2; w.part = part2; w.B = B; w.S = kS_cw2;
      ConvDgradParams d;
      d.dy = ws + L.ws_dact2; d.w = a->online + L.conv_w[1]; d.act = ws + L.ws_act1;
      d.dx = ws + L.ws_dact1; d.B = B;
      rc = launch_conv2_bwd(w, d, B, s, &wg_defer);
      if (rc) return rc;
      DZ_PROF(s, wg_defer.on2 ? "conv2_dgrad" : "conv2_wgrad+dgrad");
    }
    {  // conv1 weight+bias gradient partials straight from the uint8 states
      ConvWgradParams p;
      p.in = a->s_tm1; p.dy = ws + L.ws_dact1; p.part = part1; p.B = B; p.S = kS_cw1;
      if (prio_pending && DZ_PRIO_HOST == 2) {
        rc = launch_conv1_wgrad(p, s, &wg_defer, &prio_q);
        prio_pending = false;
      } else {
        rc = launch_conv1_wgrad(p, s, &wg_defer);
      }
      if (rc) return rc;
      DZ_PROF(s, wg_defer.on3 ? "conv_wgrads3" : wg_defer.on2 ? "conv_wgrads" : "conv1_wgrad");
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
      const unsigned presum =
          (unsigned)((fc2_slots + (onfly ? GramDSide::kBlocks : fc1_slots) + 1023) / 1024);
      n_final = (int)(acc + J.c_tiles[0] + J.c_tiles[1] + presum);
      DZ_REQUIRE(n_final <= kNormFinal);
      J.sumsq = sq_final; J.presum_src = sq_slots;
      J.presum_n = fc2_slots + (onfly ? GramDSide::kBlocks : fc1_slots);
      J.bump_count = (phases & DZ_PHASE_OPTIMIZER) ? a->adam_count : nullptr;
      J.abort = chain_abort;
      hipLaunchKernelGGL(finalize_grads_kernel, dim3((unsigned)n_final), dim3(256), 0, s, J);
      DZ_LAUNCH_CHECK();
      DZ_PROF(s, "finalize_grads");
    }
  }

  if (phases & DZ_PHASE_OPTIMIZER) {
    DZ_REQUIRE(a->grad && a->adam_m && a->adam_v && a->adam_count);
    float* sc = ws + L.ws_scalars;
    int nparts = n_final;
    if (!(phases & DZ_PHASE_BACKWARD)) {  // optimiser alone: norm from the stored gradient
      hipLaunchKernelGGL(sumsq_kernel, dim3(kNormBlocks), dim3(256), 0, s, a->grad,
                         (long)L.param_count, ws + L.ws_norm_part, a->adam_count);
      DZ_LAUNCH_CHECK();
      DZ_PROF(s, "grad_sumsq");
      nparts = kNormBlocks;
    }
    DerivedGrad dg = {};
    if (derive_sig) {
      dg.dst_off = L.fc1_sig_w; dg.src_off = L.fc1_mu_w; dg.rows = kFlat; dg.ld = L.fc1_ld;
      dg.split_col = 512;
      dg.eps_in0 = nz[0] + L.n_adv1_in; dg.eps_in1 = nz[0] + L.n_val1_in;
      dg.eps_out = nz[0] + L.n_fc1_out;
      dg.on = 1;
    }
    // sample(k+1) + gather(k+1) as the first blocks of this launch (write-back(k) was
    // carried by the conv3 backward launch above)
    SampleGatherParams q = {};
    unsigned sgb = 0;
    if (a->next_sample) {
      DZ_REQUIRE((phases & DZ_PHASE_BACKWARD) && !prio_pending);
      // (chunk caps 1..64 and 1024..2048 optimiser blocks all measure within +-1 %)
      rc = sample_gather_from_desc(a->next_sample, q, &sgb);
      if (rc) return rc;
    }
    if (onfly) {
      Fc1OnFly of;
      of.feat = ws + L.ws_feat; of.dh1 = ws + L.ws_dh1; of.B = B;
      of.mu_b = (unsigned)(L.fc1_mu_w * 4); of.sig_b = (unsigned)(L.fc1_sig_w * 4);
      of.ld = L.fc1_ld;
      of.eps_in0 = nz[0] + L.n_adv1_in; of.eps_in1 = nz[0] + L.n_val1_in;
      of.eps_out = nz[0] + L.n_fc1_out;
      // the stored gradient covers everything but the two fc1 matrices
      const long mat = (long)kFlat * L.fc1_ld;
      DZ_REQUIRE(L.fc1_mu_w < L.fc1_sig_w && (L.fc1_sig_w + mat) * 4 < ((int64_t)1 << 32));
      AdamRanges rg;
      rg.lo[0] = 0; rg.n[0] = L.fc1_mu_w >> 2;
      rg.lo[1] = (L.fc1_mu_w + mat) >> 2; rg.n[1] = (L.fc1_sig_w - (L.fc1_mu_w + mat)) >> 2;
      rg.lo[2] = (L.fc1_sig_w + mat) >> 2; rg.n[2] = (L.param_count - (L.fc1_sig_w + mat)) >> 2;
      DZ_REQUIRE(!prio_pending);   // (carried by the conv3 backward launch)
      DZ_REQUIRE(nparts >= 1 && nparts <= kAdamPartRounds * 8 * 256);   // (adam_partials_request's range)
      hipLaunchKernelGGL(adam_onfly_kernel, dim3(adam_onfly_blocks(sgb)), dim3(256), 0, s,
                         a->online, a->grad, a->adam_m, a->adam_v, ws + L.ws_norm_part, nparts,
                         a->adam_count, a->losses, a->weights, B, sc, a->lr, a->b1, a->b2, a->eps,
                         a->max_norm, of, rg, q, sgb, chain_abort);
      DZ_LAUNCH_CHECK();
      DZ_PROF(s, a->next_sample ? "adam+next_sample" : "adam");
#ifdef DZ_ADAM_REPEAT   // (timing probe, variant builds only: the same launch again, back to back,
                        //  without the next sample's blocks -- wrong parameters, right durations)
      for (int rep = 0; rep < DZ_ADAM_REPEAT; ++rep)
        hipLaunchKernelGGL(adam_onfly_kernel, dim3(adam_onfly_blocks(0)), dim3(256), 0, s,
                           a->online, a->grad, a->adam_m, a->adam_v, ws + L.ws_norm_part, nparts,
                           a->adam_count, a->losses, a->weights, B, sc, a->lr, a->b1, a->b2, a->eps,
                           a->max_norm, of, rg, SampleGatherParams{}, 0u, chain_abort);
#endif
    } else if (a->next_sample) {
      hipLaunchKernelGGL(adam_sg_kernel, dim3(sgb + (unsigned)kAdamBlocksSG), dim3(256), 0, s,
                         a->online, a->grad, a->adam_m, a->adam_v, (long)(L.param_count >> 2),
                         ws + L.ws_norm_part, nparts, a->adam_count, a->losses, a->weights, B, sc,
                         a->lr, a->b1, a->b2, a->eps, a->max_norm, dg, q, sgb, chain_abort);
      DZ_LAUNCH_CHECK();
      DZ_PROF(s, "adam+next_sample");
    } else {
      // with the write-back as block 0 the grid stays g_adam_blocks wide (all blocks
      // co-resident at 8 per CU): one optimiser block fewer
      hipLaunchKernelGGL(adam_kernel, dim3((unsigned)kAdamBlocks), dim3(256), 0, s, a->online,
                         a->grad, a->adam_m, a->adam_v, (long)(L.param_count >> 2),
                         ws + L.ws_norm_part, nparts, a->adam_count, a->losses, a->weights, B, sc,
                         a->lr, a->b1, a->b2, a->eps, a->max_norm, dg,
                         prio_pending ? prio_q : PrioUpdateParams{}, chain_abort);
      DZ_LAUNCH_CHECK();
      DZ_PROF(s, "adam");
    }
    prio_pending = false;
  }
  if (prio_pending) {  // no launch of this call carried it
    hipLaunchKernelGGL(prio_update_side_kernel, dim3(1), dim3(256), 0, s, prio_q);
    DZ_LAUNCH_CHECK();
    DZ_PROF(s, "prio_update");
  }
  return DZ_OK;
}

extern "C" int dz_rainbow_graph_capture(const dz_rainbow_args_t* args, int phases,
                                        dz_stream_t stream, void** graph_exec_out) {
  DZ_REQUIRE(args && graph_exec_out && stream && !g_dz_prof_on);
  DZ_REQUIRE(!args->next_sample);  // its draws are by-value kernel arguments
  int rc;
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

extern "C" int dz_graph_capture_begin(dz_stream_t stream) {
  DZ_REQUIRE(stream && !g_dz_prof_on);
  DZ_HIP_CHECK(hipStreamBeginCapture(dz_s(stream), hipStreamCaptureModeThreadLocal));
  return DZ_OK;
}

extern "C" int dz_graph_capture_end(dz_stream_t stream, int discard, void** graph_exec_out) {
  DZ_REQUIRE(stream && graph_exec_out);
  *graph_exec_out = nullptr;
  hipGraph_t graph = nullptr;
  hipError_t e = hipStreamEndCapture(dz_s(stream), &graph);
  if (discard) {
    if (graph) (void)hipGraphDestroy(graph);
    (void)hipGetLastError();
    return DZ_OK;
  }
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
  hipLaunchKernelGGL(rainbow_q_values_kernel<0>, dim3(batch), dim3(64), 0, s,
                     ws + L.ws_fc2_out, L.adv2_ld + L.val2_ld, L.adv2_ld, num_actions,
                     num_atoms, support, q_values_out, greedy_out, vmax_out, HeadPre{});
  DZ_LAUNCH_CHECK();
  return DZ_OK;
}

// The actor's apply (ref: rainbow/agent.py:125-131, 171-179): fresh noise from
// (noise_seed, noise_counter) drawn as a side job of the conv1 launch, the fc2
// split-K fold inside the q-value kernel: 8 launches instead of 10 at batch 1.
extern "C" int dz_rainbow_act(int num_actions, int num_atoms, int batch, const float* params,
                              const uint8_t* states, float* noise, uint64_t noise_seed,
                              uint64_t noise_counter, int32_t* step_counter,
                              const float* support, float* ws,
                              float* q_values_out, int32_t* greedy_out, float* vmax_out,
                              dz_stream_t stream);
extern "C" int dz_rainbow_act_v(const dz_rainbow_act_args_t* a, dz_stream_t stream) {
  DZ_REQUIRE(a);
  return dz_rainbow_act(a->num_actions, a->num_atoms, a->batch, a->params, a->states, a->noise,
                        a->noise_seed, a->noise_counter, a->step_counter, a->support, a->ws,
                        a->q_values_out, a->greedy_out, a->vmax_out, stream);
}
extern "C" int dz_rainbow_act(int num_actions, int num_atoms, int batch, const float* params,
                              const uint8_t* states, float* noise, uint64_t noise_seed,
                              uint64_t noise_counter, int32_t* step_counter,
                              const float* support, float* ws,
                              float* q_values_out, int32_t* greedy_out, float* vmax_out,
                              dz_stream_t stream) {
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
  const int ld2 = L.adv2_ld + L.val2_ld;
  const bool fuse = (size_t)ld2 * sizeof(float) <= 48 * 1024;
  const NoiseParams nq = {noise, (long)L.noise_stride, noise_seed, noise_counter, step_counter};
  // One observation: the whole decision is ONE launch (dz_act_one.h).
  if (batch == 1 && ld2 <= 1024 && num_atoms <= 64 && step_counter) {
    ActOneParams q;
    q.obs = states; q.prm = params;
    for (int i = 0; i < 3; ++i) { q.conv_w[i] = L.conv_w[i]; q.conv_b[i] = L.conv_b[i]; }
    q.fc1_mu_w = L.fc1_mu_w; q.fc1_sig_w = L.fc1_sig_w; q.fc1_mu_b = L.fc1_mu_b;
    q.fc1_sig_b = L.fc1_sig_b; q.fc1_ld = L.fc1_ld;
    q.noise = noise; q.n_noise = (int)L.noise_stride; q.seed = noise_seed; q.counter = noise_counter;
    q.step = step_counter;
    q.n_eps_in[0] = (int)L.n_adv1_in; q.n_eps_in[1] = (int)L.n_val1_in; q.n_fc1_out = (int)L.n_fc1_out;
    q.head[0] = H.fc2h[0]; q.head[1] = H.fc2h[1];
    q.fc2_sig_b = L.fc2_sig_b; q.n_fc2_out = (int)L.n_fc2_out;
    q.ld2 = ld2; q.val_off = L.adv2_ld; q.A = num_actions; q.K = num_atoms;
    q.support = support; q.fc2_out = ws + L.ws_fc2_out;
    q.q_out = q_values_out; q.greedy_out = greedy_out; q.vmax_out = vmax_out;
    q.tiles0 = (L.adv2_ld + 31) / 32; q.tiles = q.tiles0 + (L.val2_ld + 31) / 32;
    q.bump = step_counter;
    q.sync = reinterpret_cast<unsigned*>(ws + L.ws_act_seams);   // zero in a fresh workspace, re-armed by the kernel
    q.set_floats = act_set_floats(1024); q.ncg = 8; q.part_ld = 1024;
    q.spin_limit = g_dz_act_spin_limit;
#ifdef DZ_ACT_STAMPS
    q.dbg = reinterpret_cast<long long*>(ws + L.ws_dfeat_part);
#endif
    hipLaunchKernelGGL(rainbow_act_one_kernel,
                       dim3((unsigned)(kActTorsoBlocks + kActFc1Blocks + q.tiles)), dim3(256), 0,
                       s, q);
    DZ_LAUNCH_CHECK();
    return DZ_OK;
  }
  const bool prof = g_dz_prof_on;
  g_dz_prof_on = false;  // marks belong to dz_rainbow_learn
  // Few rows: the tail (fc1 epilogue + fc2 + q-values) is ONE launch that folds the
  // fc1 slabs itself (rainbow_act_tail_kernel): 5 launches per decision instead of 7.
  const bool tail = batch <= 8 && ld2 <= 1024 && num_atoms <= 64;
  rc = rainbow_forward(L, H, 1, batch, prm, nz, in, ws, s, &nq, fuse, tail);
  g_dz_prof_on = prof;
  if (rc) return rc;
  if (tail) {
    ActTailParams q;
    q.part = ws + L.ws_fc1_part; q.S = kFc1Splits; q.rows = batch;
    q.prm = params; q.nz = noise;
    q.fc1_mu_b = L.fc1_mu_b; q.fc1_sig_b = L.fc1_sig_b; q.n_fc1_out = (int)L.n_fc1_out;
    q.head[0] = H.fc2h[0]; q.head[1] = H.fc2h[1];
    q.fc2_sig_b = L.fc2_sig_b; q.n_fc2_out = (int)L.n_fc2_out;
    q.ld2 = ld2; q.val_off = L.adv2_ld; q.A = num_actions; q.K = num_atoms;
    q.support = support; q.fc2_out = ws + L.ws_fc2_out;
    q.q_out = q_values_out; q.greedy_out = greedy_out; q.vmax_out = vmax_out;
    q.tickets = reinterpret_cast<int*>(ws + L.ws_scalars + 8);  // 8 ints, zero in a fresh workspace
    q.tiles0 = (L.adv2_ld + 31) / 32; q.tiles = q.tiles0 + (L.val2_ld + 31) / 32;
    q.bump = step_counter;
    hipLaunchKernelGGL(rainbow_act_tail_kernel, dim3((unsigned)q.tiles, (unsigned)batch), dim3(256),
                       0, s, q);
    DZ_LAUNCH_CHECK();
    return DZ_OK;
  }
  if (fuse) {
    HeadPre pre = {};
    pre.part = ws + L.ws_fc2_part; pre.S = kFc2Splits; pre.rows = batch;
    for (int g = 0; g < kG; ++g) { pre.prm[g] = params; pre.nz[g] = noise; }
    pre.b_sig = L.fc2_sig_b; pre.eps_out = (int)L.n_fc2_out;
    hipLaunchKernelGGL(rainbow_q_values_kernel<1>, dim3(batch), dim3(256),
                       (size_t)ld2 * sizeof(float), s, ws + L.ws_fc2_out, ld2, L.adv2_ld,
                       num_actions, num_atoms, support, q_values_out, greedy_out, vmax_out, pre,
                       step_counter);
  } else {
    hipLaunchKernelGGL(rainbow_q_values_kernel<0>, dim3(batch), dim3(64), 0, s,
                       ws + L.ws_fc2_out, ld2, L.adv2_ld, num_actions, num_atoms, support,
                       q_values_out, greedy_out, vmax_out, HeadPre{}, step_counter);
  }
  DZ_LAUNCH_CHECK();
  return DZ_OK;
}

extern "C" int dz_noise_fill(float* noise, int64_t count, uint64_t seed,
                             uint64_t counter, dz_stream_t stream) {
  DZ_REQUIRE(noise && count > 0);
  NoiseParams q = {noise, (long)count, seed, counter, nullptr};
  hipLaunchKernelGGL(noise_fill_kernel, dim3((unsigned)((count + 255) / 256)),
                     dim3(256), 0, dz_s(stream), q);
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
