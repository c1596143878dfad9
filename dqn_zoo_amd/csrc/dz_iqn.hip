// Learner step and actor apply of the IQN agent on one MI355X
// (ref: iqn/agent.py:176-247, networks.py:264-292).
//
// Row layout: every per-sample tensor has one row per (apply, batch element,
// tau sample): rows [0, B*n0) online(s_tm1, tau_tm1), then B*n1 rows
// target(s_t, tau_sel), then B*n2 rows target(s_t, tau_t).  Torso groups:
// g0 = online(s_tm1), g1 = target(s_t) -- shared by the two target applies.
#include "dz_iqn_ops.h"
#include "dz_torso.h"
#include "dz_iqn_act.h"
#include "dz_iqn_emb.h"
#include "dz_iqn_fc1_dma.h"

namespace {
constexpr int kS_iqn_fc2w = 32;   // row splits of the fc2 weight gradient
constexpr int kS_iqn_embw = 8;    // row splits of the embedding weight gradient (4: +6 us; 16, 32: +-0)
constexpr int kS_iqn_bias = 32;   // row splits of the bias column sums
}
extern "C" int dz_iqn_layout(int A, int latent, int B, int n0, int n1, int n2,
                             dz_iqn_layout_t* L) {
  DZ_REQUIRE(L && A > 0 && B > 0 && B <= 1024 && latent >= 16 && latent % 16 == 0);
  DZ_REQUIRE(n0 > 0 && n0 <= 256 && n1 > 0 && n2 > 0 && n2 <= 256);
  L->num_actions = A; L->latent_dim = latent; L->batch = B;
  L->samples[0] = n0; L->samples[1] = n1; L->samples[2] = n2;
  L->emb_ld = kFlat; L->fc1_ld = kHid; L->fc2_ld = (int32_t)align4(A); L->pad_ = 0;
  int64_t o = 0;
  const int64_t cw[3] = {256 * 32, 512 * 64, 576 * 64};
  const int64_t cb[3] = {32, 64, 64};
  for (int i = 0; i < 3; ++i) {
    L->conv_w[i] = o; o = align4(o + cw[i]);
    L->conv_b[i] = o; o = align4(o + cb[i]);
  }
  L->emb_w = o; o = align4(o + (int64_t)latent * kFlat);
  L->emb_b = o; o = align4(o + kFlat);
  L->fc1_w = o; o = align4(o + (int64_t)kFlat * kHid);
  L->fc1_b = o; o = align4(o + kHid);
  L->fc2_w = o; o = align4(o + (int64_t)kHid * L->fc2_ld);
  L->fc2_b = o; o = align4(o + A);
  L->param_count = o;
  L->param_count_ref = 77984 + (int64_t)latent * kFlat + kFlat + (int64_t)kFlat * kHid +
                       kHid + (int64_t)kHid * A + A;
  const int64_t M0 = (int64_t)B * n0, Mt = (int64_t)B * (n0 + n1 + n2), ld2 = L->fc2_ld;
  int64_t w = 0;
  auto take = [&](int64_t n) { int64_t r = w; w = align4(w + n); return r; };
  L->ws_act1 = take(2LL * B * 400 * 32);
  L->ws_act2 = take(2LL * B * 81 * 64);
  L->ws_feat = take(2LL * B * kFlat);
  L->ws_cos = take(Mt * latent);
  L->ws_hin = take(Mt * kFlat);
  L->ws_temb = take(0);   // (not stored since round 5: the mix backward pass works from head_in)
  L->ws_h1 = take(Mt * kHid);
  L->ws_out = take(Mt * ld2);
  L->ws_dout = take(M0 * ld2);
  L->ws_dh1 = take(M0 * kHid);
  L->ws_dhin = take(M0 * kFlat);
  L->ws_dfeat = take((int64_t)B * kFlat);
  L->ws_dact2 = take((int64_t)B * 81 * 64);
  L->ws_dact1 = take((int64_t)B * 400 * 32);
  L->ws_wgrad_part = take(torso_wgrad_part_elems());
  L->ws_fc2w_part = take((int64_t)kS_iqn_fc2w * kHid * ld2);
  L->ws_embw_part = take((int64_t)kS_iqn_embw * latent * kFlat);
  {  // embedding bias partials ([max(B, M0/32)][kFlat]) + the fused mix backward's dfeat partials
     // ([M0/32][kFlat]) + fc1 / fc2 bias partials
    const int64_t mixb = (M0 + 31) / 32;
    L->ws_bias_part = take((mixb > B ? mixb : (int64_t)B) * kFlat + mixb * kFlat +
                           (int64_t)kS_iqn_bias * (kHid + ld2));
  }
  L->ws_norm_part = take(kNormBlocks);
  L->ws_scalars = take(16);
  L->ws_zeros = take(kFlat + 1024);
  L->ws_act_seams = take(kIqnActSeamWords);   // (the last region: callers clear [ws_act_seams, ws_count))
  L->ws_count = w;
  return DZ_OK;
}

namespace {

struct IqnApplies {
  int G;                          // applies (row groups)
  int rows[3], row0[3], samples[3];
  int feat_row0[3];               // first torso-feature row of each apply
  const float* params[3];
  const float* tau[3];
};

// The cosine table of every apply's taus (networks.py:277-278): a side job of the step's conv1
// launch (torso_forward_side<IqnCosSide>).
IqnCosParams iqn_cos_params(const dz_iqn_layout_t& L, const IqnApplies& ap, float* ws) {
  IqnCosParams q;
  q.t0 = ap.tau[0]; q.t1 = ap.tau[ap.G > 1 ? 1 : 0]; q.t2 = ap.tau[ap.G > 2 ? 2 : 0];
  q.n0 = ap.rows[0]; q.n1 = ap.G > 1 ? ap.rows[1] : 0; q.n2 = ap.G > 2 ? ap.rows[2] : 0;
  q.latent = L.latent_dim; q.out = ws + L.ws_cos;
  return q;
}

// tau embedding -> mix with the state embedding -> value head, for every apply (the cosine
// table is already in ws_cos).
int iqn_head_forward(const dz_iqn_layout_t& L, const IqnApplies& ap, float* ws,
                     float* temb, hipStream_t s) {
  int rc;
  const int latent = L.latent_dim, ld2 = L.fc2_ld;
  int max_rows = 0;
  for (int g = 0; g < ap.G; ++g) if (ap.rows[g] > max_rows) max_rows = ap.rows[g];
  IqnLinParams p;
  p.G = ap.G;
  for (int g = 0; g < 3; ++g) {
    const int gg = g < ap.G ? g : 0;
    p.row0[g] = ap.row0[gg]; p.rows[g] = ap.rows[gg]; p.params[g] = ap.params[gg];
    p.feat_row0[g] = ap.feat_row0[gg]; p.samples[g] = ap.samples[gg];
  }
  {  // relu(cos @ Wemb + b) * feat -> head_in
    p.x = ws + L.ws_cos; p.ldx = latent; p.w_off = L.emb_w; p.b_off = L.emb_b;
    p.ldw = L.emb_ld; p.K = latent; p.N = kFlat; p.epi = IQN_EPI_MIX;
    p.out = ws + L.ws_hin; p.ldo = kFlat; p.feat = ws + L.ws_feat; p.temb = temb;
    // (GEMM form, any shape: 64x64 tiles, two 32-deep stages; one stage, 32-row / 32-column tiles
    // and 2-4 accumulators per wave all measured slower)
    bool whole_e = latent == 64 && kFlat % 64 == 0 && L.emb_ld == kFlat && temb == nullptr;
    for (int g = 0; g < ap.G; ++g) whole_e = whole_e && ap.rows[g] % 64 == 0;
    if (whole_e) {   // the embedding's own kernel (dz_iqn_emb.h): 43 -> 38 us
      IqnEmbParams q;
      q.cos = ws + L.ws_cos; q.G = ap.G;
      int tiles = 0;
      for (int g = 0; g < 3; ++g) {
        const int gg = g < ap.G ? g : 0;
        q.row0[g] = ap.row0[gg]; q.tiles[g] = g < ap.G ? ap.rows[gg] / 64 : 0;
        q.params[g] = ap.params[gg]; q.feat_row0[g] = ap.feat_row0[gg]; q.samples[g] = ap.samples[gg];
        tiles += q.tiles[g];
      }
      q.w_off = L.emb_w; q.b_off = L.emb_b; q.ldw = L.emb_ld; q.N = kFlat;
      q.feat = ws + L.ws_feat; q.out = ws + L.ws_hin;
      q.segs = (kFlat / 64 + 1) / 2;   // two column tiles per workgroup
      hipLaunchKernelGGL(iqn_emb_fwd_kernel, dim3((unsigned)(tiles * q.segs)), dim3(256), 0, s, q);
      DZ_LAUNCH_CHECK();
    } else {
      rc = dz_launch_gemm<IqnLin>(p, dim3(kFlat / IqnLin::BN, (unsigned)((max_rows + IqnLin::BM - 1) / IqnLin::BM), ap.G), s);
      if (rc) return rc;
    }
    DZ_PROF(s, "emb_fwd");
  }
  {  // relu(head_in @ W1 + b1)
    p.x = ws + L.ws_hin; p.ldx = kFlat; p.w_off = L.fc1_w; p.b_off = L.fc1_b;
    p.ldw = L.fc1_ld; p.K = kFlat; p.N = kHid; p.epi = IQN_EPI_BIAS_RELU;
    p.out = ws + L.ws_h1; p.ldo = kHid; p.feat = nullptr; p.temb = nullptr;
    // Three forms, by shape (all with tiles in XCD-aware order: the column tiles of one row slab
    // run on ONE XCD, so the 77 MB activation crosses the fabric once instead of 8 times, +4 %):
    //  (1) the learner's three applies (online rows % 64 == 0, the two s_t applies' target rows
    //      contiguous, the same parameters): TWO tile sets of 256 workgroups each in one launch --
    //      64x64 tiles with two column blocks per wave for the online rows and, for the target rows,
    //      128x64 tiles with four row blocks per wave at the reference's defaults (64 / 64 / 64 taus,
    //      batch 32: 2 048 + 4 096 rows; round 6) or 96x64 tiles with three (target rows % 96 == 0,
    //      e.g. a 32-tau policy: 2 048 + 3 072 rows, the shape round 5 tuned) --
    //      i.e. two workgroups per CU and 2-3 independent MFMA chains per wave, 1.6 x less
    //      L2 -> LDS traffic than (2); operands by LDS-DMA (dz_iqn_fc1_dma.h).  The chunk loop
    //      alone runs at 0.93 of the matrix pipe's rate at 2-3 waves per SIMD and at 0.78 at
    //      five (tools/micro/lds_mfma_micro.hip).
    //  (2) other whole-tile shapes (every group's rows % 64 == 0): 64x32 tiles, two waves share a
    //      tile's depth -- 5 120 x 512 outputs are 1 280 tiles = FIVE per CU (the 64x64 tiles of
    //      rounds 2-4 were 640 = 2.5 per CU: half the CUs ran three while the others ran two) --
    //      loaders WITHOUT masks, 32-deep stages (14.5 KB of LDS: five workgroups per CU resident),
    //      compiled for five waves per SIMD: at the default target of eight the allocator parks
    //      freshly loaded registers around the MFMA block, a copy that waits for the prefetch in
    //      front of the MFMAs.  Whole step, same box: 475 (form 3) -> 459 us; <2,1,2,2> 464, 64x64
    //      <2,2,1,1> 480 / <2,2,1,4> 475-478 / <2,2,1,7> 491, 32x64 <1,2,2,1> 473, 32x32 <1,1,4,1>
    //      486; occupancy target 4 / 6 / 8 instead of 5: +4 / +5 / +6.
    //  (3) any other shape (acting applies, odd batch sizes): masked loaders, 48-deep stages
    //      (<2,2,1,4> 64x64 +16 us on the step, <2,1,2,2> +-0, <2,1,2,4> +44; 2-4 accumulators
    //      per wave 258-325 us for this launch: EXPERIMENTS.md).
    using Fc1Fwd = IqnLinOp<2, 1, 2, 3>;
    using Fc1Full = IqnLinOp<2, 1, 2, 1, 1, 1, 1>;
    bool whole = kFlat % Fc1Full::BK == 0 && kHid % Fc1Full::BN == 0;
    for (int g = 0; g < ap.G; ++g) whole = whole && ap.rows[g] % Fc1Full::BM == 0;
    const bool sets_ok = whole && ap.G == 3 && ap.params[1] == ap.params[2] &&
                         ap.row0[2] == ap.row0[1] + ap.rows[1] && ap.rows[0] % 64 == 0 &&
                         kHid % 64 == 0 && kFlat % 32 == 0;
    const int trows = ap.rows[1] + ap.rows[2];
    // the reference's defaults (iqn/run_atari.py:98-100: 64 / 64 / 64 tau samples, batch 32):
    // 2 048 online + 4 096 target rows -- 64x64 tiles (two column blocks per wave) + 128x64 tiles
    // (FOUR row blocks per wave): 256 + 256 workgroups, one of each per CU
    const bool two_sets128 = sets_ok && trows % 128 == 0 && trows == 2 * ap.rows[0];
    const bool two_sets = sets_ok && !two_sets128 && trows % 96 == 0;
    if (two_sets128) {
      using CA = DzDmaCfg<1, 2, 2, 1, 1, 2, true, false>;
      using CB = DzDmaCfg<4, 1, 1, 2, 1, 2, true, false>;
      DzDmaOperands qa, qb;
      qa.a = p.x + (long)ap.row0[0] * kFlat; qa.lda = kFlat;
      qa.b = ap.params[0] + L.fc1_w; qa.ldb = L.fc1_ld; qa.K = kFlat;
      qb = qa;
      qb.a = p.x + (long)ap.row0[1] * kFlat; qb.b = ap.params[1] + L.fc1_w;
      const IqnFwdEpi::Params ea = {ap.params[0] + L.fc1_b, ws + L.ws_h1 + (long)ap.row0[0] * kHid, kHid};
      const IqnFwdEpi::Params eb = {ap.params[1] + L.fc1_b, ws + L.ws_h1 + (long)ap.row0[1] * kHid, kHid};
      rc = dz_launch_dma_gemm2<CA, IqnFwdEpi, CB, IqnFwdEpi, 2>(
          qa, ea, dim3(kHid / CA::BN, (unsigned)(ap.rows[0] / CA::BM), 1),
          qb, eb, dim3(kHid / CB::BN, (unsigned)(trows / CB::BM), 1), s);
    } else if (two_sets) {
      // 32-deep stages, two stage buffers (40 KB per workgroup).  Whole step, same box: form (2)
      // 437.7; the same two tile sets on the register-staged skeleton (64-deep stages) 432.7-434.2;
      // this 419.6; three stage buffers 424.8; 64-deep stages x two buffers 422.9.
      using CA = DzDmaCfg<1, 2, 2, 1, 1, 2, true, false>;
      using CB = DzDmaCfg<3, 1, 1, 2, 1, 2, true, false>;
      DzDmaOperands qa, qb;
      qa.a = p.x + (long)ap.row0[0] * kFlat; qa.lda = kFlat;
      qa.b = ap.params[0] + L.fc1_w; qa.ldb = L.fc1_ld; qa.K = kFlat;
      qb = qa;
      qb.a = p.x + (long)ap.row0[1] * kFlat; qb.b = ap.params[1] + L.fc1_w;
      const IqnFwdEpi::Params ea = {ap.params[0] + L.fc1_b, ws + L.ws_h1 + (long)ap.row0[0] * kHid, kHid};
      const IqnFwdEpi::Params eb = {ap.params[1] + L.fc1_b, ws + L.ws_h1 + (long)ap.row0[1] * kHid, kHid};
      rc = dz_launch_dma_gemm2<CA, IqnFwdEpi, CB, IqnFwdEpi, 2>(
          qa, ea, dim3(kHid / CA::BN, (unsigned)(ap.rows[0] / CA::BM), 1),
          qb, eb, dim3(kHid / CB::BN, (unsigned)((ap.rows[1] + ap.rows[2]) / CB::BM), 1), s);
    } else if (whole)
      rc = dz_launch_gemm_xcd_occ<Fc1Full, 5>(p, dim3(kHid / Fc1Full::BN, (unsigned)(max_rows / Fc1Full::BM), ap.G), s);
    else
      rc = dz_launch_gemm_xcd<Fc1Fwd>(p, dim3(kHid / Fc1Fwd::BN, (unsigned)((max_rows + Fc1Fwd::BM - 1) / Fc1Fwd::BM), ap.G), s);
    if (rc) return rc;
    DZ_PROF(s, "fc1_fwd");
  }
  {  // h1 @ W2 + b2
    p.x = ws + L.ws_h1; p.ldx = kHid; p.w_off = L.fc2_w; p.b_off = L.fc2_b;
    p.ldw = ld2; p.K = kHid; p.N = L.num_actions; p.epi = IQN_EPI_BIAS;
    p.out = ws + L.ws_out; p.ldo = ld2;
    // N = num_actions <= 32 columns: 32x32 tiles with the FOUR waves sharing the depth (128-deep
    // stages) -- 160 workgroups of 4 stages instead of 80 of 16 (the 64x64 form: 24 -> 13 us)
    using Fc2Fwd = IqnLinOp<1, 1, 4, 4>;
    rc = dz_launch_gemm<Fc2Fwd>(
        p, dim3((L.num_actions + Fc2Fwd::BN - 1) / Fc2Fwd::BN, (unsigned)((max_rows + Fc2Fwd::BM - 1) / Fc2Fwd::BM), ap.G), s);
    if (rc) return rc;
    DZ_PROF(s, "fc2_fwd");
  }
  return DZ_OK;
}

}  // namespace

extern "C" int dz_iqn_learn(const dz_iqn_args_t* a, int phases, dz_stream_t stream) {
  DZ_REQUIRE(a && a->online && a->target && a->ws && a->s_tm1 && a->s_t && a->a_tm1 &&
             a->r_t && a->discount_t && a->losses && a->tau_tm1 && a->tau_sel && a->tau_t);
  dz_iqn_layout_t L;
  int rc = dz_iqn_layout(a->num_actions, a->latent_dim, a->batch, a->samples[0],
                         a->samples[1], a->samples[2], &L);
  if (rc) return rc;
  hipStream_t s = dz_s(stream);
  float* ws = a->ws;
  const int B = a->batch, A = a->num_actions, ld2 = L.fc2_ld, latent = L.latent_dim;
  const int n0 = a->samples[0], n1 = a->samples[1], n2 = a->samples[2];
  const int M0 = B * n0;
  const float* zeros = ws + L.ws_zeros;
  const TorsoBufs T = {L.conv_w, L.conv_b, ws + L.ws_act1, ws + L.ws_act2, ws + L.ws_feat};
  if (g_dz_prof_on) dz_prof_begin(s);

  if (phases & DZ_PHASE_FORWARD) {
    const float* prm[2] = {a->online, a->target};
    const uint8_t* in[2] = {a->s_tm1, a->s_t};
    IqnApplies ap;
    ap.G = 3;
    ap.rows[0] = B * n0; ap.rows[1] = B * n1; ap.rows[2] = B * n2;
    ap.row0[0] = 0; ap.row0[1] = ap.rows[0]; ap.row0[2] = ap.rows[0] + ap.rows[1];
    ap.samples[0] = n0; ap.samples[1] = n1; ap.samples[2] = n2;
    ap.feat_row0[0] = 0; ap.feat_row0[1] = B; ap.feat_row0[2] = B;
    ap.params[0] = a->online; ap.params[1] = a->target; ap.params[2] = a->target;
    ap.tau[0] = a->tau_tm1; ap.tau[1] = a->tau_sel; ap.tau[2] = a->tau_t;
    const IqnCosParams cq = iqn_cos_params(L, ap, ws);
    rc = torso_forward_side<IqnCosSide>(T, 2, B, prm, in, s, cq, IqnCosSide::blocks(cq));
    if (rc) return rc;
    rc = iqn_head_forward(L, ap, ws, nullptr, s);   // (temb is not stored: iqn_mix_bwd_kernel)
    if (rc) return rc;
    hipLaunchKernelGGL(iqn_loss_kernel, dim3(B), dim3(256), 0, s, ws + L.ws_out, ld2, B, A,
                       n0, n1, n2, a->tau_tm1, a->a_tm1, a->r_t, a->discount_t, a->huber,
                       ws + L.ws_dout, a->losses);
    DZ_LAUNCH_CHECK();
    DZ_PROF(s, "loss");
  }

  bool norm_skipped = false;   // (DZ_SC_GNORM then reads 0)
  if (phases & DZ_PHASE_BACKWARD) {
    DZ_REQUIRE(a->grad);
    float* grad = a->grad;
    FcHead h1, h2;
    h1.w_mu = L.fc1_w; h1.w_sig = L.fc1_w; h1.ldw = L.fc1_ld; h1.N = kHid; h1.K = kFlat;
    h1.x_off = 0; h1.eps_in = 0; h1.eps_out = 0; h1.out_off = 0;
    h2 = h1;
    h2.w_mu = L.fc2_w; h2.w_sig = L.fc2_w; h2.ldw = ld2; h2.N = A; h2.K = kHid;
    {  // fc2: weight-gradient partials + input gradient
      IqnWgradParams w;
      w.x = ws + L.ws_h1; w.ldx = kHid; w.dy = ws + L.ws_dout; w.ldy = ld2; w.M = M0;
      w.K = kHid; w.N = A; w.ldw = ld2; w.S = kS_iqn_fc2w; w.part = ws + L.ws_fc2w_part;
      FcDgradParams d;
      d.dy = ws + L.ws_dout; d.ldy = ld2; d.M = M0; d.NH = 1; d.S = 1; d.noisy = 0;
      d.params = a->online; d.noise = zeros; d.head[0] = h2; d.head[1] = h2;
      d.part = ws + L.ws_dh1; d.ldo = kHid; d.K = kHid; d.x_off = 0;
      d.relu_mask = ws + L.ws_h1;  // dh1 *= (h1 > 0) in the store (S == 1)
      rc = dz_launch_gemm2<IqnWg, IqnDg>(
          w, dim3((A + IqnWg::BN - 1) / IqnWg::BN, kHid / IqnWg::BM, kS_iqn_fc2w), d,
          dim3(kHid / IqnDg::BN, (M0 + IqnDg::BM - 1) / IqnDg::BM, 1), s);
      if (rc) return rc;
      DZ_PROF(s, "fc2_wgrad+dgrad");
    }
    // bias-gradient partials.  Embedding: one slab per 32-row block out of the fused mix backward
    // (or one per batch element out of iqn_mix_bwd_kernel); fc1's and fc2's column sums ride in
    // the embedding weight-gradient launch.  mix_s1: the fused form's dfeat partials.
    const int mix_blocks = (M0 + 31) / 32;
    float* bp_emb = ws + L.ws_bias_part;
    float* mix_s2 = bp_emb;
    float* mix_s1 = bp_emb + (long)(mix_blocks > B ? mix_blocks : B) * kFlat;
    float* bp_fc1 = mix_s1 + (long)mix_blocks * kFlat;
    float* bp_fc2 = bp_fc1 + (long)kS_iqn_bias * kHid;
    bool fuse_mix = false;
    {  // fc1: weight gradient (straight into grad) + input gradient
      IqnWgradParams w;
      w.x = ws + L.ws_hin; w.ldx = kFlat; w.dy = ws + L.ws_dh1; w.ldy = kHid; w.M = M0;
      w.K = kFlat; w.N = kHid; w.ldw = L.fc1_ld; w.S = 1; w.part = grad + L.fc1_w;
      IqnDgradParams d;   // (no ReLU between the mix and fc1: no mask)
      d.dy = ws + L.ws_dh1; d.ldy = kHid; d.w = a->online + L.fc1_w; d.ldw = L.fc1_ld;
      d.M = M0; d.N = kHid; d.K = kFlat; d.dx = ws + L.ws_dhin; d.ldo = kFlat;
      d.relu_mask = nullptr;
      // 64x64 tiles; 64-deep stages for the weight gradient, 32-deep for the input gradient (one
      // launch: the LDS block is the larger of the two).  With the loaders selecting on addresses
      // (third loader rule) the pair went 158 -> 140 us, the deeper weight-gradient stages took
      // another 10.  Measured against it, whole step, same box: weight gradient <2,2,1,2> +10.5,
      // <2,1,2,3> +3, <2,2,1,3> +10, <2,2,1,5> +1.5, <2,2,1,6> +3 (with the input gradient at
      // 64-deep stages); input gradient <2,2,1,4> +3 (alone -9.5, together with the deeper weight
      // gradient it loses), <2,2,1,3> +1, <2,1,2,2> -4 alone / not combined, <2,1,2,4> +3,
      // <2,2,1,6> +18, both <2,2,1,8> +32: EXPERIMENTS.md.
      using Dg1 = IqnDgradOp<2, 2, 1, 2>;
      using Wg1 = IqnWgradOp<2, 2, 1, 4>;
      // whole tiles (every learner step): loaders without masks, compiled for three waves per SIMD
      // (-4.5 us on the step; targets 4 / 5 / 6: -3 / +-0 / -4; 64-deep input-gradient stages -4 at
      // target 5, not additive)
      using Dg1F = IqnDgradOp<2, 2, 1, 2, 1>;
      using Wg1F = IqnWgradOp<2, 2, 1, 4, 1, 1, 1>;
      const dim3 gw(kHid / Wg1::BN, kFlat / Wg1::BM, 1), gd(kFlat / Dg1::BN, (M0 + Dg1::BM - 1) / Dg1::BM, 1);
      bool whole = M0 % Wg1F::BK == 0 && M0 % Dg1F::BM == 0 && kHid % Dg1F::BK == 0 && kHid % Wg1F::BN == 0 &&
                   kFlat % Wg1F::BM == 0 && kFlat % Dg1F::BN == 0 && L.fc1_ld == kHid;
      // ... and, when the 32 rows of a wave's block belong to one batch element (samples % 32
      // == 0), with the backward of the mix in the input gradient's store (IqnDgradMixEpi): the
      // separate pass over the 26 MB gradient and its launch are gone
      // ... on the LDS-DMA mainloop (dz_dma_gemm.h), as for the forward: the weight gradient as
      // 64x128 tiles (four column blocks per wave), the input gradient as 64x64 tiles (two), 32-deep
      // stages, two stage buffers, two workgroups per CU.  Whole step, same box: register-staged
      // skeleton pair (the same store on IqnDgradOp) 420.2; this 411.4-412.5; weight gradient 64x64 416.7;
      // input gradient 128x64 (four chains) 412.4 / 64x128 419.8; three stage buffers 416.1; 64-deep
      // stages 432.6; occupancy target 3: 414.4 / 425.6.
      using CW = DzDmaCfg<1, 4, 2, 1, 1, 2, false, false>;
      using CD = DzDmaCfg<1, 2, 2, 1, 1, 2, true, true>;
      static_assert(kFlat % CW::BM == 0 && kHid % CW::BN == 0 && kFlat % CD::BN == 0 && kHid % CD::BK == 0,
                    "the backward tile sets divide the layer");
      fuse_mix = whole && n0 % 32 == 0 && M0 % CW::BK == 0 && M0 % CD::BM == 0;
      if (fuse_mix) {
        DzDmaOperands qw, qd;
        qw.a = ws + L.ws_hin; qw.lda = kFlat; qw.b = ws + L.ws_dh1; qw.ldb = kHid; qw.K = M0;
        qd.a = ws + L.ws_dh1; qd.lda = kHid; qd.b = a->online + L.fc1_w; qd.ldb = L.fc1_ld; qd.K = kHid;
        const IqnPlainEpi::Params ew = {grad + L.fc1_w, L.fc1_ld};
        const IqnDgradMixEpi::Params ed = {ws + L.ws_dhin, kFlat, ws + L.ws_hin, ws + L.ws_feat, n0, mix_s1, mix_s2};
        rc = dz_launch_dma_gemm2<CW, IqnPlainEpi, CD, IqnDgradMixEpi, 2>(
            qw, ew, dim3(kHid / CW::BN, kFlat / CW::BM, 1), qd, ed, dim3(kFlat / CD::BN, M0 / CD::BM, 1), s);
      } else if (whole) {
        rc = dz_launch_gemm2_occ<Wg1F, Dg1F, 3>(
            w, dim3(kHid / Wg1F::BN, kFlat / Wg1F::BM, 1), d, dim3(kFlat / Dg1F::BN, M0 / Dg1F::BM, 1), s);
      } else {
        rc = dz_launch_gemm2<Wg1, Dg1>(w, gw, d, gd, s);
      }
      if (rc) return rc;
      DZ_PROF(s, fuse_mix ? "fc1_wgrad+dgrad+mix" : "fc1_wgrad+dgrad");
    }
    if (!fuse_mix) {
      hipLaunchKernelGGL(iqn_mix_bwd_kernel, dim3((kFlat + 255) / 256, B), dim3(256), 0, s,
                         ws + L.ws_dhin, ws + L.ws_hin, ws + L.ws_feat, B, n0, kFlat,
                         ws + L.ws_dfeat, bp_emb);
      DZ_LAUNCH_CHECK();
      DZ_PROF(s, "mix_bwd");
    }
    {  // tau-embedding weight-gradient partials (+ the two column sums -- and the fused form's
       // dfeat fold -- as side workgroups)
      IqnWgradParams w;
      w.x = ws + L.ws_cos; w.ldx = latent; w.dy = ws + L.ws_dhin; w.ldy = kFlat; w.M = M0;
      w.K = latent; w.N = kFlat; w.ldw = L.emb_ld; w.S = kS_iqn_embw;
      w.part = ws + L.ws_embw_part;
      ColPartJobs J;
      J.j[0] = {ws + L.ws_dh1, M0, kHid, kHid, bp_fc1};
      J.j[1] = {ws + L.ws_dout, M0, A, ld2, bp_fc2};
      J.S = kS_iqn_bias;
      J.end0 = ColsumSide::blocks_of(J.j[0], J.S);
      IqnBwdSideParams sp;
      sp.col = J;
      sp.col_blocks = J.end0 + ColsumSide::blocks_of(J.j[1], J.S);
      sp.df = {mix_s1, ws + L.ws_feat, ws + L.ws_dfeat, B, kFlat, n0 / 32};
      const unsigned df_blocks = fuse_mix ? (unsigned)(((long)B * kFlat + 255) / 256) : 0u;
      rc = dz_launch_gemm_side<IqnWg, IqnBwdSide>(
          w, dim3(kFlat / IqnWg::BN, (latent + IqnWg::BM - 1) / IqnWg::BM, kS_iqn_embw), sp,
          sp.col_blocks + df_blocks, s);
      if (rc) return rc;
      DZ_PROF(s, "emb_wgrad+colsum");
    }
    ReduceJob conv_jobs[3];
    rc = torso_backward(T, B, a->online, a->s_tm1, ws + L.ws_dfeat, ws + L.ws_dact2,
                        ws + L.ws_dact1, ws + L.ws_wgrad_part, grad, conv_jobs, s);
    if (rc) return rc;
    {
      ReduceJobs8 J;
      J.n = 8;
      for (int j = 0; j < 3; ++j) J.r[j] = conv_jobs[j];
      J.r[3] = {ws + L.ws_fc2w_part, kS_iqn_fc2w, (long)kHid * ld2, grad + L.fc2_w};
      J.r[4] = {ws + L.ws_embw_part, kS_iqn_embw, (long)latent * kFlat, grad + L.emb_w};
      J.r[5] = {bp_emb, fuse_mix ? mix_blocks : B, (long)kFlat, grad + L.emb_b};
      J.r[6] = {bp_fc1, kS_iqn_bias, (long)kHid, grad + L.fc1_b};
      J.r[7] = {bp_fc2, kS_iqn_bias, (long)A, grad + L.fc2_b};
      unsigned acc = 0;
      for (int j = 0; j < 8; ++j) { acc += (unsigned)((J.r[j].n + 63) / 64); J.r_end[j] = acc; }
      // without clipping (iqn/run_atari.py:213-215: plain optax.adam) nothing reads the global
      // norm: no norm launch, the step count is incremented here
      norm_skipped = (phases & DZ_PHASE_OPTIMIZER) && !(a->max_norm > 0.f);
      if (norm_skipped) J.bump_count = a->opt_count;
      hipLaunchKernelGGL(reduce_jobs_kernel, dim3(acc), dim3(256), 0, s, J);
      DZ_LAUNCH_CHECK();
      DZ_PROF(s, "finalize_grads");
    }
  }

  if (phases & DZ_PHASE_OPTIMIZER) {
    DZ_REQUIRE(a->grad && a->opt_m && a->opt_v && a->opt_count);
    float* sc = ws + L.ws_scalars;
    if (!norm_skipped) {
      hipLaunchKernelGGL(sumsq_kernel, dim3(kNormBlocks), dim3(256), 0, s, a->grad,
                         (long)L.param_count, ws + L.ws_norm_part, a->opt_count);
      DZ_LAUNCH_CHECK();
      DZ_PROF(s, "grad_sumsq");
    }
    hipLaunchKernelGGL(adam_kernel, dim3(2048), dim3(256), 0, s, a->online, a->grad, a->opt_m,
                       a->opt_v, (long)(L.param_count >> 2), ws + L.ws_norm_part,
                       norm_skipped ? 0 : kNormBlocks,
                       a->opt_count, a->losses, zeros, B, sc, a->lr, a->b1, a->b2, a->eps,
                       a->max_norm);
    DZ_LAUNCH_CHECK();
    DZ_PROF(s, "adam");
  }
  return DZ_OK;
}

extern "C" int dz_iqn_apply(int A, int latent, int B, int samples, const float* params,
                            const uint8_t* states, const float* taus, float* ws,
                            float* q_dist_out, float* q_values_out, int32_t* greedy_out,
                            float* vmax_out, dz_stream_t stream) {
  DZ_REQUIRE(params && states && taus && ws);
  dz_iqn_layout_t L;
  // the workspace of a (B, samples, 1, 1) layout is a prefix-compatible subset
  int rc = dz_iqn_layout(A, latent, B, samples, 1, 1, &L);
  if (rc) return rc;
  hipStream_t s = dz_s(stream);
  const TorsoBufs T = {L.conv_w, L.conv_b, ws + L.ws_act1, ws + L.ws_act2, ws + L.ws_feat};
  const float* prm[1] = {params};
  const uint8_t* in[1] = {states};
  const bool prof = g_dz_prof_on;
  g_dz_prof_on = false;
  IqnApplies ap;
  ap.G = 1;
  ap.rows[0] = B * samples; ap.row0[0] = 0; ap.samples[0] = samples; ap.feat_row0[0] = 0;
  ap.params[0] = params; ap.tau[0] = taus;
  const IqnCosParams cq = iqn_cos_params(L, ap, ws);
  rc = torso_forward_side<IqnCosSide>(T, 1, B, prm, in, s, cq, IqnCosSide::blocks(cq));
  if (!rc) rc = iqn_head_forward(L, ap, ws, nullptr, s);
  g_dz_prof_on = prof;
  if (rc) return rc;
  if (q_dist_out)
    DZ_HIP_CHECK(hipMemcpy2DAsync(q_dist_out, (size_t)A * sizeof(float), ws + L.ws_out,
                                  (size_t)L.fc2_ld * sizeof(float), (size_t)A * sizeof(float),
                                  (size_t)B * samples, hipMemcpyDeviceToDevice, s));
  if (q_values_out || greedy_out || vmax_out) {
    hipLaunchKernelGGL(iqn_q_values_kernel, dim3(B), dim3(64), 0, s, ws + L.ws_out, L.fc2_ld,
                       A, samples, q_values_out, greedy_out, vmax_out);
    DZ_LAUNCH_CHECK();
  }
  return DZ_OK;
}

extern "C" int dz_iqn_act(int A, int latent, int samples, const float* params, const uint8_t* state,
                          uint64_t tau_seed, uint64_t tau_counter, float* taus_out, float* ws,
                          void* pairs_out, dz_stream_t stream) {
  DZ_REQUIRE(params && state && ws && pairs_out && ((uintptr_t)pairs_out & 7) == 0);
  DZ_REQUIRE(A > 0 && A <= 32 && samples > 0 && samples <= kIqnActMaxTaus && latent >= 16 &&
             latent <= kIqnActMaxLatent && latent % 8 == 0);
  dz_iqn_layout_t L;
  int rc = dz_iqn_layout(A, latent, 1, samples, 1, 1, &L);
  if (rc) return rc;
  IqnActParams q;
  q.obs = state; q.prm = params;
  for (int i = 0; i < 3; ++i) { q.conv_w[i] = L.conv_w[i]; q.conv_b[i] = L.conv_b[i]; }
  q.sync = reinterpret_cast<unsigned*>(ws + L.ws_act_seams);   // zero in a fresh workspace
  q.set_floats = act_set_floats(kIqnActPartLd); q.ncg = 4; q.part_ld = kIqnActPartLd;
  q.spin_limit = g_dz_act_spin_limit;
  q.fc1_mu_w = L.fc1_w; q.fc1_ld = L.fc1_ld;
#ifdef DZ_ACT_STAMPS
  q.dbg = reinterpret_cast<long long*>(ws + L.ws_hin);
#endif
  q.latent = latent; q.N = samples; q.A = A; q.ld2 = L.fc2_ld;
  q.emb_w = L.emb_w; q.emb_b = L.emb_b; q.fc1_b = L.fc1_b; q.fc2_w = L.fc2_w; q.fc2_b = L.fc2_b;
  q.tau_seed = tau_seed; q.tau_counter = tau_counter; q.taus_out = taus_out;
  q.out = ws + L.ws_out; q.pairs_out = (unsigned long long*)pairs_out;
  hipLaunchKernelGGL(iqn_act_one_kernel,
                     dim3((unsigned)(kActTorsoBlocks + kIqnActFc1Blocks + samples)), dim3(256), 0,
                     dz_s(stream), q);
  DZ_LAUNCH_CHECK();
  return DZ_OK;
}

extern "C" int dz_uniform_fill(float* out, int64_t n, uint64_t seed, uint64_t counter,
                               const int32_t* step, dz_stream_t stream) {
  DZ_REQUIRE(out && n > 0);
  hipLaunchKernelGGL(uniform_fill_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                     dz_s(stream), out, (long)n, seed, counter, step);
  DZ_LAUNCH_CHECK();
  return DZ_OK;
}
