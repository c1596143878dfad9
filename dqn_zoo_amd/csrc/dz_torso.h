// The convolutional torso shared by the dense-head and IQN learner steps
// (ref: networks.py:186-204 dqn_torso): forward for G parameter/input groups
// and the backward pass of group 0, as launches of the implicit-GEMM Ops.
#pragma once

#include "dz_qnet_kernels.h"
#include "dz_sumtree_dev.h"
#include "dz_conv_dma.h"
#include "dz_conv1_dma.h"
#include "dz_conv_bwd_dma.h"

// A/B switches (tools/build_variant.sh): SUBN * 100 + KT * 10 + NBUF of the LDS-DMA forward
// convolutions, 0 = the register-staged ConvFwdOp
#ifndef DZ_CONV2_DMA
#define DZ_CONV2_DMA 223
#endif
#ifndef DZ_CONV3_DMA
#define DZ_CONV3_DMA 223
#endif
#ifndef DZ_CONV1_DMA   // row blocks of 16 per wave (workgroup = 32 x this rows), 0 = ConvFwdOp
#define DZ_CONV1_DMA 21
#endif
// backward pairs on the LDS-DMA skeleton: (dgrad SUBN) * 1000 + (dgrad KT) * 100 + (wgrad KT) * 10 + NBUF, 0 = off
#ifndef DZ_CONV3_BWD_DMA
#define DZ_CONV3_BWD_DMA 1123
#endif
#ifndef DZ_CONV2_BWD_DMA
#define DZ_CONV2_BWD_DMA 123
#endif
#ifndef DZ_CONV1_WG_DMA   // conv1's weight gradient: KT * 10 + NBUF, 0 = ConvWgradOp
#define DZ_CONV1_WG_DMA 0
#endif
#ifndef DZ_CONV_DMA_OCC   // waves per SIMD the forward kernels are compiled for (one workgroup per CU)
#define DZ_CONV_DMA_OCC 2
#endif
#ifndef DZ_CONV_BWD_OCC   // ... and the backward pairs (conv2's is 643 workgroups: three per CU must fit; 2: 10.6 us, 3: 9.3)
#define DZ_CONV_BWD_OCC 3
#endif

namespace {

struct TorsoBufs {
  const int64_t* conv_w;   // [3] parameter offsets
  const int64_t* conv_b;   // [3]
  float* act1;             // [G*B][20][20][32]
  float* act2;             // [G*B][9][9][64]
  float* feat;             // [G*B][3136]
};

// Forward launch shapes: the M dimension (3 applies x 32 images x output pixels)
// gives 300-600 workgroups of the shapes in dz_qnet_kernels.h on 256 CUs (the
// measured best of a sweep over 4-6 tile shapes per layer; XCD-ordered tiles made
// no difference for these: EXPERIMENTS.md).
template <class Op>
inline int launch_conv_fwd(const ConvFwdParams& p, int CO, int G, int B, hipStream_t s) {
  return dz_launch_gemm<Op>(p, dim3(CO / Op::BN, G * Op::tiles_per_group(B), 1), s);
}
template <class Op>
inline int launch_conv1_fwd(const ConvFwdParams& p, int G, int B, const NoiseParams* side,
                            hipStream_t s, const SeamClear* clr = nullptr) {
  const dim3 g(32 / Op::BN, G * Op::tiles_per_group(B), 1);
  if (clr) {  // + the step's seam buffers back to all-zero bits (dz_head_chain.h)
    StepPre q;
    q.noise = side ? *side : NoiseParams{};
    q.noise_blocks = side ? (unsigned)((side->n + 255) / 256) : 0u;
    q.clr = *clr;
    return dz_launch_gemm_side<Op, StepPreSide>(p, g, q, q.noise_blocks + clr->blocks, s);
  }
  if (side)  // the step's noise draw rides along as extra blocks: conv1 does not read it
    return dz_launch_gemm_side<Op, NoiseSide>(p, g, *side, (unsigned)((side->n + 255) / 256), s);
  return dz_launch_gemm<Op>(p, g, s);
}

// relu(conv3(relu(conv2(relu(conv1(u8/255)))))) for G groups; group g reads
// images in[g] with parameters prm[g].  `side`: optional noise draw fused into
// the conv1 launch.
inline ConvFwdParams torso_conv1_params(const TorsoBufs& T, int G, int B, const float* const* prm,
                                        const uint8_t* const* in, long long* dbg) {
  ConvFwdParams p;
  for (int g = 0; g < G; ++g) {
    p.in[g] = in[g]; p.in_img_base[g] = 0;
    p.w[g] = prm[g] + T.conv_w[0]; p.bias[g] = prm[g] + T.conv_b[0];
  }
  p.out = T.act1; p.B = B; p.G = G; p.dbg = dbg;
  return p;
}
inline int torso_forward_rest(const TorsoBufs& T, int G, int B, const float* const* prm,
                              hipStream_t s, long long* dbg) {
  int rc;
  {
    ConvFwdParams p;
    for (int g = 0; g < G; ++g) {
      p.in[g] = T.act1; p.in_img_base[g] = g * B;
      p.w[g] = prm[g] + T.conv_w[1]; p.bias[g] = prm[g] + T.conv_b[1];
    }
    p.out = T.act2; p.B = B; p.G = G; p.dbg = dbg ? dbg + 65536 * 8 : nullptr;
#if DZ_CONV2_DMA
    using C2 = ConvDmaCfg<20, 20, 32, 4, 2, 9, 9, 64, (DZ_CONV2_DMA / 100) % 10, (DZ_CONV2_DMA / 10) % 10, DZ_CONV2_DMA % 10, DZ_CONV2_DMA / 1000 ? DZ_CONV2_DMA / 1000 : 1>;   // NI * 1000 + SUBN * 100 + KT * 10 + NBUF
    if (C2::fits(B)) {
      ConvDmaParams q;
      for (int g = 0; g < DZ_MAX_GROUPS; ++g) {
        q.in[g] = (const float*)p.in[g < G ? g : 0]; q.in_img_base[g] = p.in_img_base[g < G ? g : 0];
        q.w[g] = p.w[g < G ? g : 0]; q.bias[g] = p.bias[g < G ? g : 0];
      }
      q.out = T.act2; q.B = B; q.G = G; q.dbg = p.dbg;
      rc = dz_launch_conv_dma_fwd<C2, DZ_CONV_DMA_OCC>(q, s);
    } else
#endif
    rc = launch_conv_fwd<Conv2Fwd>(p, 64, G, B, s);
    if (rc) return rc;
    DZ_PROF(s, "conv2_fwd");
  }
  {
    ConvFwdParams p;
    for (int g = 0; g < G; ++g) {
      p.in[g] = T.act2; p.in_img_base[g] = g * B;
      p.w[g] = prm[g] + T.conv_w[2]; p.bias[g] = prm[g] + T.conv_b[2];
    }
    p.out = T.feat; p.B = B; p.G = G; p.dbg = dbg ? dbg + 2 * 65536 * 8 : nullptr;
#if DZ_CONV3_DMA
    using C3 = ConvDmaCfg<9, 9, 64, 3, 1, 7, 7, 64, (DZ_CONV3_DMA / 100) % 10, (DZ_CONV3_DMA / 10) % 10, DZ_CONV3_DMA % 10, DZ_CONV3_DMA / 1000 ? DZ_CONV3_DMA / 1000 : 1>;
    if (C3::fits(B)) {
      ConvDmaParams q;
      for (int g = 0; g < DZ_MAX_GROUPS; ++g) {
        q.in[g] = (const float*)p.in[g < G ? g : 0]; q.in_img_base[g] = p.in_img_base[g < G ? g : 0];
        q.w[g] = p.w[g < G ? g : 0]; q.bias[g] = p.bias[g < G ? g : 0];
      }
      q.out = T.feat; q.B = B; q.G = G; q.dbg = p.dbg;
      rc = dz_launch_conv_dma_fwd<C3, DZ_CONV_DMA_OCC>(q, s);
    } else
#endif
    rc = launch_conv_fwd<Conv3Fwd>(p, 64, G, B, s);
    if (rc) return rc;
    DZ_PROF(s, "conv3_fwd");
  }
  return DZ_OK;
}
#if DZ_CONV1_DMA
using Conv1Dma = Conv1DmaCfg<DZ_CONV1_DMA % 10, (DZ_CONV1_DMA / 10) % 10, (DZ_CONV1_DMA / 100) % 10 ? (DZ_CONV1_DMA / 100) % 10 : 4,
                             DZ_CONV1_DMA / 1000>;   // BRING * 1000 + AHEAD * 100 + NRG * 10 + NRB
inline Conv1DmaParams torso_conv1_dma_params(const TorsoBufs& T, int G, int B, const float* const* prm,
                                             const uint8_t* const* in) {
  Conv1DmaParams q;
  for (int g = 0; g < DZ_MAX_GROUPS; ++g) {
    const int gg = g < G ? g : 0;
    q.in[g] = in[gg]; q.w[g] = prm[gg] + T.conv_w[0]; q.bias[g] = prm[gg] + T.conv_b[0];
  }
  q.out = T.act1; q.B = B; q.G = G;
  return q;
}
#endif
inline int torso_forward(const TorsoBufs& T, int G, int B, const float* const* prm,
                         const uint8_t* const* in, hipStream_t s,
                         const NoiseParams* side = nullptr, long long* dbg = nullptr,
                         const SeamClear* clr = nullptr) {
#if DZ_CONV1_DMA
  if (B > 8 && Conv1Dma::fits(B)) {
    Conv1DmaParams q = torso_conv1_dma_params(T, G, B, prm, in);
    q.dbg = dbg;
    int rc;
    if (clr) {
      StepPre sp;
      sp.noise = side ? *side : NoiseParams{};
      sp.noise_blocks = side ? (unsigned)((side->n + 255) / 256) : 0u;
      sp.clr = *clr;
      rc = dz_launch_conv1_dma<Conv1Dma, StepPreSide>(q, sp, sp.noise_blocks + clr->blocks, s);
    } else if (side) {
      rc = dz_launch_conv1_dma<Conv1Dma, NoiseSide>(q, *side, (unsigned)((side->n + 255) / 256), s);
    } else {
      rc = dz_launch_conv1_dma<Conv1Dma, DzNoSide>(q, DzNoSide::Params{0}, 0u, s);
    }
    if (rc) return rc;
    DZ_PROF(s, side ? "conv1_fwd+noise" : "conv1_fwd");
    return torso_forward_rest(T, G, B, prm, s, dbg);
  }
#endif
  const ConvFwdParams p = torso_conv1_params(T, G, B, prm, in, dbg);
  const int rc = B <= 8 ? launch_conv1_fwd<Conv1FwdAct>(p, G, B, side, s, clr)
                        : launch_conv1_fwd<Conv1Fwd>(p, G, B, side, s, clr);
  if (rc) return rc;
  DZ_PROF(s, side ? "conv1_fwd+noise" : "conv1_fwd");
  return torso_forward_rest(T, G, B, prm, s, dbg);
}
// The same with any side job that conv1 does not depend on as `side_blocks` extra workgroups
// of the conv1 launch (the IQN steps' cosine table: dz_iqn.hip).
template <class Side>
inline int torso_forward_side(const TorsoBufs& T, int G, int B, const float* const* prm,
                              const uint8_t* const* in, hipStream_t s,
                              const typename Side::Params& sp, unsigned side_blocks) {
#if DZ_CONV1_DMA
  if (B > 8 && Conv1Dma::fits(B)) {
    const Conv1DmaParams q = torso_conv1_dma_params(T, G, B, prm, in);
    const int rc1 = dz_launch_conv1_dma<Conv1Dma, Side>(q, sp, side_blocks, s);
    if (rc1) return rc1;
    DZ_PROF(s, "conv1_fwd+side");
    return torso_forward_rest(T, G, B, prm, s, nullptr);
  }
#endif
  const ConvFwdParams p = torso_conv1_params(T, G, B, prm, in, nullptr);
  int rc;
  if (B <= 8)
    rc = dz_launch_gemm_side<Conv1FwdAct, Side>(
        p, dim3(32 / Conv1FwdAct::BN, G * Conv1FwdAct::tiles_per_group(B), 1), sp, side_blocks, s);
  else
    rc = dz_launch_gemm_side<Conv1Fwd, Side>(
        p, dim3(32 / Conv1Fwd::BN, G * Conv1Fwd::tiles_per_group(B), 1), sp, side_blocks, s);
  if (rc) return rc;
  DZ_PROF(s, "conv1_fwd+side");
  return torso_forward_rest(T, G, B, prm, s, nullptr);
}

// The backward chain's CRITICAL PATH is dfeat -> dact2 -> dact1: the input gradients.  The weight
// gradients are leaves (nothing of the backward pass reads them), so they need not share their own
// layer's launch.  DZ_CONV_BWD_SPLIT, same box, us per launch:
//   0  [w3 + d3] [w2 + d2] [w1]          8.1 + 9.5 + 7.9  = 25.4   (rounds 2-5: each layer's pair)
//   1  [d3] [d2] [w1 + w2 + w3]          7.7 + 6.1 + 13.4 = 27.2
//   2  [w3 + d3] [d2] [w1 + w2]          8.1 + 6.1 + 10.2 = 24.3   <- conv2's weight gradient rides
//      with conv1's (the chain's last launch, 250 + 243 workgroups); conv3's stays (it costs its
//      launch 0.4 us).
#ifndef DZ_CONV_BWD_SPLIT
#define DZ_CONV_BWD_SPLIT 2
#endif
struct ConvWgDeferred { bool on3 = false, on2 = false; ConvWgradParams w3, w2; };

// conv1's weight gradient (register-staged skeleton), conv2's and conv3's (LDS-DMA Ops) in ONE launch
template <class W1, class W2, class W3, int OCC>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(OCC, OCC)))
void dz_conv_wgrad3_kernel(ConvWgradParams p1, dim3 g1, ConvWgradParams p2, dim3 g2, ConvWgradParams p3, dim3 g3,
                           PrioUpdateParams prio) {
  constexpr int A1 = DzGemmSmem<W1>::ELEMS, A2 = DzDmaOpShape<W2>::LDS_FLOATS, A3 = DzDmaOpShape<W3>::LDS_FLOATS;
  constexpr int LDS = A1 > A2 ? (A1 > A3 ? A1 : A3) : (A2 > A3 ? A2 : A3);
  __shared__ __attribute__((aligned(1024))) float lds[LDS];
  const unsigned n1 = g1.x * g1.y * g1.z, n2 = g2.x * g2.y * g2.z, n3 = g3.x * g3.y * g3.z;
  if (blockIdx.x < n1) dz_gemm_body<W1>(p1, dz_unflatten(blockIdx.x, g1), lds);
  else if (blockIdx.x < n1 + n2) dz_dmaop_body<W2>(p2, dz_unflatten(blockIdx.x - n1, g2), lds);
  else if (blockIdx.x < n1 + n2 + n3) dz_dmaop_body<W3>(p3, dz_unflatten(blockIdx.x - n1 - n2, g3), lds);
  else PrioUpdateSideFast::run(prio, 0, lds, (int)sizeof(lds));   // (optional last block: the sum-tree write-back)
}

// conv3 / conv2 backward: weight (+ bias) gradient slabs fused with the layer's input gradient.
// `prio`: optional sum-tree priority write-back carried as one extra block.  `defer` (nullable):
// the weight gradient is left for launch_conv1_wgrad (DZ_CONV_BWD_SPLIT).
inline int launch_conv3_bwd(const ConvWgradParams& w, const ConvDgradParams& d, int B, hipStream_t s,
                            const PrioUpdateParams* prio = nullptr, ConvWgDeferred* defer = nullptr) {
#if DZ_CONV3_BWD_DMA
  using Wg = ConvWgDmaOp<9, 9, 64, 3, 1, 7, 7, 64, (DZ_CONV3_BWD_DMA / 10) % 10, DZ_CONV3_BWD_DMA % 10>;
  using Dg = ConvDgDmaOp<9, 9, 64, 3, 1, 7, 7, 64, DZ_CONV3_BWD_DMA / 1000, (DZ_CONV3_BWD_DMA / 100) % 10, DZ_CONV3_BWD_DMA % 10>;
  static_assert(Wg::KROWS == Conv3Wg::KROWS, "slab layout");
  if (Dg::fits(B)) {
    const dim3 gw(1, Wg::MT, w.S), gd(64 / Dg::BN, Dg::tiles(B), 1);
#if DZ_CONV_BWD_SPLIT == 1 && DZ_CONV2_BWD_DMA
    if (defer && !prio) {
      defer->on3 = true; defer->w3 = w;
      return dz_launch_dmaop<Dg, DZ_CONV_BWD_OCC>(d, gd, s);
    }
#endif
    if (prio) return dz_launch_dmaop2_side<Wg, Dg, PrioUpdateSideFast, DZ_CONV_BWD_OCC>(w, gw, d, gd, *prio, 1, s);
    return dz_launch_dmaop2<Wg, Dg, DZ_CONV_BWD_OCC>(w, gw, d, gd, s);
  }
#endif
  const dim3 gw(64 / Conv3Wg::BN, Conv3Wg::MT, w.S), gd(64 / Conv3Dg::BN, Conv3Dg::tiles(B), 1);
  if (prio) return dz_launch_gemm2_side<Conv3Wg, Conv3Dg, PrioUpdateSideFast>(w, gw, d, gd, *prio, 1, s);
  return dz_launch_gemm2<Conv3Wg, Conv3Dg>(w, gw, d, gd, s);
}
inline int launch_conv2_bwd(const ConvWgradParams& w, const ConvDgradParams& d, int B, hipStream_t s,
                            ConvWgDeferred* defer = nullptr) {
#if DZ_CONV2_BWD_DMA
  using Wg = ConvWgDmaOp<20, 20, 32, 4, 2, 9, 9, 64, (DZ_CONV2_BWD_DMA / 10) % 10, DZ_CONV2_BWD_DMA % 10>;
  using Dg = ConvDgDmaOp<20, 20, 32, 4, 2, 9, 9, 64, 1, (DZ_CONV2_BWD_DMA / 100) % 10, DZ_CONV2_BWD_DMA % 10>;
  static_assert(Wg::KROWS == Conv2Wg::KROWS, "slab layout");
#ifdef DZ_BWD_ABL   // timing ablations (results are wrong): 1 = weight gradient only, 2 = input gradient only
  if (DZ_BWD_ABL == 1) return dz_launch_dmaop<Wg, DZ_CONV_BWD_OCC>(w, dim3(1, Wg::MT, w.S), s);
  if (DZ_BWD_ABL == 2) return dz_launch_dmaop<Dg, DZ_CONV_BWD_OCC>(d, dim3(1, Dg::tiles(B), 4), s);
#endif
#if DZ_CONV_BWD_SPLIT && DZ_CONV3_BWD_DMA
  if (Dg::fits(B) && defer && (defer->on3 || DZ_CONV_BWD_SPLIT == 2)) {   // (1: conv3's is waiting already)
    defer->on2 = true; defer->w2 = w;
    return dz_launch_dmaop<Dg, DZ_CONV_BWD_OCC>(d, dim3(1, Dg::tiles(B), 4), s);
  }
#endif
  if (Dg::fits(B))
    return dz_launch_dmaop2<Wg, Dg, DZ_CONV_BWD_OCC>(w, dim3(1, Wg::MT, w.S), d, dim3(1, Dg::tiles(B), 4), s);
#endif
#ifdef DZ_BWD_ABL
  if (DZ_BWD_ABL == 3) return dz_launch_gemm<Conv2Wg>(w, dim3(64 / Conv2Wg::BN, Conv2Wg::MT, w.S), s);
  if (DZ_BWD_ABL == 4) return dz_launch_gemm<Conv2Dg>(d, dim3(32 / Conv2Dg::BN, Conv2Dg::tiles(B), 4), s);
#endif
  return dz_launch_gemm2<Conv2Wg, Conv2Dg>(w, dim3(64 / Conv2Wg::BN, Conv2Wg::MT, w.S), d,
                                          dim3(32 / Conv2Dg::BN, Conv2Dg::tiles(B), 4), s);
}

inline int launch_conv1_wgrad(const ConvWgradParams& p, hipStream_t s, const ConvWgDeferred* defer = nullptr,
                              const PrioUpdateParams* prio = nullptr) {
#if DZ_CONV_BWD_SPLIT && DZ_CONV3_BWD_DMA && DZ_CONV2_BWD_DMA
  if (defer && defer->on2) {
    using W3 = ConvWgDmaOp<9, 9, 64, 3, 1, 7, 7, 64, (DZ_CONV3_BWD_DMA / 10) % 10, DZ_CONV3_BWD_DMA % 10>;
    using W2 = ConvWgDmaOp<20, 20, 32, 4, 2, 9, 9, 64, (DZ_CONV2_BWD_DMA / 10) % 10, DZ_CONV2_BWD_DMA % 10>;
    const dim3 g1(32 / Conv1Wg::BN, Conv1Wg::MT, p.S), g2(1, W2::MT, defer->w2.S),
        g3(1, W3::MT, defer->on3 ? defer->w3.S : 0);
    hipLaunchKernelGGL((dz_conv_wgrad3_kernel<Conv1Wg, W2, W3, DZ_CONV_BWD_OCC>),
                       dim3(dz_count(g1) + dz_count(g2) + dz_count(g3) + (prio ? 1 : 0)), dim3(256), 0, s, p, g1,
                       defer->w2, g2, defer->on3 ? defer->w3 : defer->w2, g3, prio ? *prio : PrioUpdateParams{});
    DZ_LAUNCH_CHECK();
    return DZ_OK;
  }
#endif
#if DZ_CONV1_WG_DMA
  using Wg = Conv1WgDmaOp<DZ_CONV1_WG_DMA / 10, DZ_CONV1_WG_DMA % 10>;
  static_assert(Wg::KROWS == Conv1Wg::KROWS, "slab layout");
  return dz_launch_dmaop<Wg, DZ_CONV_BWD_OCC>(p, dim3(1, Wg::MT, p.S), s);
#else
  return dz_launch_gemm<Conv1Wg>(p, dim3(32 / Conv1Wg::BN, Conv1Wg::MT, p.S), s);
#endif
}

inline int64_t torso_wgrad_part_elems() {
  return (int64_t)kS_cw1 * Conv1Wg::KROWS * 32 + (int64_t)kS_cw2 * Conv2Wg::KROWS * 64 +
         (int64_t)kS_cw3 * Conv3Wg::KROWS * 64;
}

// Backward of group 0 from dfeat (already masked by feat > 0): split-K partial
// slabs of the three weight(+bias-row) gradients are left in `part`; the three
// ReduceJobs that fold them into the gradient buffer are returned in `jobs`.
// `prio`: optional sum-tree priority write-back carried by the conv3 launch as one
// extra block (it depends only on the loss kernel's priorities).
inline int torso_backward(const TorsoBufs& T, int B, const float* online,
                          const uint8_t* s_tm1, const float* dfeat, float* dact2,
                          float* dact1, float* part, float* grad, ReduceJob* jobs,
                          hipStream_t s, const PrioUpdateParams* prio = nullptr) {
  int rc;
  ConvWgDeferred defer;
  float* part1 = part;
  float* part2 = part1 + (long)kS_cw1 * Conv1Wg::KROWS * 32;
  float* part3 = part2 + (long)kS_cw2 * Conv2Wg::KROWS * 64;
  {
    ConvWgradParams w;
    w.in = T.act2; w.dy = dfeat; w.part = part3; w.B = B; w.S = kS_cw3;
    ConvDgradParams d;
    d.dy = dfeat; d.w = online + T.conv_w[2]; d.act = T.act2; d.dx = dact2; d.B = B;
    rc = launch_conv3_bwd(w, d, B, s, prio, &defer);
    if (rc) return rc;
    DZ_PROF(s, "conv3_wgrad+dgrad");
  }
  {
    ConvWgradParams w;
    w.in = T.act1; w.dy = dact2; w.part = part2; w.B = B; w.S = kS_cw2;
    ConvDgradParams d;
    d.dy = dact2; d.w = online + T.conv_w[1]; d.act = T.act1; d.dx = dact1; d.B = B;
    rc = launch_conv2_bwd(w, d, B, s, &defer);
    if (rc) return rc;
    DZ_PROF(s, "conv2_wgrad+dgrad");
  }
  {
    ConvWgradParams p;
    p.in = s_tm1; p.dy = dact1; p.part = part1; p.B = B; p.S = kS_cw1;
    rc = launch_conv1_wgrad(p, s, &defer);
    if (rc) return rc;
    DZ_PROF(s, defer.on3 ? "conv_wgrads3" : defer.on2 ? "conv_wgrads" : "conv1_wgrad");
  }
  jobs[0] = {part1, kS_cw1, (long)Conv1Wg::KROWS * 32, grad + T.conv_w[0]};
  jobs[1] = {part2, kS_cw2, (long)Conv2Wg::KROWS * 64, grad + T.conv_w[1]};
  jobs[2] = {part3, kS_cw3, (long)Conv3Wg::KROWS * 64, grad + T.conv_w[2]};
  return DZ_OK;
}

}  // namespace
