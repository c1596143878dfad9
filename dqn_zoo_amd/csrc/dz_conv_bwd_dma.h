// The backward pass of the float-input convolutions (conv2, conv3) as Ops of the LDS-DMA skeleton
// (dz_dma_op.h); same contractions, slab layouts and masks as ConvWgradOp / ConvDgradOp
// (dz_qnet_ops.h), which stay for batch sizes whose pixel count is not a whole number of tiles.
//   weight + bias gradient  dW[k][co] = sum_pixels patch(pixel, k) dY[pixel][co]  (+ the row k == K
//       whose patch value is 1: the bias gradient), split over the pixels: part[S][K + 1][CO]
//   input gradient          dX[img, h, w, ci] = relu'(act) sum_{kh, kw, co} dY[img, (h - kh) / S, (w - kw) / S, co] W[kh, kw, ci, co]
//       per stride-parity class z = (h % S) S + w % S, (KS / S)^2 candidate taps each
// ref: jax.grad through hk.Conv2D (networks.py:194-198), rainbow/agent.py:112-118.
#pragma once

#include "dz_dma_op.h"
#include "dz_qnet_ops.h"

namespace {

struct DzNoPre {};

// ---- weight (+ bias) gradient: tile = 64 k-rows x 64 output channels, depth = pixels ----
// Both operands are "output-contiguous": a pixel's 64 consecutive k (inside one kernel row of the
// NHWC patch) and its 64 output-channel gradients are 256 contiguous bytes each.
template <int H, int W, int C, int KS, int S, int OH, int OW, int CO, int KT_, int NBUF_>
struct ConvWgDmaOp {
  static constexpr int MI = 1, NI = 1, SUBM = 2, SUBN = 2, WKD = 1, KT = KT_, NBUF = NBUF_;
  static constexpr bool A_KC = false, B_KC = false;
  static constexpr int BM = 64, BN = 64, BK = 16 * KT_;
  static constexpr int K = KS * KS * C, KROWS = K + 1, ROWLEN = KS * C, MT = K / BM + 1;
  static_assert(CO == 64 && K % BM == 0 && ROWLEN % BM == 0, "tile shape");
  typedef ConvWgradParams Params;
  struct Tile { int nst, m0, z, st0, rows; long koff; };
  typedef DzNoPre Pre;

  __device__ static bool tile(const Params& p, const dim3& bid, Tile& t) {
    t.rows = p.B * OH * OW;
    const int stages = (t.rows + BK - 1) / BK;
    const int per = (stages + p.S - 1) / p.S;
    t.z = bid.z; t.m0 = bid.y * BM;
    t.st0 = t.z * per;
    t.nst = max(min(stages, t.st0 + per) - t.st0, 0);
    t.koff = (long)(t.m0 / ROWLEN) * (W * C) + (t.m0 % ROWLEN);
    return true;
  }
  __device__ static const float* a_src(const Params& p, const Tile& t, int st, int r, int u) {
    const int ml = (t.st0 + st) * BK + r;
    const int mc = min(ml, t.rows - 1);
    const int img = mc / (OH * OW), pix = mc - img * (OH * OW);
    const int oh = pix / OW, ow = pix - oh * OW;
    const float* src = (const float*)p.in + (((long)img * H + oh * S) * W + ow * S) * C + t.koff + 4 * u;
    // the bias tile (m0 == K): row K reads 1, the 63 rows behind it 0
    const float* one = u == 0 ? (const float*)dz_page_one : (const float*)dz_page_zero;
    src = dz_val(t.m0 < K, src, one);
    return dz_val(ml < t.rows, src, (const float*)dz_page_zero);
  }
  __device__ static const float* b_src(const Params& p, const Tile& t, int st, int r, int u) {
    const int ml = (t.st0 + st) * BK + r;
    const float* src = p.dy + (long)min(ml, t.rows - 1) * CO + 4 * u;
    return dz_val(ml < t.rows, src, (const float*)dz_page_zero);
  }
  __device__ static Pre prefetch(const Params&, const Tile&, int, int, int, unsigned) { return Pre{}; }
  __device__ static void store(const Params& p, const Tile& t, int bi, int bj, int lane, const f32x16& acc,
                               unsigned, const Pre&) {
    float* base = p.part + (long)t.z * KROWS * CO + bj * 32 + (lane & 31);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int k = t.m0 + bi * 32 + dz_acc_row(r, lane);
      if (k < KROWS) base[(long)k * CO] = acc[r];
    }
  }
};

// ---- conv1's weight (+ bias) gradient straight from the uint8 frames: tile = 64 k-rows (two
// kernel rows of 8 pixels x 4 channels = 2 x 32 contiguous BYTES per output pixel) x 32 output
// channels, depth = output pixels; two depth halves per stage meet in LDS.
template <int KT_, int NBUF_>
struct Conv1WgDmaOp {
  static constexpr int H = 84, W = 84, C = 4, KS = 8, S = 4, OH = 20, OW = 20, CO = 32;
  static constexpr int MI = 1, NI = 1, SUBM = 2, SUBN = 1, WKD = 2, KT = KT_, NBUF = NBUF_;
  static constexpr bool A_KC = false, B_KC = false;
  static constexpr int A_U8 = 1;
  static constexpr int BM = 64, BN = 32, BK = 16 * WKD * KT_;
  static constexpr int K = 256, KROWS = K + 1, MT = K / BM + 1;
  typedef ConvWgradParams Params;
  struct Tile { int nst, m0, z, st0, rows; };
  typedef DzNoPre Pre;

  __device__ static bool tile(const Params& p, const dim3& bid, Tile& t) {
    t.rows = p.B * OH * OW;
    const int stages = (t.rows + BK - 1) / BK;
    const int per = (stages + p.S - 1) / p.S;
    t.z = bid.z; t.m0 = bid.y * BM;
    t.st0 = t.z * per;
    t.nst = max(min(stages, t.st0 + per) - t.st0, 0);
    return true;
  }
  __device__ static const float* a_src(const Params& p, const Tile& t, int st, int r, int u) {
    const int ml = (t.st0 + st) * BK + r;
    const int mc = min(ml, t.rows - 1);
    const int img = mc / (OH * OW), pix = mc - img * (OH * OW);
    const int oh = pix / OW, ow = pix - oh * OW;
    const int kh = min(t.m0 / 32 + (u >> 1), KS - 1);
    const unsigned char* src = (const unsigned char*)p.in + (((long)img * H + oh * S + kh) * W + ow * S) * C + 16 * (u & 1);
    const unsigned char* one = u == 0 ? (const unsigned char*)dz_page_u8one : (const unsigned char*)dz_page_zero;
    src = dz_val(t.m0 < K, src, one);
    return (const float*)dz_val(ml < t.rows, src, (const unsigned char*)dz_page_zero);
  }
  __device__ static const float* b_src(const Params& p, const Tile& t, int st, int r, int u) {
    const int ml = (t.st0 + st) * BK + r;
    const float* src = p.dy + (long)min(ml, t.rows - 1) * CO + 4 * u;
    return dz_val(ml < t.rows, src, (const float*)dz_page_zero);
  }
  __device__ static Pre prefetch(const Params&, const Tile&, int, int, int, unsigned) { return Pre{}; }
  __device__ static void store(const Params& p, const Tile& t, int bi, int bj, int lane, const f32x16& acc,
                               unsigned rmask, const Pre&) {
    float* base = p.part + (long)t.z * KROWS * CO + bj * 32 + (lane & 31);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int k = t.m0 + bi * 32 + dz_acc_row(r, lane);
      if (((rmask >> r) & 1u) && k < KROWS) base[(long)k * CO] = acc[r];
    }
  }
};

// ---- input gradient: tile = 32 input pixels of one parity class x 32 SUBN input channels ----
// depth = (tap, co): one tap (CO = 64 output-channel gradients, 256 contiguous bytes) per stage.
template <int H, int W, int C, int KS, int S, int OH, int OW, int CO, int SUBN_, int KT_, int NBUF_>
struct ConvDgDmaOp {
  static constexpr int MI = 1, NI = 1, SUBM = 1, SUBN = SUBN_, WKD = 4 / SUBN_, KT = KT_, NBUF = NBUF_;
  static constexpr bool A_KC = true, B_KC = true;
  static constexpr int BM = 32, BN = 32 * SUBN_, BK = 16 * WKD * KT_;
  static constexpr int TS = (KS + S - 1) / S, HP = H / S, WP = W / S, RED = TS * TS * CO;
  static constexpr bool SPLIT = WKD > 1;
  static constexpr int NPRE = 16 / WKD;   // accumulator registers a storing wave finishes (WKD = 2: wave 0 all 16)
  static_assert(CO % BK == 0 && C % BN == 0 && H % S == 0 && W % S == 0 && KS % S == 0, "tile shape");
  typedef ConvDgradParams Params;
  struct Tile { int nst, m0, n0, z; };
  struct Pre { unsigned o[16 / WKD]; float mk[16 / WKD]; };

  static int tiles(int B) { return B * HP * WP / BM; }
  static bool fits(int B) { return (B * HP * WP) % BM == 0; }

  __device__ static bool tile(const Params&, const dim3& bid, Tile& t) {
    t.z = bid.z; t.m0 = bid.y * BM; t.n0 = bid.x * BN; t.nst = RED / BK;
    return true;
  }
  __device__ static void pixel(const Tile& t, int row, int& img, int& h, int& w) {
    const int m = t.m0 + row;
    img = m / (HP * WP);
    const int pix = m - img * (HP * WP);
    const int ph = pix / WP;
    h = ph * S + t.z / S;
    w = (pix - ph * WP) * S + t.z % S;
  }
  __device__ static void tap(const Tile& t, int st, int u, int& kh, int& kw, int& co) {
    const int r0 = st * BK + 4 * u;
    const int tp = r0 / CO;
    co = r0 - tp * CO;
    kh = (t.z / S) + (tp / TS) * S; kw = (t.z % S) + (tp % TS) * S;   // < KS
  }
  __device__ static const float* a_src(const Params& p, const Tile& t, int st, int r, int u) {
    int img, h, w, kh, kw, co;
    pixel(t, r, img, h, w);
    tap(t, st, u, kh, kw, co);
    const int oh = (h - kh) / S, ow = (w - kw) / S;   // exact when h >= kh, w >= kw
    const bool ok = (h >= kh) & (w >= kw) & (oh < OH) & (ow < OW);
    const int ohc = min(max(oh, 0), OH - 1), owc = min(max(ow, 0), OW - 1);
    const float* src = p.dy + (((long)img * OH + ohc) * OW + owc) * CO + co;
    return dz_val(ok, src, (const float*)dz_page_zero);
  }
  __device__ static const float* b_src(const Params& p, const Tile& t, int st, int r, int u) {
    int kh, kw, co;
    tap(t, st, u, kh, kw, co);
    return p.w + ((long)(kh * KS + kw) * C + t.n0 + r) * CO + co;
  }
  // the ReLU mask values (and addresses) of the rows this wave stores, requested before the DMAs
  __device__ static Pre prefetch(const Params& p, const Tile& t, int bi, int bj, int lane, unsigned rmask) {
    Pre pre;
    const int ci = t.n0 + bj * 32 + (lane & 31);
    const int wk = rmask == 0xffffu ? 0 : (31 - __builtin_clz(rmask)) / NPRE;   // (wave-uniform)
#pragma unroll
    for (int j = 0; j < NPRE; ++j) {
      const int i = wk * NPRE + j;
      const int row = bi * 32 + (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5);
      int img, h, w;
      pixel(t, row, img, h, w);
      pre.o[j] = (unsigned)(((img * H + h) * W + w) * C + ci);
      pre.mk[j] = p.act[pre.o[j]];
    }
    return pre;
  }
  __device__ static void store(const Params& p, const Tile&, int, int, int, const f32x16& acc, unsigned rmask,
                               const Pre& pre) {
#pragma unroll
    for (int i = 0; i < 16; ++i)
      if ((rmask >> i) & 1u) p.dx[pre.o[i % NPRE]] = pre.mk[i % NPRE] > 0.f ? acc[i] : 0.f;
  }
};

}  // namespace
