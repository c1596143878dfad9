// Implicit-GEMM operand views ("Ops") for dz_mfma_gemm: the convolutions and
// linear layers of the DQN-family networks (ref: dqn_zoo/networks.py:82-221),
// forward and backward.  Activations are NHWC, conv weights HWIO flattened to
// [kh*kw*cin][cout], linear weights [in][out] -- the reference's own layouts
// (networks_test.py:44,53), so weights are always a row-major [K][N] matrix.
#pragma once

#include "dz_gemm.h"

#define DZ_MAX_GROUPS 3

// --------------------------------------------------------------------------- //
//  Convolution forward: out[img,oh,ow,:] = relu(sum_k patch(img,oh,ow,k) W[k,:] + b)
//  GEMM rows are output pixels (per group), columns output channels.
// --------------------------------------------------------------------------- //
struct ConvFwdParams {
  const void* in[DZ_MAX_GROUPS];  // per group: u8 or f32 [images][H][W][C]
  int in_img_base[DZ_MAX_GROUPS]; // first input image of each group in in[g]
  const float* w[DZ_MAX_GROUPS];  // [K][CO]
  const float* bias[DZ_MAX_GROUPS];
  float* out;                     // [G*B][OH][OW][CO]
  int B;                          // images per group
  int G;
};

template <int IN_U8, int H, int W, int C, int KS, int S, int OH, int OW, int CO,
          int WM_, int WN_, int WK_>
struct ConvFwdOp {
  static constexpr int WM = WM_, WN = WN_, WK = WK_;
  static constexpr int A_LAYOUT = DZ_KC, B_LAYOUT = DZ_RC;
  static constexpr int A_MAP = IN_U8 ? DZ_MAP_ROW16 : DZ_MAP_QUAD;
  static constexpr int BM = 32 * WM, BN = 32 * WN, BK = 16 * WK;
  static constexpr int K = KS * KS * C;
  static_assert(K % BK == 0, "K must be a multiple of the stage depth");
  static_assert(IN_U8 ? (KS * C == 32) : (C % 16 == 0), "chunk must not straddle taps");
  typedef ConvFwdParams Params;

  static int tiles_per_group(int B) { return (B * OH * OW + BM - 1) / BM; }

  __device__ static bool tile(const Params& p, DzTile& t) {
    const int tpg = (p.B * OH * OW + BM - 1) / BM;
    t.z = blockIdx.y / tpg;
    t.m0 = (blockIdx.y % tpg) * BM;
    t.n0 = blockIdx.x * BN;
    t.st_begin = 0;
    t.st_end = K / BK;
    return t.z < p.G;
  }
  // pixel index (within group) -> element offset of input pixel (oh*S, ow*S, 0)
  __device__ static bool pixel_base(const Params& p, const DzTile& t, int row,
                                    long& off) {
    const int ml = t.m0 + row;
    if (ml >= p.B * OH * OW) return false;
    const int img = ml / (OH * OW), pix = ml % (OH * OW);
    const int oh = pix / OW, ow = pix % OW;
    off = (((long)(p.in_img_base[t.z] + img) * H + oh * S) * W + ow * S) * C;
    return true;
  }
  __device__ static float4 load_a(const Params& p, const DzTile& t, int st, int c,
                                  int row, int q) {
    long off;
    if (!pixel_base(p, t, row, off)) return dz_f4zero();
    const int k0 = st * BK + c * 16 + 4 * q;
    const int tap = k0 / C, ci = k0 % C;
    const int kh = tap / KS, kw = tap % KS;
    return *(const float4*)((const float*)p.in[t.z] + off + ((long)kh * W + kw) * C + ci);
  }
  __device__ static void load_a16(const Params& p, const DzTile& t, int st, int c,
                                  int row, float4 (&v)[4]) {
    long off;
    if (!pixel_base(p, t, row, off)) return;
    const int k0 = st * BK + c * 16;  // KS*C == 32 bytes per kernel row
    const int kh = k0 / 32, o = k0 % 32;
    const uint4 raw = *(const uint4*)((const uint8_t*)p.in[t.z] + off + (long)kh * W * C + o);
    const unsigned wds[4] = {raw.x, raw.y, raw.z, raw.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      // networks.py:193: x.astype(float32) / 255.0 (a true division).
      v[i] = dz_f4((float)(wds[i] & 0xff) / 255.0f, (float)((wds[i] >> 8) & 0xff) / 255.0f,
                   (float)((wds[i] >> 16) & 0xff) / 255.0f, (float)(wds[i] >> 24) / 255.0f);
    }
  }
  __device__ static float4 load_b(const Params& p, const DzTile& t, int st, int c,
                                  int kk, int rq) {
    const int k = st * BK + c * 16 + kk;
    return *(const float4*)(p.w[t.z] + (long)k * CO + t.n0 + 4 * rq);
  }
  __device__ static void store(const Params& p, const DzTile& t, int wm, int wn,
                               int lane, const f32x16& acc) {
    const int col = t.n0 + wn * 32 + (lane & 31);
    const float b = p.bias[t.z][col];
    const int rows = p.B * OH * OW;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int ml = t.m0 + wm * 32 + dz_acc_row(r, lane);
      if (ml < rows) {
        const float v = acc[r] + b;
        p.out[((long)t.z * rows + ml) * CO + col] = v > 0.f ? v : 0.f;
      }
    }
  }
};

// --------------------------------------------------------------------------- //
//  Linear layers.  A "head" is one hk.Linear / noisy_linear; several heads that
//  share the input batch are launched together (z = (group, head, split)).
//  Noisy form (networks.py:168-176):
//     y = x Wmu + bmu + ((x . eps_in) Wsig + bsig) . eps_out
//  is evaluated as ONE contraction of depth 2K: [x | x.eps_in] [Wmu ; Wsig.eps_out].
// --------------------------------------------------------------------------- //
struct FcHead {
  long w_mu;     // offset of [K][ldw] matrix in the parameter buffer
  long w_sig;    // (noisy only)
  int ldw;
  int N;
  int K;         // multiple of 16
  int x_off;     // column offset of this head's input in x
  int eps_in;    // offsets into the group's noise block (noisy only)
  int eps_out;
  int out_off;   // column offset in the output row
};

struct FcFwdParams {
  const float* x;  // [G*M][ldx]
  int ldx;
  int M;           // rows per group (batch)
  int G;
  int NH;
  int S;           // grid split-K factor
  int noisy;
  const float* params[DZ_MAX_GROUPS];
  const float* noise[DZ_MAX_GROUPS];
  FcHead head[2];
  float* part;     // [S][G*M][ldo]
  int ldo;
};

template <int WM_, int WN_, int WK_>
struct FcFwdOp {
  static constexpr int WM = WM_, WN = WN_, WK = WK_;
  static constexpr int A_LAYOUT = DZ_KC, B_LAYOUT = DZ_RC, A_MAP = DZ_MAP_QUAD;
  static constexpr int BM = 32 * WM, BN = 32 * WN, BK = 16 * WK;
  typedef FcFwdParams Params;

  __device__ static bool tile(const Params& p, DzTile& t) {
    const int split = blockIdx.z % p.S;
    const int gh = blockIdx.z / p.S;
    const int h = gh % p.NH, g = gh / p.NH;
    const FcHead& hd = p.head[h];
    t.z = g; t.z2 = h | (split << 8);
    t.m0 = blockIdx.y * BM;
    t.n0 = blockIdx.x * BN;
    const int chunks = (hd.K / 16) * (p.noisy ? 2 : 1);
    const int stages = (chunks + WK - 1) / WK;
    const int per = (stages + p.S - 1) / p.S;
    t.st_begin = split * per;
    t.st_end = min(stages, t.st_begin + per);
    return g < p.G && t.n0 < hd.N && t.m0 < p.M;
  }
  __device__ static float4 load_a(const Params& p, const DzTile& t, int st, int c,
                                  int row, int q) {
    const FcHead& hd = p.head[t.z2 & 0xff];
    const int kc = hd.K / 16;
    int gc = st * WK + c;
    const int m = t.m0 + row;
    if (m >= p.M || gc >= kc * (p.noisy ? 2 : 1)) return dz_f4zero();
    const bool sig = gc >= kc;
    if (sig) gc -= kc;
    const int k = gc * 16 + 4 * q;
    float4 v = *(const float4*)(p.x + (long)(t.z * p.M + m) * p.ldx + hd.x_off + k);
    if (sig) v = dz_mul4(v, *(const float4*)(p.noise[t.z] + hd.eps_in + k));
    return v;
  }
  __device__ static float4 load_b(const Params& p, const DzTile& t, int st, int c,
                                  int kk, int rq) {
    const FcHead& hd = p.head[t.z2 & 0xff];
    const int kc = hd.K / 16;
    int gc = st * WK + c;
    if (gc >= kc * (p.noisy ? 2 : 1)) return dz_f4zero();
    const bool sig = gc >= kc;
    if (sig) gc -= kc;
    const int k = gc * 16 + kk;
    const int n = t.n0 + 4 * rq;
    const float* wrow = p.params[t.z] + (sig ? hd.w_sig : hd.w_mu) + (long)k * hd.ldw;
    float4 v = dz_load4_masked(wrow, n, hd.N);
    if (sig) v = dz_mul4(v, dz_load4_masked(p.noise[t.z] + hd.eps_out, n, hd.N));
    return v;
  }
  __device__ static void store(const Params& p, const DzTile& t, int wm, int wn,
                               int lane, const f32x16& acc) {
    const FcHead& hd = p.head[t.z2 & 0xff];
    const int split = t.z2 >> 8;
    const int col = t.n0 + wn * 32 + (lane & 31);
    if (col >= hd.N) return;
    float* base = p.part + ((long)split * p.G * p.M + (long)t.z * p.M) * p.ldo +
                  hd.out_off + col;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = t.m0 + wm * 32 + dz_acc_row(r, lane);
      if (m < p.M) base[(long)m * p.ldo] = acc[r];
    }
  }
};

// dX[m][x_off+k] = sum_n dY[m][n] Wmu[k][n] + eps_in[k] sum_n dY[m][n] eps_out[n] Wsig[k][n]
// accumulated over every head that reads the same input columns (fc1: adv1 and
// val1 both read the torso features).  Reduction index = (head, mu|sigma, n).
struct FcDgradParams {
  const float* dy;  // [M][ldy]
  int ldy;
  int M;
  int NH;           // heads summed into the same output columns
  int S;
  int noisy;
  const float* params;
  const float* noise;
  FcHead head[2];   // out_off = column offset of the head's dY
  float* part;      // [S][M][ldo]
  int ldo;
  int K;            // output columns (= heads' K)
  int x_off;        // output column offset
};

template <int WM_, int WN_, int WK_>
struct FcDgradOp {
  static constexpr int WM = WM_, WN = WN_, WK = WK_;
  static constexpr int A_LAYOUT = DZ_KC, B_LAYOUT = DZ_KC, A_MAP = DZ_MAP_QUAD;
  static constexpr int BM = 32 * WM, BN = 32 * WN, BK = 16 * WK;
  typedef FcDgradParams Params;

  __device__ static int chunks_per_part(const FcHead& hd) { return (hd.N + 15) / 16; }
  // global chunk -> (head, sigma?, n0)
  __device__ static bool locate(const Params& p, int gc, int& h, bool& sig, int& n0) {
    for (h = 0; h < p.NH; ++h) {
      const int cp = chunks_per_part(p.head[h]);
      const int tot = cp * (p.noisy ? 2 : 1);
      if (gc < tot) {
        sig = gc >= cp;
        n0 = (sig ? gc - cp : gc) * 16;
        return true;
      }
      gc -= tot;
    }
    return false;
  }
  __device__ static bool tile(const Params& p, DzTile& t) {
    int chunks = 0;
    for (int h = 0; h < p.NH; ++h)
      chunks += chunks_per_part(p.head[h]) * (p.noisy ? 2 : 1);
    const int stages = (chunks + WK - 1) / WK;
    const int per = (stages + p.S - 1) / p.S;
    t.z = blockIdx.z;  // split
    t.m0 = blockIdx.y * BM;
    t.n0 = blockIdx.x * BN;
    t.st_begin = t.z * per;
    t.st_end = min(stages, t.st_begin + per);
    return t.n0 < p.K && t.m0 < p.M;
  }
  __device__ static float4 load_a(const Params& p, const DzTile& t, int st, int c,
                                  int row, int q) {
    int h, n0; bool sig;
    const int m = t.m0 + row;
    if (m >= p.M || !locate(p, st * WK + c, h, sig, n0)) return dz_f4zero();
    const FcHead& hd = p.head[h];
    const int n = n0 + 4 * q;
    float4 v = dz_load4_masked(p.dy + (long)m * p.ldy + hd.out_off, n, hd.N);
    if (sig) v = dz_mul4(v, dz_load4_masked(p.noise + hd.eps_out, n, hd.N));
    return v;
  }
  // B tile row = output column k; 4 consecutive reduction indices n.
  __device__ static float4 load_b(const Params& p, const DzTile& t, int st, int c,
                                  int row, int q) {
    int h, n0; bool sig;
    const int k = t.n0 + row;
    if (k >= p.K || !locate(p, st * WK + c, h, sig, n0)) return dz_f4zero();
    const FcHead& hd = p.head[h];
    const int n = n0 + 4 * q;
    float4 v = dz_load4_masked(p.params + (sig ? hd.w_sig : hd.w_mu) + (long)k * hd.ldw,
                               n, hd.N);
    if (sig) v = dz_scale4(v, p.noise[hd.eps_in + k]);
    return v;
  }
  __device__ static void store(const Params& p, const DzTile& t, int wm, int wn,
                               int lane, const f32x16& acc) {
    const int col = t.n0 + wn * 32 + (lane & 31);
    if (col >= p.K) return;
    float* base = p.part + (long)t.z * p.M * p.ldo + p.x_off + col;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = t.m0 + wm * 32 + dz_acc_row(r, lane);
      if (m < p.M) base[(long)m * p.ldo] = acc[r];
    }
  }
};

// dWmu[k][n] = sum_m x[m][x_off+k] dY[m][out_off+n];  dWsig = dWmu eps_in[k] eps_out[n]
struct FcWgradParams {
  const float* x;
  int ldx;
  const float* dy;
  int ldy;
  int M;            // batch rows (reduction)
  int NH;
  int noisy;
  const float* noise;
  FcHead head[2];
  float* grad;      // gradient buffer with the parameter layout
};

template <int WM_, int WN_, int WK_>
struct FcWgradOp {
  static constexpr int WM = WM_, WN = WN_, WK = WK_;
  static constexpr int A_LAYOUT = DZ_RC, B_LAYOUT = DZ_RC, A_MAP = DZ_MAP_QUAD;
  static constexpr int BM = 32 * WM, BN = 32 * WN, BK = 16 * WK;
  typedef FcWgradParams Params;

  __device__ static bool tile(const Params& p, DzTile& t) {
    t.z = blockIdx.z;  // head
    const FcHead& hd = p.head[t.z];
    t.m0 = blockIdx.y * BM;  // k rows
    t.n0 = blockIdx.x * BN;
    t.st_begin = 0;
    t.st_end = (p.M + BK - 1) / BK;
    return t.z < p.NH && t.m0 < hd.K && t.n0 < hd.N;
  }
  __device__ static float4 load_a(const Params& p, const DzTile& t, int st, int c,
                                  int kk, int rq) {
    const FcHead& hd = p.head[t.z];
    const int m = st * BK + c * 16 + kk;
    const int k = t.m0 + 4 * rq;
    if (m >= p.M || k >= hd.K) return dz_f4zero();
    return *(const float4*)(p.x + (long)m * p.ldx + hd.x_off + k);
  }
  __device__ static float4 load_b(const Params& p, const DzTile& t, int st, int c,
                                  int kk, int rq) {
    const FcHead& hd = p.head[t.z];
    const int m = st * BK + c * 16 + kk;
    if (m >= p.M) return dz_f4zero();
    return dz_load4_masked(p.dy + (long)m * p.ldy + hd.out_off, t.n0 + 4 * rq, hd.N);
  }
  __device__ static void store(const Params& p, const DzTile& t, int wm, int wn,
                               int lane, const f32x16& acc) {
    const FcHead& hd = p.head[t.z];
    const int col = t.n0 + wn * 32 + (lane & 31);
    if (col >= hd.N) return;
    const float eo = p.noisy ? p.noise[hd.eps_out + col] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int k = t.m0 + wm * 32 + dz_acc_row(r, lane);
      if (k < hd.K) {
        p.grad[hd.w_mu + (long)k * hd.ldw + col] = acc[r];
        if (p.noisy)
          p.grad[hd.w_sig + (long)k * hd.ldw + col] = acc[r] * (p.noise[hd.eps_in + k] * eo);
      }
    }
  }
};

// --------------------------------------------------------------------------- //
//  Convolution weight gradient: dW[k][co] = sum_pixels patch(pixel,k) dY[pixel][co]
//  rows = k (contiguous inside a kernel row), reduction = output pixels (split).
// --------------------------------------------------------------------------- //
struct ConvWgradParams {
  const void* in;   // layer input, u8 or f32 [B][H][W][C]
  const float* dy;  // [B*OH*OW][CO]
  float* part;      // [S][K][CO]
  int B;
  int S;
};

template <int IN_U8, int H, int W, int C, int KS, int S, int OH, int OW, int CO,
          int WM_, int WN_, int WK_>
struct ConvWgradOp {
  static constexpr int WM = WM_, WN = WN_, WK = WK_;
  static constexpr int A_LAYOUT = DZ_RC, B_LAYOUT = DZ_RC, A_MAP = DZ_MAP_QUAD;
  static constexpr int BM = 32 * WM, BN = 32 * WN, BK = 16 * WK;
  static constexpr int K = KS * KS * C;
  static_assert(K % BM == 0 && CO % BN == 0 && C % 4 == 0, "tile shape");
  typedef ConvWgradParams Params;

  __device__ static bool tile(const Params& p, DzTile& t) {
    const int stages = (p.B * OH * OW + BK - 1) / BK;
    const int per = (stages + p.S - 1) / p.S;
    t.z = blockIdx.z;
    t.m0 = blockIdx.y * BM;
    t.n0 = blockIdx.x * BN;
    t.st_begin = t.z * per;
    t.st_end = min(stages, t.st_begin + per);
    return true;
  }
  __device__ static float4 load_a(const Params& p, const DzTile& t, int st, int c,
                                  int kk, int rq) {
    const int ml = st * BK + c * 16 + kk;
    if (ml >= p.B * OH * OW) return dz_f4zero();
    const int img = ml / (OH * OW), pix = ml % (OH * OW);
    const int oh = pix / OW, ow = pix % OW;
    const int k = t.m0 + 4 * rq;
    const int tap = k / C, ci = k % C;
    const int kh = tap / KS, kw = tap % KS;
    const long off = (((long)img * H + oh * S + kh) * W + ow * S + kw) * C + ci;
    if (IN_U8) {
      const unsigned wd = *(const unsigned*)((const uint8_t*)p.in + off);
      return dz_f4((float)(wd & 0xff) / 255.0f, (float)((wd >> 8) & 0xff) / 255.0f,
                   (float)((wd >> 16) & 0xff) / 255.0f, (float)(wd >> 24) / 255.0f);
    }
    return *(const float4*)((const float*)p.in + off);
  }
  __device__ static float4 load_b(const Params& p, const DzTile& t, int st, int c,
                                  int kk, int rq) {
    const int ml = st * BK + c * 16 + kk;
    if (ml >= p.B * OH * OW) return dz_f4zero();
    return *(const float4*)(p.dy + (long)ml * CO + t.n0 + 4 * rq);
  }
  __device__ static void store(const Params& p, const DzTile& t, int wm, int wn,
                               int lane, const f32x16& acc) {
    const int col = t.n0 + wn * 32 + (lane & 31);
    float* base = p.part + (long)t.z * K * CO + col;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int k = t.m0 + wm * 32 + dz_acc_row(r, lane);
      base[(long)k * CO] = acc[r];
    }
  }
};

// --------------------------------------------------------------------------- //
//  Convolution input gradient (transposed convolution in gather form):
//    dX[img,h,w,ci] = sum_{kh,kw,co} dY[img,(h-kh)/S,(w-kw)/S,co] W[kh,kw,ci,co]
//  Input pixels are processed per stride-parity class (z = (h%S)*S + w%S) so
//  that every class has exactly (KS/S)^2 candidate taps; the result is masked
//  with the ReLU of the layer that produced the input (act > 0).
// --------------------------------------------------------------------------- //
struct ConvDgradParams {
  const float* dy;   // [B][OH][OW][CO]
  const float* w;    // [KS*KS*C][CO]
  const float* act;  // [B][H][W][C] post-ReLU input activation (mask)
  float* dx;         // [B][H][W][C]
  int B;
};

template <int H, int W, int C, int KS, int S, int OH, int OW, int CO,
          int WM_, int WN_, int WK_>
struct ConvDgradOp {
  static constexpr int WM = WM_, WN = WN_, WK = WK_;
  static constexpr int A_LAYOUT = DZ_KC, B_LAYOUT = DZ_KC, A_MAP = DZ_MAP_QUAD;
  static constexpr int BM = 32 * WM, BN = 32 * WN, BK = 16 * WK;
  static constexpr int TS = (KS + S - 1) / S;   // taps per dimension per class
  static constexpr int HP = (H + S - 1) / S, WP = (W + S - 1) / S;  // class grid
  static constexpr int RED = TS * TS * CO;
  static_assert(CO % 16 == 0 && RED % BK == 0 && C % BN == 0, "tile shape");
  static_assert(H % S == 0 && W % S == 0, "every parity class has the same size");
  typedef ConvDgradParams Params;

  static int tiles(int B) { return (B * HP * WP + BM - 1) / BM; }

  __device__ static bool tile(const Params& p, DzTile& t) {
    t.z = blockIdx.z;  // parity class
    t.m0 = blockIdx.y * BM;
    t.n0 = blockIdx.x * BN;
    t.st_begin = 0;
    t.st_end = RED / BK;
    return true;
  }
  __device__ static bool pixel(const Params& p, const DzTile& t, int row, int& img,
                               int& h, int& w) {
    const int ml = t.m0 + row;
    if (ml >= p.B * HP * WP) return false;
    img = ml / (HP * WP);
    const int pix = ml % (HP * WP);
    h = (pix / WP) * S + t.z / S;
    w = (pix % WP) * S + t.z % S;
    return true;
  }
  __device__ static float4 load_a(const Params& p, const DzTile& t, int st, int c,
                                  int row, int q) {
    int img, h, w;
    if (!pixel(p, t, row, img, h, w)) return dz_f4zero();
    const int r0 = st * BK + c * 16 + 4 * q;
    const int tap = r0 / CO, co = r0 % CO;
    const int kh = (t.z / S) + (tap / TS) * S, kw = (t.z % S) + (tap % TS) * S;
    const int oh = (h - kh) / S, ow = (w - kw) / S;  // exact by construction
    if (kh >= KS || kw >= KS || h < kh || w < kw || oh >= OH || ow >= OW)
      return dz_f4zero();
    return *(const float4*)(p.dy + (((long)img * OH + oh) * OW + ow) * CO + co);
  }
  __device__ static float4 load_b(const Params& p, const DzTile& t, int st, int c,
                                  int row, int q) {
    const int ci = t.n0 + row;
    const int r0 = st * BK + c * 16 + 4 * q;
    const int tap = r0 / CO, co = r0 % CO;
    const int kh = (t.z / S) + (tap / TS) * S, kw = (t.z % S) + (tap % TS) * S;
    if (kh >= KS || kw >= KS) return dz_f4zero();
    return *(const float4*)(p.w + ((long)(kh * KS + kw) * C + ci) * CO + co);
  }
  __device__ static void store(const Params& p, const DzTile& t, int wm, int wn,
                               int lane, const f32x16& acc) {
    const int ci = t.n0 + wn * 32 + (lane & 31);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      int img, h, w;
      if (pixel(p, t, wm * 32 + dz_acc_row(r, lane), img, h, w)) {
        const long o = (((long)img * H + h) * W + w) * C + ci;
        p.dx[o] = p.act[o] > 0.f ? acc[r] : 0.f;
      }
    }
  }
};
