// Implicit-GEMM operand views ("Ops") for dz_mfma_gemm: the convolutions and
// linear layers of the DQN-family networks (ref: dqn_zoo/networks.py:82-221),
// forward and backward.  Activations are NHWC, conv weights HWIO flattened to
// [kh*kw*cin][cout], linear weights [in][out] -- the reference's own layouts
// (networks_test.py:44,53), so weights are always a row-major [K][ld] matrix.
//
// LOADER RULE: every load_a/load_b is BRANCH-FREE.  Out-of-range rows, taps or
// reduction indices are handled by clamping the address into the buffer and
// zeroing the value with a select afterwards.  A conditional load makes hipcc
// wrap it in an exec-mask branch with its own `s_waitcnt vmcnt(0)`, which
// serialises the 8-12 loads of a stage into 8-12 dependent memory round trips
// (measured: 385 s_and_saveexec / 40 vmcnt waits in the fc1 kernel, 65 us
// instead of ~15 us).  All operand rows are 16-byte aligned by construction
// (leading dimensions are multiples of 4 floats), so loads are always float4.
//
// THIRD LOADER RULE (round 4): nothing in a loader may CONSUME a loaded value.  A select
// on the value (`ok ? v : 0`), a conversion or a product placed behind the load is
// scheduled right behind it, in front of the stage's MFMA block, with the `s_waitcnt
// vmcnt` it needs: the stage then pays the full global-load latency before its first MFMA
// (conv3 forward ISA: six `vmcnt` waits + 24 `v_cndmask` between the loads and the MFMAs;
// the conv weight-gradient halves waited for ALL their loads).  Masked slots therefore
// select on the ADDRESS -- they read dz_page_zero / dz_page_one (dz_gemm.h) -- and
// conversions happen when the stage is written to LDS (DzRaw16 / DzRaw4).
//
// SECOND LOADER RULE: loaders never index a kernel-argument array with a runtime
// value (p.in[g], p.head[h] ...).  The compiler turns that into a LOAD of the
// pointer from the kernarg segment followed by s_waitcnt vmcnt(0) in front of
// every operand load -- two dependent memory round trips per stage (seen in the
// conv1 ISA).  tile() resolves the group's pointers and scalars once into the
// Op's Tile with static-index selects (dz_pick3), so they live in SGPRs.
#pragma once

#include "dz_gemm.h"

#define DZ_MAX_GROUPS 3

__device__ __forceinline__ float4 dz_ld4(const float* p) { return *(const float4*)p; }
__device__ __forceinline__ float4 dz_sel4(bool ok, float4 v) {
  return dz_f4(ok ? v.x : 0.f, ok ? v.y : 0.f, ok ? v.z : 0.f, ok ? v.w : 0.f);
}
// c ? e : (1,1,1,1): a divergent `c ? v * e : v` becomes an exec-masked block with
// its own vmcnt(0) wait; v * (c ? e : 1) is branch-free and exact (x * 1 == x)
__device__ __forceinline__ float4 dz_one_or4(bool c, float4 e) {
  return dz_f4(c ? e.x : 1.f, c ? e.y : 1.f, c ? e.z : 1.f, c ? e.w : 1.f);
}
// c ? a : b on VALUES (see FcDgradOp::locate)
template <class T>
__device__ __forceinline__ T dz_val(bool c, T a, T b) { return c ? a : b; }
// keep v[j] iff i+j < n
__device__ __forceinline__ float4 dz_mask4(float4 v, int i, int n) {
  return dz_f4(i < n ? v.x : 0.f, i + 1 < n ? v.y : 0.f, i + 2 < n ? v.z : 0.f,
               i + 3 < n ? v.w : 0.f);
}
// x / 255.0f for x in [0, 255], bit-identical to the IEEE division
// (networks.py:193 `x.astype(float32) / 255.0`) in 3 instructions instead of the
// ~11 of v_div_*: y = x * fl(1/255) and one Newton correction with FMAs; checked
// exhaustively for all 256 inputs (tests/test_abi.py::test_div255_identity).
__device__ __forceinline__ float dz_div255(float x) {
  const float rc = 1.0f / 255.0f;
  const float y = x * rc;
  const float r = __builtin_fmaf(-y, 255.0f, x);
  return __builtin_fmaf(r, rc, y);
}
__device__ __forceinline__ float4 dz_u8x4_to_unit(unsigned w) {
  return dz_f4(dz_div255((float)(w & 0xff)), dz_div255((float)((w >> 8) & 0xff)),
               dz_div255((float)((w >> 16) & 0xff)), dz_div255((float)(w >> 24)));
}
// Wave-wide reductions on the DPP crossbar.  `__shfl_xor` compiles to ds_bpermute_b32 +
// s_waitcnt lgkmcnt(0): six dependent LDS round trips per reduction (54 of them on
// the critical path of the Rainbow loss kernel).  Here: two quad permutes, row_half_mirror
// and row_mirror leave every lane of a 16-lane row with the row's total (4 VALU ops), the
// four row totals are read with v_readlane and added in a fixed order.  Every lane gets
// the same bits; all 64 lanes must be active.
template <int CTRL>
__device__ __forceinline__ float dz_dpp(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(
      0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float dz_lane(float v, int l) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l));
}
__device__ __forceinline__ float dz_wave_sum(float v) {
  v += dz_dpp<0xB1>(v);    // quad_perm [1,0,3,2]
  v += dz_dpp<0x4E>(v);    // quad_perm [2,3,0,1]
  v += dz_dpp<0x141>(v);   // row_half_mirror
  v += dz_dpp<0x140>(v);   // row_mirror
  return (dz_lane(v, 0) + dz_lane(v, 16)) + (dz_lane(v, 32) + dz_lane(v, 48));
}
__device__ __forceinline__ float dz_wave_max(float v) {
  v = fmaxf(v, dz_dpp<0xB1>(v));
  v = fmaxf(v, dz_dpp<0x4E>(v));
  v = fmaxf(v, dz_dpp<0x141>(v));
  v = fmaxf(v, dz_dpp<0x140>(v));
  return fmaxf(fmaxf(dz_lane(v, 0), dz_lane(v, 16)), fmaxf(dz_lane(v, 32), dz_lane(v, 48)));
}
// arr[g] for g in [0, 3) with static indices only (see the second loader rule).
template <class T>
__device__ __forceinline__ T dz_pick3(const T* arr, int g) {
  T v = arr[0];
  v = g == 1 ? arr[1] : v;
  v = g == 2 ? arr[2] : v;
  return v;
}

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
  long long* dbg = nullptr;       // (DZ_GEMM_STAMPS builds) per-workgroup wall-clock stamps
};

// AM_ (all three convolution Ops): 1 = masked slots select on the ADDRESS (third loader rule)
// and the next stage's loads are pinned in front of the MFMA block; 0 = the round-1..3 form
// (select on the loaded value, scheduler's order).  Chosen per instantiation by measurement
// (same box, us): conv1 fwd 14.0 -> 13.5, conv2 fwd 10.95 -> 10.15, conv3 bwd 10.1 -> 9.4,
// conv1 wgrad 8.7 -> 7.6 with 1; conv3 fwd 10.9 -> 11.2 and conv2 bwd 10.7 -> 11.8 with 1
// (those two keep 0): with two or three workgroups per CU a wave that waits for its loads
// leaves the matrix pipe to its neighbours, so the exposed wait is not always the loss the
// ISA suggests.
template <int IN_U8, int H, int W, int C, int KS, int S, int OH, int OW, int CO,
          int WM_, int WN_, int WK_, int KT_ = 1, int AM_ = 1>
struct ConvFwdOp {
  static constexpr int WM = WM_, WN = WN_, WK = WK_, KT = KT_, CPS = WK_ * KT_;
  static constexpr int PIN_LOADS = AM_;
  static constexpr int HAS_DBG = 1;
  static constexpr int A_LAYOUT = DZ_KC, B_LAYOUT = DZ_RC;
  static constexpr int A_MAP = IN_U8 ? DZ_MAP_ROW16 : DZ_MAP_QUAD;
  static constexpr int BM = 32 * WM, BN = 32 * WN, BK = 16 * CPS;
  static constexpr int K = KS * KS * C;
  static_assert(K % BK == 0, "K must be a multiple of the stage depth");
  static_assert(IN_U8 ? (KS * C == 32) : (C % 16 == 0), "chunk must not straddle taps");
  static_assert(CO % BN == 0, "column tiles are full");
  typedef ConvFwdParams Params;

  struct Tile : DzTile { const void* in; const float* w; const float* bias; int img_base; };

  static int tiles_per_group(int B) { return (B * OH * OW + BM - 1) / BM; }

  __device__ static bool tile(const Params& p, const dim3& bid, Tile& t) {
    const int tpg = (p.B * OH * OW + BM - 1) / BM;
    t.z = bid.y / tpg;
    t.m0 = (bid.y % tpg) * BM;
    t.n0 = bid.x * BN;
    t.st_begin = 0;
    t.st_end = K / BK;
    t.in = dz_pick3(p.in, t.z); t.w = dz_pick3(p.w, t.z); t.bias = dz_pick3(p.bias, t.z);
    t.img_base = dz_pick3(p.in_img_base, t.z);
    return t.z < p.G;
  }
  // pixel index (within group, clamped) -> element offset of input pixel
  // (oh*S, ow*S, 0); returns whether the row is real.
  __device__ static bool pixel_base(const Params& p, const Tile& t, int row,
                                    long& off) {
    const int rows = p.B * OH * OW;
    const int ml = t.m0 + row;
    const int mc = min(ml, rows - 1);
    const int img = mc / (OH * OW), pix = mc % (OH * OW);
    const int oh = pix / OW, ow = pix % OW;
    off = (((long)(t.img_base + img) * H + oh * S) * W + ow * S) * C;
    return ml < rows;
  }
  __device__ static float4 load_a(const Params& p, const Tile& t, int st, int c,
                                  int row, int q) {
    long off;
    const bool ok = pixel_base(p, t, row, off);
    const int k0 = st * BK + c * 16 + 4 * q;
    const int tap = k0 / C, ci = k0 % C;
    const int kh = tap / KS, kw = tap % KS;
    const float* src = (const float*)t.in + off + ((long)kh * W + kw) * C + ci;
    if constexpr (AM_) return dz_ld4(ok ? src : dz_page_zero);
    else return dz_sel4(ok, dz_ld4(src));
  }
  // deferred conversion (dz_gemm.h DzRaw16): the loader keeps the 16 raw bytes
  static constexpr int A_RAW16 = IN_U8;
  __device__ static uint4 load_a16_raw(const Params& p, const Tile& t, int st, int c, int row) {
    long off;
    const bool ok = pixel_base(p, t, row, off);
    const int k0 = st * BK + c * 16;  // KS*C == 32 bytes per kernel row
    const int kh = k0 / 32, o = k0 % 32;
    const uint8_t* src = (const uint8_t*)t.in + off + (long)kh * W * C + o;
    return *(const uint4*)(ok ? src : (const uint8_t*)dz_page_zero);
  }
  __device__ static void cook16(uint4 raw, float4 (&v)[4]) {  // byte 0 -> 0/255 == 0.0f
    v[0] = dz_u8x4_to_unit(raw.x); v[1] = dz_u8x4_to_unit(raw.y);
    v[2] = dz_u8x4_to_unit(raw.z); v[3] = dz_u8x4_to_unit(raw.w);
  }
  __device__ static void load_a16(const Params& p, const Tile& t, int st, int c,
                                  int row, float4 (&v)[4]) {
    long off;
    const bool ok = pixel_base(p, t, row, off);
    const int k0 = st * BK + c * 16;  // KS*C == 32 bytes per kernel row
    const int kh = k0 / 32, o = k0 % 32;
    const uint4 raw = *(const uint4*)((const uint8_t*)t.in + off + (long)kh * W * C + o);
    v[0] = dz_sel4(ok, dz_u8x4_to_unit(raw.x));
    v[1] = dz_sel4(ok, dz_u8x4_to_unit(raw.y));
    v[2] = dz_sel4(ok, dz_u8x4_to_unit(raw.z));
    v[3] = dz_sel4(ok, dz_u8x4_to_unit(raw.w));
  }
  __device__ static float4 load_b(const Params& p, const Tile& t, int st, int c,
                                  int kk, int rq) {
    const int k = st * BK + c * 16 + kk;
    return dz_ld4(t.w + (long)k * CO + t.n0 + 4 * rq);
  }
  static constexpr int SPLIT_STORE = 1;
  // the bias of this lane's output column, requested in the prologue (dz_gemm.h DzHasPre)
  struct Pre { float bias; };
  __device__ static Pre prefetch(const Params& p, const Tile& t, int wm, int wn, int lane,
                                 unsigned rmask) {
    return Pre{t.bias[t.n0 + wn * 32 + (lane & 31)]};
  }
  __device__ static void store(const Params& p, const Tile& t, int wm, int wn,
                               int lane, const f32x16& acc, unsigned rmask, const Pre& pre) {
    const int col = t.n0 + wn * 32 + (lane & 31);
    float b = pre.bias;
    // pinned as "available" once, here: used first inside the conditional store blocks below,
    // the compiler puts an `s_waitcnt vmcnt(0)` into every one of them, and from the second
    // block on that wait is for the previous block's STORE (gfx950 counts stores in vmcnt):
    // the wave's four row stores became four serial round trips
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(b));
    const int rows = p.B * OH * OW;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int ml = t.m0 + wm * 32 + dz_acc_row(r, lane);
      if (((rmask >> r) & 1u) && ml < rows) {
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
//  Weight rows have pitch ldw (multiple of 4, >= N): columns in [N, ldw) are zero
//  padding; a tile may read columns beyond N (clamped to the pitch), whose
//  products land in output columns that are never stored.
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
  int out_off;   // column offset in the output row (multiple of 4)
};

// head[h] for h in {0, 1}, field by field (scalar selects, no memory indexing).
__device__ __forceinline__ FcHead dz_pick_head(const FcHead* hd, int h) {
  FcHead r;
  const bool b = h != 0;
  r.w_mu = b ? hd[1].w_mu : hd[0].w_mu; r.w_sig = b ? hd[1].w_sig : hd[0].w_sig;
  r.ldw = b ? hd[1].ldw : hd[0].ldw; r.N = b ? hd[1].N : hd[0].N;
  r.K = b ? hd[1].K : hd[0].K; r.x_off = b ? hd[1].x_off : hd[0].x_off;
  r.eps_in = b ? hd[1].eps_in : hd[0].eps_in; r.eps_out = b ? hd[1].eps_out : hd[0].eps_out;
  r.out_off = b ? hd[1].out_off : hd[0].out_off;
  return r;
}

// An upstream split-K input gradient consumed WITHOUT its reduction launch: the
// layer's dY operand is read as the sum of S partial slabs times the ReLU mask of
// the activation it flows through,
//     dY[m][c] = (act[m][c] > 0) * (part[0][m][c] + part[1][m][c] + ...)   (in that order)
// by the loaders of BOTH contractions that consume it (weight gradient and input
// gradient: same order, same bits).  S is a template parameter of the consuming
// Ops: a run-time switch in a loader splits it into basic blocks that each wait
// for their own loads.  Saves a ~4.7 us launch per layer (reduce_parts_kernel).
struct DyParts {
  const float* part = nullptr;  // [S][M][ld]
  long stride = 0;              // floats between slabs
  const float* mask = nullptr;  // [M][ld] post-ReLU activation
  float* out = nullptr;         // optional [M][ld]: one consumer materialises dY (bias sums)
};
template <int S>
__device__ __forceinline__ float4 dz_dy_parts4(const DyParts& q, long o) {
  float4 v = dz_ld4(q.part + o);
  const float4 mk = dz_ld4(q.mask + o);
  float4 r[S > 1 ? S - 1 : 1];
#pragma unroll
  for (int i = 1; i < S; ++i) r[i - 1] = dz_ld4(q.part + i * q.stride + o);
#pragma unroll
  for (int i = 1; i < S; ++i) { v.x += r[i - 1].x; v.y += r[i - 1].y; v.z += r[i - 1].z; v.w += r[i - 1].w; }
  return dz_f4(mk.x > 0.f ? v.x : 0.f, mk.y > 0.f ? v.y : 0.f, mk.z > 0.f ? v.z : 0.f,
               mk.w > 0.f ? v.w : 0.f);
}

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

// NZ_ >= 0 fixes Params::noisy at compile time: the run-time `noisy == 2` test in
// the loader splits it into basic blocks, and each block waits for its own loads
// before the next block issues any (one round trip per block instead of one per
// stage).
template <int WM_, int WN_, int WK_, int KT_ = 1, int NZ_ = -1>
struct FcFwdOp {
  __device__ static int nzy(const FcFwdParams& p) { return NZ_ >= 0 ? NZ_ : p.noisy; }
  static constexpr int WM = WM_, WN = WN_, WK = WK_, KT = KT_, CPS = WK_ * KT_;
  static constexpr int A_LAYOUT = DZ_KC, B_LAYOUT = DZ_RC, A_MAP = DZ_MAP_QUAD;
  static constexpr int BM = 32 * WM, BN = 32 * WN, BK = 16 * CPS;
  typedef FcFwdParams Params;

  struct Tile : DzTile { FcHead hd; const float* prm; const float* nz; };

  __device__ static bool tile(const Params& p, const dim3& bid, Tile& t) {
    const int split = bid.z % p.S;
    const int gh = bid.z / p.S;
    const int h = gh % p.NH, g = gh / p.NH;
    t.hd = dz_pick_head(p.head, h);
    t.prm = dz_pick3(p.params, g); t.nz = dz_pick3(p.noise, g);
    const FcHead& hd = t.hd;
    t.z = g; t.z2 = h | (split << 8);
    t.m0 = bid.y * BM;
    t.n0 = bid.x * BN;
    // noisy == 1: depth 2K over [x | x.eps_in] [Wmu ; Wsig.eps_out];
    // noisy == 2: depth K against W_eff = Wmu + Wsig (eps_in (x) eps_out), built in load_b
    const int chunks = (hd.K / 16) * (nzy(p) == 1 ? 2 : 1);
    const int stages = (chunks + CPS - 1) / CPS;
    const int per = (stages + p.S - 1) / p.S;
    t.st_begin = split * per;
    t.st_end = min(stages, t.st_begin + per);
    return g < p.G && t.n0 < hd.N && t.m0 < p.M;
  }
  __device__ static float4 load_a(const Params& p, const Tile& t, int st, int c,
                                  int row, int q) {
    const FcHead& hd = t.hd;
    const int kc = hd.K / 16, total = kc * (nzy(p) == 1 ? 2 : 1);
    const int gc = st * CPS + c;
    const int m = t.m0 + row;
    const bool ok = (m < p.M) & (gc < total);
    const int gcc = min(gc, total - 1);
    const bool sig = gcc >= kc;
    const int k = (gcc - (sig ? kc : 0)) * 16 + 4 * q;
    const float4 v = dz_ld4(p.x + (long)(t.z * p.M + min(m, p.M - 1)) * p.ldx + hd.x_off + k);
    const float4 e = dz_ld4(t.nz + hd.eps_in + k);  // L2-resident, tiny
    return dz_sel4(ok, dz_mul4(v, dz_one_or4(sig, e)));
  }
  __device__ static float4 load_b(const Params& p, const Tile& t, int st, int c,
                                  int kk, int rq) {
    const FcHead& hd = t.hd;
    const int kc = hd.K / 16, total = kc * (nzy(p) == 1 ? 2 : 1);
    const int gc = st * CPS + c;
    const bool ok = gc < total;
    const int gcc = min(gc, total - 1);
    const bool sig = gcc >= kc;
    const int k = (gcc - (sig ? kc : 0)) * 16 + kk;
    const int n = min(t.n0 + 4 * rq, hd.ldw - 4);
    const float4 v = dz_ld4(t.prm + (sig ? hd.w_sig : hd.w_mu) + (long)k * hd.ldw + n);
    const float4 e = dz_ld4(t.nz + hd.eps_out + n);
    if (nzy(p) == 2) {
      const float4 sg = dz_ld4(t.prm + hd.w_sig + (long)k * hd.ldw + n);
      const float ei = t.nz[hd.eps_in + k];
      return dz_sel4(ok, dz_f4(__builtin_fmaf(sg.x, ei * e.x, v.x), __builtin_fmaf(sg.y, ei * e.y, v.y),
                               __builtin_fmaf(sg.z, ei * e.z, v.z), __builtin_fmaf(sg.w, ei * e.w, v.w)));
    }
    return dz_sel4(ok, dz_mul4(v, dz_one_or4(sig, e)));
  }
  static constexpr int SPLIT_STORE = 0;  // distributed epilogue measured slower for this Op (EXPERIMENTS.md)
  __device__ static void store(const Params& p, const Tile& t, int wm, int wn,
                               int lane, const f32x16& acc, unsigned rmask = 0xffffu) {
    const FcHead& hd = t.hd;
    const int split = t.z2 >> 8;
    const int col = t.n0 + wn * 32 + (lane & 31);
    if (col >= hd.N) return;
    float* base = p.part + ((long)split * p.G * p.M + (long)t.z * p.M) * p.ldo +
                  hd.out_off + col;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = t.m0 + wm * 32 + dz_acc_row(r, lane);
      if (((rmask >> r) & 1u) && m < p.M) base[(long)m * p.ldo] = acc[r];
    }
  }
};

// dX[m][x_off+k] = sum_n dY[m][n] Wmu[k][n] + eps_in[k] sum_n dY[m][n] eps_out[n] Wsig[k][n]
// accumulated over every head that reads the same input columns (fc1: adv1 and
// val1 both read the torso features).  Reduction index = (head, mu|sigma, n).
// dY values with n >= N are zeroed, so weight columns beyond N never matter.
struct FcDgradParams {
  const float* dy;  // [M][ldy]
  int ldy;
  int M;
  int NH;           // heads summed into the same output columns (1 or 2)
  int S;
  int noisy;
  const float* params;
  const float* noise;
  FcHead head[2];   // out_off = column offset of the head's dY
  float* part;      // [S][M][ldo]
  int ldo;
  int K;            // output columns (= heads' K)
  int x_off;        // output column offset
  // S == 1 only: ReLU mask of the layer input fused into the store
  // (out = mask[m][col] > 0 ? acc : 0), saving the separate masking pass
  const float* relu_mask = nullptr;
  DyParts dyp;      // DYS_ > 0: dY = masked sum of dyp's slabs instead of `dy`
};

template <int WM_, int WN_, int WK_, int KT_ = 1, int MI_ = 1, int NI_ = 1, int NZ_ = -1,
          int DYS_ = 0>
struct FcDgradOp {
  __device__ __forceinline__ static int nzy(const FcDgradParams& p) { return NZ_ >= 0 ? NZ_ : p.noisy; }  // see FcFwdOp
  static constexpr int WM = WM_, WN = WN_, WK = WK_, KT = KT_, CPS = WK_ * KT_;
  static constexpr int MI = MI_, NI = NI_;
  static constexpr int A_LAYOUT = DZ_KC, B_LAYOUT = DZ_KC, A_MAP = DZ_MAP_QUAD;
  static constexpr int BM = 32 * WM * MI, BN = 32 * WN * NI, BK = 16 * CPS;
  typedef FcDgradParams Params;
  // both heads' descriptors copied into registers once: a per-thread select between
  // two KERNEL-ARGUMENT fields is compiled as a select of their addresses and a
  // vector load from the kernarg segment + vmcnt(0) in front of every operand load
  // (second loader rule; seen in the fc1 backward ISA)
  // (every member function is __forceinline__: inlined before the optimiser runs,
  // the Tile is split into SSA values first; otherwise `h1 ? t.hb.N : t.ha.N` is
  // canonicalised inside the not-yet-inlined helper to a load through a selected
  // ADDRESS, and the Tile then lives in scratch memory)
  struct Tile : DzTile { FcHead ha, hb; int tot0, tot1; };

  struct Loc { int N, ldw, eps_in, eps_out, out_off, n0; long w, w2; bool sig, ok; };

  // noisy == 1: reduction over (mu | sigma) chunks, depth 2N (two-GEMM form);
  // noisy == 2: ONE pass of depth N over the effective weight
  //   W_eff[k][n] = Wmu[k][n] + Wsig[k][n] * (eps_in[k] * eps_out[n])
  // built in the B loader (two loads + 2 VALU per element, half the MFMAs).
  __device__ __forceinline__ static int chunks_of(const Params& p, int h) {
    return (((h ? p.head[1].N : p.head[0].N) + 15) / 16) * (nzy(p) == 1 ? 2 : 1);
  }
  // global chunk -> (head, mu|sigma, first n), arithmetic only (NH <= 2).
  __device__ __forceinline__ static Loc locate(const Tile& t, int gc) {
    const int tot0 = t.tot0, tot1 = t.tot1;
    Loc L;
    L.ok = gc < tot0 + tot1;
    gc = min(gc, tot0 + tot1 - 1);
    const bool h1 = gc >= tot0;
    const FcHead& a = t.ha;
    const FcHead& b = t.hb;  // == head[0] when NH == 1 (callers fill both)
    // dz_val: by-value arguments.  `h1 ? b.N : a.N` on two lvalues is itself an
    // LVALUE in C++ -- a selected address and a load through it, which pins the
    // descriptors in memory (scratch, or the kernarg segment) behind a vmcnt(0).
    L.N = dz_val(h1, b.N, a.N); L.ldw = dz_val(h1, b.ldw, a.ldw);
    L.eps_in = dz_val(h1, b.eps_in, a.eps_in); L.eps_out = dz_val(h1, b.eps_out, a.eps_out);
    L.out_off = dz_val(h1, b.out_off, a.out_off);
    const int gl = gc - (h1 ? tot0 : 0);
    const int cp = (L.N + 15) / 16;
    L.sig = gl >= cp;
    L.n0 = (gl - (L.sig ? cp : 0)) * 16;
    const long w_sig = dz_val(h1, b.w_sig, a.w_sig), w_mu = dz_val(h1, b.w_mu, a.w_mu);
    L.w = L.sig ? w_sig : w_mu;
    L.w2 = w_sig;
    return L;
  }
  __device__ __forceinline__ static bool tile(const Params& p, const dim3& bid, Tile& t) {
    t.ha = p.head[0]; t.hb = p.head[1];
    t.tot0 = chunks_of(p, 0); t.tot1 = p.NH > 1 ? chunks_of(p, 1) : 0;
    const int chunks = t.tot0 + t.tot1;
    const int stages = (chunks + CPS - 1) / CPS;
    const int per = (stages + p.S - 1) / p.S;
    t.z = bid.z;  // split
    t.m0 = bid.y * BM;
    t.n0 = bid.x * BN;
    t.st_begin = t.z * per;
    t.st_end = min(stages, t.st_begin + per);
    return t.n0 < p.K && t.m0 < p.M;
  }
  __device__ __forceinline__ static float4 load_a(const Params& p, const Tile& t, int st, int c,
                                  int row, int q) {
    const Loc L = locate(t, st * CPS + c);
    const int m = t.m0 + row;
    const int n = L.n0 + 4 * q;
    const int nc = min(n, L.ldw - 4);  // dY columns share the weights' padded pitch
    const long o = (long)min(m, p.M - 1) * p.ldy + L.out_off + nc;
    float4 v;
    if constexpr (DYS_ > 0) v = dz_dy_parts4<DYS_>(p.dyp, o);
    else v = dz_ld4(p.dy + o);
    const float4 e = dz_ld4(p.noise + L.eps_out + nc);
    v = dz_mul4(v, dz_one_or4(L.sig, e));
    return dz_mask4(dz_sel4(L.ok & (m < p.M), v), n, L.N);
  }
  // B tile row = output column k; 4 consecutive reduction indices n.
  __device__ __forceinline__ static float4 load_b(const Params& p, const Tile& t, int st, int c,
                                  int row, int q) {
    const Loc L = locate(t, st * CPS + c);
    const int k = min(t.n0 + row, p.K - 1);
    const int nc = min(L.n0 + 4 * q, L.ldw - 4);
    const float4 v = dz_ld4(p.params + L.w + (long)k * L.ldw + nc);
    const float e = p.noise[L.eps_in + k];
    if (nzy(p) == 2) {
      const float4 sg = dz_ld4(p.params + L.w2 + (long)k * L.ldw + nc);
      const float4 eo = dz_ld4(p.noise + L.eps_out + nc);
      return dz_f4(__builtin_fmaf(sg.x, e * eo.x, v.x), __builtin_fmaf(sg.y, e * eo.y, v.y),
                   __builtin_fmaf(sg.z, e * eo.z, v.z), __builtin_fmaf(sg.w, e * eo.w, v.w));
    }
    return dz_scale4(v, L.sig ? e : 1.f);
  }
  static constexpr int SPLIT_STORE = 0;  // distributed epilogue measured slower for this Op (EXPERIMENTS.md)
  __device__ __forceinline__ static void store(const Params& p, const Tile& t, int wm, int wn,
                               int lane, const f32x16& acc, unsigned rmask = 0xffffu) {
    const int col = t.n0 + wn * 32 + (lane & 31);
    if (col >= p.K) return;
    float* base = p.part + (long)t.z * p.M * p.ldo + p.x_off + col;
    if (p.relu_mask) {  // (uniform) all 16 mask values first, then the stores
      const float* mk = p.relu_mask + p.x_off + col;
      float mv[16];
#pragma unroll
      for (int r = 0; r < 16; ++r)
        mv[r] = mk[(long)min(t.m0 + wm * 32 + dz_acc_row(r, lane), p.M - 1) * p.ldo];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = t.m0 + wm * 32 + dz_acc_row(r, lane);
        if (((rmask >> r) & 1u) && m < p.M) base[(long)m * p.ldo] = mv[r] > 0.f ? acc[r] : 0.f;
      }
      return;
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = t.m0 + wm * 32 + dz_acc_row(r, lane);
      if (((rmask >> r) & 1u) && m < p.M) base[(long)m * p.ldo] = acc[r];
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
  // Optional fused global-norm partials: every storing wave writes the sum of
  // squares of the gradient elements it produced (mu and sigma) to
  //   sumsq[((z*sq_ny + y)*sq_nx + x) * WM*WN + wave]   (x,y,z = tile index),
  // tiles outside the problem write zeros, so the slot array is always complete.
  float* sumsq = nullptr;     // null: off
  int sq_nx = 0, sq_ny = 0;   // grid dimensions of this Op's launch
  // noisy only: do not store the sigma-weight gradient (it still enters sumsq);
  // the optimiser re-derives it as dWmu * eps_in (x) eps_out (adam_kernel DerivedGrad)
  int skip_sig_store = 0;
  DyParts dyp;      // DYS_ > 0: dY = masked sum of dyp's slabs instead of `dy`; the
                    // first row tile of every column tile also writes it to dyp.out
};

template <int WM_, int WN_, int WK_, int KT_ = 1, int DYS_ = 0>
struct FcWgradOp {
  static constexpr int WM = WM_, WN = WN_, WK = WK_, KT = KT_, CPS = WK_ * KT_;
  static constexpr int A_LAYOUT = DZ_RC, B_LAYOUT = DZ_RC, A_MAP = DZ_MAP_QUAD;
  static constexpr int BM = 32 * WM, BN = 32 * WN, BK = 16 * CPS;
  typedef FcWgradParams Params;

  struct Tile : DzTile { FcHead hd; };

  __device__ static bool tile(const Params& p, const dim3& bid, Tile& t) {
    t.z = bid.z;  // head
    t.hd = dz_pick_head(p.head, t.z);
    const FcHead& hd = t.hd;
    t.m0 = bid.y * BM;  // k rows
    t.n0 = bid.x * BN;
    t.st_begin = 0;
    t.st_end = (p.M + BK - 1) / BK;
    t.z2 = (int)(((bid.z * p.sq_ny + bid.y) * p.sq_nx + bid.x) * (WM * WN));
    const bool ok = t.z < p.NH && t.m0 < hd.K && t.n0 < hd.N;
    if (!ok && p.sumsq && threadIdx.x < WM * WN) p.sumsq[t.z2 + threadIdx.x] = 0.f;
    return ok;
  }
  __device__ static float4 load_a(const Params& p, const Tile& t, int st, int c,
                                  int kk, int rq) {
    const FcHead& hd = t.hd;
    const int m = st * BK + c * 16 + kk;
    const int k = min(t.m0 + 4 * rq, hd.K - 4);
    return dz_sel4(m < p.M, dz_ld4(p.x + (long)min(m, p.M - 1) * p.ldx + hd.x_off + k));
  }
  __device__ static float4 load_b(const Params& p, const Tile& t, int st, int c,
                                  int kk, int rq) {
    const FcHead& hd = t.hd;
    const int m = st * BK + c * 16 + kk;
    const int n = min(t.n0 + 4 * rq, hd.ldw - 4);
    const long o = (long)min(m, p.M - 1) * p.ldy + hd.out_off + n;
    if constexpr (DYS_ > 0) {
      const float4 v = dz_dy_parts4<DYS_>(p.dyp, o);
      if (t.m0 == 0 && p.dyp.out && m < p.M) *(float4*)(p.dyp.out + o) = v;
      return dz_sel4(m < p.M, v);
    } else {
      return dz_sel4(m < p.M, dz_ld4(p.dy + o));
    }
  }
  __device__ static void store(const Params& p, const Tile& t, int wm, int wn,
                               int lane, const f32x16& acc) {
    const FcHead& hd = t.hd;
    const int col = t.n0 + wn * 32 + (lane & 31);
    const bool colok = col < hd.N;
    const float eo = (p.noisy && colok) ? p.noise[hd.eps_out + col] : 0.f;
    // all 16 eps_in values up front (clamped, unconditional): loaded inside the
    // row loop they are 16 serial load -> wait -> multiply -> store round trips
    // (not noisy: p.noise may be null -> any valid K floats, values unused)
    const float* eip = p.noisy ? p.noise + hd.eps_in : p.x;
    float ei[16];
#pragma unroll
    for (int r = 0; r < 16; ++r)
      ei[r] = eip[min(t.m0 + wm * 32 + dz_acc_row(r, lane), hd.K - 1)];
    float sq = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int k = t.m0 + wm * 32 + dz_acc_row(r, lane);
      if (colok && k < hd.K) {
        const float v = acc[r];
        p.grad[hd.w_mu + (long)k * hd.ldw + col] = v;
        sq += v * v;
        if (p.noisy) {
          const float vs = v * (ei[r] * eo);
          if (!p.skip_sig_store) p.grad[hd.w_sig + (long)k * hd.ldw + col] = vs;
          sq += vs * vs;
        }
      }
    }
    if (p.sumsq) {
      sq = dz_wave_sum(sq);
      if (lane == 0) p.sumsq[t.z2 + wm * WN + wn] = sq;
    }
  }
};

// --------------------------------------------------------------------------- //
//  Convolution weight AND bias gradient:
//     dW[k][co] = sum_pixels patch(pixel,k) dY[pixel][co],   db[co] = sum_pixels dY
//  rows = k (contiguous inside a kernel row) plus ONE extra row k == K whose
//  "patch" value is 1, so the bias gradient falls out of the same MFMAs;
//  reduction = output pixels (grid split).  part layout [S][KROWS][CO] with
//  rows [0,K) the weights and row K the bias -- the same order as the parameter
//  buffer (conv_b directly follows conv_w), so one reduction pass writes both.
// --------------------------------------------------------------------------- //
struct ConvWgradParams {
  const void* in;   // layer input, u8 or f32 [B][H][W][C]
  const float* dy;  // [B*OH*OW][CO]
  float* part;      // [S][KROWS][CO]
  int B;
  int S;
};

template <int IN_U8, int H, int W, int C, int KS, int S, int OH, int OW, int CO,
          int WM_, int WN_, int WK_, int KT_ = 1, int AM_ = 1>
struct ConvWgradOp {
  static constexpr int WM = WM_, WN = WN_, WK = WK_, KT = KT_, CPS = WK_ * KT_;
  static constexpr int PIN_LOADS = AM_;
  static constexpr int A_LAYOUT = DZ_RC, B_LAYOUT = DZ_RC, A_MAP = DZ_MAP_QUAD;
  static constexpr int BM = 32 * WM, BN = 32 * WN, BK = 16 * CPS;
  static constexpr int K = KS * KS * C;
  static constexpr int KROWS = K + 1;                  // + the bias row
  static constexpr int MT = (KROWS + BM - 1) / BM;     // row tiles
  static_assert(K % BM == 0 && CO % BN == 0 && C % 4 == 0, "tile shape");
  typedef ConvWgradParams Params;
  typedef DzTile Tile;

  __device__ static bool tile(const Params& p, const dim3& bid, Tile& t) {
    const int stages = (p.B * OH * OW + BK - 1) / BK;
    const int per = (stages + p.S - 1) / p.S;
    t.z = bid.z;
    t.m0 = bid.y * BM;
    t.n0 = bid.x * BN;
    t.st_begin = t.z * per;
    t.st_end = min(stages, t.st_begin + per);
    return true;
  }
  __device__ static float4 load_a(const Params& p, const Tile& t, int st, int c,
                                  int kk, int rq) {
    const int rows = p.B * OH * OW;
    const int ml = st * BK + c * 16 + kk;
    const int mc = min(ml, rows - 1);
    const int img = mc / (OH * OW), pix = mc % (OH * OW);
    const int oh = pix / OW, ow = pix % OW;
    const int k = t.m0 + 4 * rq;
    const int kc = min(k, K - 4);
    const int tap = kc / C, ci = kc % C;
    const int kh = tap / KS, kw = tap % KS;
    const long off = (((long)img * H + oh * S + kh) * W + ow * S + kw) * C + ci;
    // rows >= K: the bias row (value 1 at k == K) then zero padding; pixels beyond the
    // batch: zero -- by ADDRESS (third loader rule)
    // (dz_val: selects on VALUES; nested ?: here become branches with one load each)
    const float* src = (const float*)p.in + off;
    if constexpr (AM_) {
      src = dz_val(k < K, src, dz_val(k == K, (const float*)dz_page_one, (const float*)dz_page_zero));
      return dz_ld4(dz_val(ml < rows, src, (const float*)dz_page_zero));
    } else {
      float4 v = dz_ld4(src);
      v = k < K ? v : dz_f4(k == K ? 1.f : 0.f, 0.f, 0.f, 0.f);
      return dz_sel4(ml < rows, v);
    }
  }
  // uint8 input (conv1): the four raw bytes, converted when the stage is written to LDS
  static constexpr int A_RAW4 = IN_U8;
  __device__ static unsigned load_a_raw4(const Params& p, const Tile& t, int st, int c,
                                         int kk, int rq) {
    const int rows = p.B * OH * OW;
    const int ml = st * BK + c * 16 + kk;
    const int mc = min(ml, rows - 1);
    const int img = mc / (OH * OW), pix = mc % (OH * OW);
    const int oh = pix / OW, ow = pix % OW;
    const int k = t.m0 + 4 * rq;
    const int kc = min(k, K - 4);
    const int tap = kc / C, ci = kc % C;
    const int kh = tap / KS, kw = tap % KS;
    const long off = (((long)img * H + oh * S + kh) * W + ow * S + kw) * C + ci;
    const unsigned* src = (const unsigned*)((const uint8_t*)p.in + off);
    src = dz_val(k < K, src, dz_val(k == K, (const unsigned*)dz_page_u8one, (const unsigned*)dz_page_zero));
    return *dz_val(ml < rows, src, (const unsigned*)dz_page_zero);
  }
  __device__ static float4 cook4(unsigned raw) { return dz_u8x4_to_unit(raw); }
  __device__ static float4 load_b(const Params& p, const Tile& t, int st, int c,
                                  int kk, int rq) {
    const int rows = p.B * OH * OW;
    const int ml = st * BK + c * 16 + kk;
    const float* src = p.dy + (long)min(ml, rows - 1) * CO + t.n0 + 4 * rq;
    if constexpr (AM_) return dz_ld4(ml < rows ? src : dz_page_zero);
    else return dz_sel4(ml < rows, dz_ld4(src));
  }
  __device__ static void store(const Params& p, const Tile& t, int wm, int wn,
                               int lane, const f32x16& acc) {
    const int col = t.n0 + wn * 32 + (lane & 31);
    float* base = p.part + (long)t.z * KROWS * CO + col;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int k = t.m0 + wm * 32 + dz_acc_row(r, lane);
      if (k < KROWS) base[(long)k * CO] = acc[r];
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
          int WM_, int WN_, int WK_, int KT_ = 1, int AM_ = 1>
struct ConvDgradOp {
  static constexpr int WM = WM_, WN = WN_, WK = WK_, KT = KT_, CPS = WK_ * KT_;
  static constexpr int PIN_LOADS = AM_;
  static constexpr int A_LAYOUT = DZ_KC, B_LAYOUT = DZ_KC, A_MAP = DZ_MAP_QUAD;
  static constexpr int BM = 32 * WM, BN = 32 * WN, BK = 16 * CPS;
  static constexpr int TS = (KS + S - 1) / S;   // taps per dimension per class
  static constexpr int HP = (H + S - 1) / S, WP = (W + S - 1) / S;  // class grid
  static constexpr int RED = TS * TS * CO;
  static_assert(CO % 16 == 0 && RED % BK == 0 && C % BN == 0, "tile shape");
  static_assert(H % S == 0 && W % S == 0 && KS % S == 0, "uniform parity classes");
  typedef ConvDgradParams Params;
  typedef DzTile Tile;

  static int tiles(int B) { return (B * HP * WP + BM - 1) / BM; }

  __device__ static bool tile(const Params& p, const dim3& bid, Tile& t) {
    t.z = bid.z;  // parity class
    t.m0 = bid.y * BM;
    t.n0 = bid.x * BN;
    t.st_begin = 0;
    t.st_end = RED / BK;
    return true;
  }
  __device__ static bool pixel(const Params& p, const Tile& t, int row, int& img,
                               int& h, int& w) {
    const int rows = p.B * HP * WP;
    const int ml = t.m0 + row;
    const int mc = min(ml, rows - 1);
    img = mc / (HP * WP);
    const int pix = mc % (HP * WP);
    h = (pix / WP) * S + t.z / S;
    w = (pix % WP) * S + t.z % S;
    return ml < rows;
  }
  __device__ static float4 load_a(const Params& p, const Tile& t, int st, int c,
                                  int row, int q) {
    int img, h, w;
    bool ok = pixel(p, t, row, img, h, w);
    const int r0 = st * BK + c * 16 + 4 * q;
    const int tap = r0 / CO, co = r0 % CO;
    const int kh = (t.z / S) + (tap / TS) * S, kw = (t.z % S) + (tap % TS) * S;  // < KS
    const int oh = (h - kh) / S, ow = (w - kw) / S;  // exact when h >= kh, w >= kw
    ok = ok & (h >= kh) & (w >= kw) & (oh < OH) & (ow < OW);
    const int ohc = min(max(oh, 0), OH - 1), owc = min(max(ow, 0), OW - 1);
    const float* src = p.dy + (((long)img * OH + ohc) * OW + owc) * CO + co;
    if constexpr (AM_) return dz_ld4(ok ? src : dz_page_zero);
    else return dz_sel4(ok, dz_ld4(src));
  }
  __device__ static float4 load_b(const Params& p, const Tile& t, int st, int c,
                                  int row, int q) {
    const int ci = t.n0 + row;
    const int r0 = st * BK + c * 16 + 4 * q;
    const int tap = r0 / CO, co = r0 % CO;
    const int kh = (t.z / S) + (tap / TS) * S, kw = (t.z % S) + (tap % TS) * S;
    return dz_ld4(p.w + ((long)(kh * KS + kw) * C + ci) * CO + co);
  }
  static constexpr int SPLIT_STORE = 0;  // distributed epilogue measured slower for this Op (EXPERIMENTS.md)
  // the ReLU mask values (and addresses) of this wave's 16 output rows, requested in the
  // prologue: loaded inside store() they were a memory round trip behind the last MFMA
  struct Pre { unsigned o[16]; float mk[16]; };
  __device__ static Pre prefetch(const Params& p, const Tile& t, int wm, int wn, int lane,
                                 unsigned rmask) {
    Pre pre;
    const int ci = t.n0 + wn * 32 + (lane & 31);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      int img, h, w;
      const bool ok = pixel(p, t, wm * 32 + dz_acc_row(r, lane), img, h, w);   // (clamped: valid address)
      pre.o[r] = (unsigned)(((img * H + h) * W + w) * C + ci);
      pre.mk[r] = p.act[pre.o[r]];
      if (!ok) pre.o[r] = 0xffffffffu;
    }
    return pre;
  }
  __device__ static void store(const Params& p, const Tile& t, int wm, int wn,
                               int lane, const f32x16& acc, unsigned rmask, const Pre& pre) {
#pragma unroll
    for (int r = 0; r < 16; ++r)
      if (pre.o[r] != 0xffffffffu && ((rmask >> r) & 1u))
        p.dx[pre.o[r]] = pre.mk[r] > 0.f ? acc[r] : 0.f;
  }
};
