// fp32-MFMA tile GEMM template for gfx950 (wave64, v_mfma_f32_32x32x2_f32).
//
// C[m][n] = sum_k A(m,k) * B(k,n) where A and B are produced by an `Op`
// (implicit-GEMM gathers for convolutions, plain / noisy / transposed views for
// the linear layers), so one scheduling skeleton serves every contraction of the
// learner step, forward and backward.
//
// Structure (one workgroup = 4 waves = 256 threads):
//   * the 4 waves are arranged WM x WN x WK; every wave owns ONE 32x32 output
//     tile and one 16-deep k-chunk of each stage, i.e. a single MFMA accumulator
//     chain (the 32x32x2 f32 MFMA issues every 64 cycles and its dependent
//     latency is also 64 cycles, so one chain per wave already runs the matrix
//     pipe at its issue rate; MI355X_MICROARCH.md "Per-instruction cycle
//     constants");
//   * a stage = BK = 16*WK*KT reduction indices (every wave consumes KT chunks
//     of 16 per stage: 8*KT MFMAs between barriers, and 16*WK*KT rows of both
//     operands in flight per workgroup, which is what hides HBM latency at the
//     1-2 workgroups/CU these small problems give).  Global loads for stage s+1
//     are issued into registers before the MFMAs of stage s (register-staged
//     pipeline, cdna_hip_programming.md T14), written to LDS after the barrier;
//   * LDS tiles come in two layouts chosen per operand so that global loads are
//     16-byte and contiguous in whichever dimension memory is contiguous:
//       KC: [rows][16 (+4 pad)]  reduction-contiguous, fragment = 2 x ds_read_b128
//       RC: [16][rows]           row-contiguous,       fragment = 8 x ds_read_b32
//     Within a 16-chunk the MFMA k-slots are permuted (lane half h takes
//     k = 8h+s at step s) identically for A and B, which is what makes the KC
//     fragment two contiguous 16-byte reads;
//   * WK > 1 waves reduce their accumulators through LDS in the epilogue.
//
// fp32 in / fp32 accumulate is required by the 1e-5 loss tolerance
// (BASELINE.json north_star); there is no TF32 on gfx950.
#pragma once

#include "dz_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

enum { DZ_KC = 0, DZ_RC = 1 };           // LDS layouts
enum { DZ_MAP_QUAD = 0, DZ_MAP_ROW16 = 1 };  // KC loader thread mappings

struct DzTile {
  int m0;        // first row of the tile (Op-defined space)
  int n0;        // first column
  int st_begin;  // stage range [st_begin, st_end)
  int st_end;
  int z;         // Op-defined (group / head / split / parity class)
  int z2;
};

__device__ __forceinline__ float4 dz_f4(float a, float b, float c, float d) {
  float4 v; v.x = a; v.y = b; v.z = c; v.w = d; return v;
}
__device__ __forceinline__ float4 dz_f4zero() { return dz_f4(0.f, 0.f, 0.f, 0.f); }
__device__ __forceinline__ float4 dz_mul4(float4 a, float4 b) {
  return dz_f4(a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w);
}
__device__ __forceinline__ float4 dz_scale4(float4 a, float s) {
  return dz_f4(a.x * s, a.y * s, a.z * s, a.w * s);
}
// 4 consecutive floats p[i..i+3] with bounds [0, n) and no alignment assumption.
__device__ __forceinline__ float4 dz_load4_masked(const float* __restrict__ p,
                                                  int i, int n) {
  if (i + 3 < n && ((((uintptr_t)(p + i)) & 15) == 0)) return *(const float4*)(p + i);
  float4 v = dz_f4zero();
  if (i < n) v.x = p[i];
  if (i + 1 < n) v.y = p[i + 1];
  if (i + 2 < n) v.z = p[i + 2];
  if (i + 3 < n) v.w = p[i + 3];
  return v;
}

// What a masked-out loader slot READS (third loader rule, dz_qnet_ops.h): 16 zero bytes, or
// (1, 0, 0, 0) for the bias row of a weight-gradient operand (as floats, and as the bytes
// 255, 0, 0, 0 that the uint8 conversion turns into the same values).  Deliberately not
// `const`: the compiler must not fold the load into a select on the loaded VALUE.
__device__ __attribute__((aligned(16))) static float dz_page_zero[4] = {0.f, 0.f, 0.f, 0.f};
__device__ __attribute__((aligned(16))) static float dz_page_one[4] = {1.f, 0.f, 0.f, 0.f};
__device__ __attribute__((aligned(16))) static unsigned dz_page_u8one[4] = {255u, 0u, 0u, 0u};

// Optional per-wave register tiling: an Op may define MI / NI (default 1): every
// wave then owns MI x NI accumulators (a (32 MI) x (32 NI) output block), i.e.
// MI*NI independent MFMA chains fed by MI + NI operand fragments -- twice the
// MFMAs per LDS byte and no back-to-back dependent issue at MI = NI = 2.
template <class Op, class = void> struct DzMI { static constexpr int v = 1; };
template <class Op> struct DzMI<Op, decltype((void)Op::MI)> { static constexpr int v = Op::MI; };
template <class Op, class = void> struct DzNI { static constexpr int v = 1; };
template <class Op> struct DzNI<Op, decltype((void)Op::NI)> { static constexpr int v = Op::NI; };

// Optional deferred operand conversion (KC+ROW16 A operands): an Op with
// A_RAW16 = 1 provides  uint4 load_a16_raw(p, t, st, c, row)  and
// void cook16(uint4, float4 (&v)[4]).  The raw 16 bytes stay in 4 registers while
// the MFMAs of the current stage run and are converted when they are written to
// LDS; converting in the loader makes the compiler wait for the prefetch right
// after issuing it (vmcnt(0) + v_cvt in the middle of the MFMA block: conv1 ISA).
template <class Op, class = void> struct DzRaw16 { static constexpr int v = 0; };
template <class Op> struct DzRaw16<Op, decltype((void)Op::A_RAW16)> { static constexpr int v = Op::A_RAW16; };
// The same for RC A operands fed from uint8 (conv1's weight gradient): an Op with A_RAW4 = 1
// provides  unsigned load_a_raw4(p, t, st, c, kk, rq)  (4 bytes = 4 consecutive rows) and
// float4 cook4(unsigned); the 4 raw bytes wait in ONE register under the MFMAs.
// Optional epilogue operands requested in the PROLOGUE: an Op with `struct Pre` and
//   Pre prefetch(p, t, wm, wn, lane [, rmask])
// gets `pre` as the last argument of store().  A bias or a ReLU mask first loaded inside
// store() is a full memory round trip between the last MFMA and the first store (in-kernel
// stamps, round 4: 0.7-1.0 us of every conv launch's "store" phase).
template <class Op, class = void> struct DzHasPre { static constexpr int v = 0; };
template <class Op> struct DzHasPre<Op, decltype((void)sizeof(typename Op::Pre))> { static constexpr int v = 1; };
template <class Op, class = void> struct DzHasDbg { static constexpr int v = 0; };
template <class Op> struct DzHasDbg<Op, decltype((void)Op::HAS_DBG)> { static constexpr int v = 1; };
template <class Op, class = void> struct DzRaw4 { static constexpr int v = 0; };
template <class Op> struct DzRaw4<Op, decltype((void)Op::A_RAW4)> { static constexpr int v = Op::A_RAW4; };
// Optional: Op::PIN_LOADS = 1 keeps the next stage's global loads in front of the MFMA block.
template <class Op, class = void> struct DzPinLoads { static constexpr int v = 0; };
template <class Op> struct DzPinLoads<Op, decltype((void)Op::PIN_LOADS)> { static constexpr int v = Op::PIN_LOADS; };

// Optional distributed epilogue (WK > 1, MI = NI = 1): an Op with SPLIT_STORE = 1
// takes  store(p, t, wm, wn, lane, acc, rmask)  and stores only the accumulator
// registers r with bit r of rmask set.  All WK waves of a tile then exchange their
// accumulators through LDS and each finishes 16/WK of the 16 registers (sum over
// the k-groups in the order 0, 1, .. WK-1, as the single-wave epilogue does), instead
// of WK-1 waves retiring while wave 0 reads 16 (WK-1) partials and issues all 16
// row stores.
template <class Op, class = void> struct DzSplitStore { static constexpr int v = 0; };
template <class Op> struct DzSplitStore<Op, decltype((void)Op::SPLIT_STORE)> { static constexpr int v = Op::SPLIT_STORE; };

template <int ROWS, int CPS, int LAYOUT>
struct DzLdsTile {
  static constexpr int LD = (LAYOUT == DZ_KC) ? 20 : ROWS;
  // KC chunk pitch: ROWS*20 floats is a multiple of 64 banks for ROWS = 64, 128, so
  // the ROW16 loader's 4 lanes that write the same row of 4 different chunks all hit
  // one bank (conv1 forward: 921 600 conflict cycles per launch); 16 floats of
  // skew put the chunks of a row 16 banks apart
  static constexpr int CHUNK = (LAYOUT == DZ_KC)
      ? ROWS * 20 + (((ROWS * 20) % 64 == 0 && CPS > 1) ? 16 : 0) : 16 * ROWS;
  static constexpr int ELEMS = CPS * CHUNK;
  // number of float4 "load slots" per stage and per thread
  static constexpr int SLOTS = ROWS * 16 * CPS / 4;
  static constexpr int PER_THREAD = (SLOTS + 255) / 256;
};

// The kernel.  Op interface (all static, __device__):
//   constants  WM, WN, WK, A_LAYOUT, B_LAYOUT, A_MAP (KC only)
//   struct Params
//   struct Tile : DzTile (or DzTile itself): tile coordinates plus per-group
//          pointers / scalars the Op resolves ONCE (see the loader rule in
//          dz_qnet_ops.h: no dynamically indexed kernel-argument arrays in loaders)
//   bool  tile(const Params&, bid, Tile&)                    -- from blockIdx
//   KC+QUAD : float4 load_a(p, t, st, c, row, q)  4 consecutive reduction idx
//   KC+ROW16: void   load_a16(p, t, st, c, row, float4 (&v)[4])
//   RC      : float4 load_a(p, t, st, c, kk, rq)  4 consecutive rows
//   (same three forms for load_b; rows are tile columns)
//   void  store(p, t, wm, wn, lane, acc)
template <class Op>
struct DzGemmSmem {
  static constexpr int CPS = Op::WK * Op::KT;
  static constexpr int MI = DzMI<Op>::v, NI = DzNI<Op>::v;
  using AT = DzLdsTile<32 * Op::WM * MI, CPS, Op::A_LAYOUT>;
  using BT = DzLdsTile<32 * Op::WN * NI, CPS, Op::B_LAYOUT>;
  static constexpr int RED =
      (Op::WK > 1) ? (Op::WK - (DzSplitStore<Op>::v ? 0 : 1)) * Op::WM * Op::WN * MI * NI * 16 * 64 : 0;
  static constexpr int TILE = AT::ELEMS + BT::ELEMS;
  static constexpr int ELEMS = TILE > RED ? TILE : RED;
};

// One workgroup's worth of the contraction; `bid` is the (possibly virtual)
// block index the Op decodes its tile from, `smem` >= DzGemmSmem<Op>::ELEMS.
template <class Op>
__device__ __forceinline__ void dz_gemm_body(const typename Op::Params& p, const dim3& bid,
                                             float* smem) {
  constexpr int WM = Op::WM, WN = Op::WN, WK = Op::WK, KT = Op::KT;
  constexpr int CPS = WK * KT;  // 16-deep chunks per stage
  static_assert(WM * WN * WK == 4, "4 waves per workgroup");
  constexpr int MI = DzMI<Op>::v, NI = DzNI<Op>::v;
  constexpr int BM = 32 * WM * MI, BN = 32 * WN * NI;
  using AT = DzLdsTile<BM, CPS, Op::A_LAYOUT>;
  using BT = DzLdsTile<BN, CPS, Op::B_LAYOUT>;
  float* As = smem;
  float* Bs = smem + AT::ELEMS;

  typename Op::Tile t;  // DzTile + whatever the Op resolves once per workgroup
  if (!Op::tile(p, bid, t)) return;
#ifdef DZ_GEMM_STAMPS
  long long* dz_dbg = nullptr;
  if constexpr (DzHasDbg<Op>::v) dz_dbg = p.dbg ? p.dbg + (long)(bid.x + bid.y * 4) * 8 : nullptr;
#define DZ_GSTAMP(i) do { if (dz_dbg && threadIdx.x == 0) dz_dbg[i] = wall_clock64(); } while (0)
#else
#define DZ_GSTAMP(i) do {} while (0)
#endif
  DZ_GSTAMP(0);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wk = wave / (WM * WN);
  const int wm = (wave % (WM * WN)) / WN;
  const int wn = wave % WN;
  const int half = lane >> 5;
  const int l31 = lane & 31;

  // (the epilogue's row mask of the distributed store, known now)
  constexpr unsigned kRpw = (WK > 1 && DzSplitStore<Op>::v && DzMI<Op>::v == 1 && DzNI<Op>::v == 1)
                                ? 16 / WK : 16;
  const unsigned my_rmask = kRpw == 16 ? 0xffffu : ((1u << kRpw) - 1u) << (wk * kRpw);
  auto prefetch = [&]() {
    if constexpr (DzHasPre<Op>::v) return Op::prefetch(p, t, wm, wn, lane, my_rmask);
    else return 0;
  };
  auto pre = prefetch();
  (void)pre;
  constexpr int A_ROW16 = (Op::A_LAYOUT == DZ_KC && Op::A_MAP == DZ_MAP_ROW16);
  constexpr int NA = A_ROW16 ? ((BM * CPS + 255) / 256) * 4 : AT::PER_THREAD;
  constexpr int NB = BT::PER_THREAD;
  float4 ra[NA];
  float4 rb[NB];
  // When the slot count is a multiple of the workgroup size every thread's slot
  // is real: the bounds test must fold away (a residual runtime test makes the
  // compiler wrap each load in its own exec-mask block with a vmcnt(0) wait).
  constexpr bool A_FULL = A_ROW16 ? ((BM * CPS) % 256 == 0) : (AT::SLOTS % 256 == 0);
  constexpr bool B_FULL = BT::SLOTS % 256 == 0;

  constexpr bool A_RAW = A_ROW16 && DzRaw16<Op>::v;
  constexpr bool A_RAW4 = Op::A_LAYOUT == DZ_RC && DzRaw4<Op>::v;
  auto load_stage = [&](int st) {
    if constexpr (A_RAW4) {
#pragma unroll
      for (int j = 0; j < NA; ++j) {
        const int idx = tid + j * 256;
        const int kidx = idx / (BM / 4), rq = idx % (BM / 4);
        unsigned raw = 0u;
        if (A_FULL || idx < AT::SLOTS) raw = Op::load_a_raw4(p, t, st, kidx >> 4, kidx & 15, rq);
        ra[j].x = __builtin_bit_cast(float, raw);  // (.y .z .w stay unused)
      }
    } else if constexpr (A_RAW) {
#pragma unroll
      for (int j = 0; j < NA / 4; ++j) {
        const int idx = tid + j * 256;
        uint4 raw = make_uint4(0u, 0u, 0u, 0u);
        if (A_FULL || idx < BM * CPS) raw = Op::load_a16_raw(p, t, st, idx % CPS, idx / CPS);
        ra[4 * j] = __builtin_bit_cast(float4, raw);  // the other three stay unused
      }
    } else if constexpr (A_ROW16) {
#pragma unroll
      for (int j = 0; j < NA / 4; ++j) {
        const int idx = tid + j * 256;
        float4 v[4] = {dz_f4zero(), dz_f4zero(), dz_f4zero(), dz_f4zero()};
        if (A_FULL || idx < BM * CPS) Op::load_a16(p, t, st, idx % CPS, idx / CPS, v);
        ra[4 * j] = v[0]; ra[4 * j + 1] = v[1]; ra[4 * j + 2] = v[2]; ra[4 * j + 3] = v[3];
      }
    } else if constexpr (Op::A_LAYOUT == DZ_KC) {
#pragma unroll
      for (int j = 0; j < NA; ++j) {
        const int idx = tid + j * 256;
        const int row = idx / (4 * CPS), rem = idx % (4 * CPS);
        ra[j] = (A_FULL || idx < AT::SLOTS) ? Op::load_a(p, t, st, rem >> 2, row, rem & 3)
                                            : dz_f4zero();
      }
    } else {
#pragma unroll
      for (int j = 0; j < NA; ++j) {
        const int idx = tid + j * 256;
        const int kidx = idx / (BM / 4), rq = idx % (BM / 4);
        ra[j] = (A_FULL || idx < AT::SLOTS) ? Op::load_a(p, t, st, kidx >> 4, kidx & 15, rq)
                                            : dz_f4zero();
      }
    }
    if constexpr (Op::B_LAYOUT == DZ_KC) {
#pragma unroll
      for (int j = 0; j < NB; ++j) {
        const int idx = tid + j * 256;
        const int row = idx / (4 * CPS), rem = idx % (4 * CPS);
        rb[j] = (B_FULL || idx < BT::SLOTS) ? Op::load_b(p, t, st, rem >> 2, row, rem & 3)
                                            : dz_f4zero();
      }
    } else {
#pragma unroll
      for (int j = 0; j < NB; ++j) {
        const int idx = tid + j * 256;
        const int kidx = idx / (BN / 4), rq = idx % (BN / 4);
        rb[j] = (B_FULL || idx < BT::SLOTS) ? Op::load_b(p, t, st, kidx >> 4, kidx & 15, rq)
                                            : dz_f4zero();
      }
    }
  };

  auto store_stage = [&]() {
    if constexpr (A_RAW4) {
#pragma unroll
      for (int j = 0; j < NA; ++j) {
        const int idx = tid + j * 256;
        const int kidx = idx / (BM / 4), rq = idx % (BM / 4);
        if (A_FULL || idx < AT::SLOTS)
          *(float4*)(As + (kidx >> 4) * AT::CHUNK + (kidx & 15) * BM + 4 * rq) =
              Op::cook4(__builtin_bit_cast(unsigned, ra[j].x));
      }
    } else if constexpr (A_RAW) {
#pragma unroll
      for (int j = 0; j < NA / 4; ++j) {
        const int idx = tid + j * 256;
        if (A_FULL || idx < BM * CPS) {
          float4 v[4];
          Op::cook16(__builtin_bit_cast(uint4, ra[4 * j]), v);
          float* dst = As + (idx % CPS) * AT::CHUNK + (idx / CPS) * 20;
#pragma unroll
          for (int q = 0; q < 4; ++q) *(float4*)(dst + 4 * q) = v[q];
        }
      }
    } else if constexpr (A_ROW16) {
#pragma unroll
      for (int j = 0; j < NA / 4; ++j) {
        const int idx = tid + j * 256;
        if (A_FULL || idx < BM * CPS) {
          float* dst = As + (idx % CPS) * AT::CHUNK + (idx / CPS) * 20;
#pragma unroll
          for (int q = 0; q < 4; ++q) *(float4*)(dst + 4 * q) = ra[4 * j + q];
        }
      }
    } else if constexpr (Op::A_LAYOUT == DZ_KC) {
#pragma unroll
      for (int j = 0; j < NA; ++j) {
        const int idx = tid + j * 256;
        const int row = idx / (4 * CPS), rem = idx % (4 * CPS);
        if (A_FULL || idx < AT::SLOTS)
          *(float4*)(As + (rem >> 2) * AT::CHUNK + row * 20 + 4 * (rem & 3)) = ra[j];
      }
    } else {
#pragma unroll
      for (int j = 0; j < NA; ++j) {
        const int idx = tid + j * 256;
        const int kidx = idx / (BM / 4), rq = idx % (BM / 4);
        if (A_FULL || idx < AT::SLOTS)
          *(float4*)(As + (kidx >> 4) * AT::CHUNK + (kidx & 15) * BM + 4 * rq) = ra[j];
      }
    }
    if constexpr (Op::B_LAYOUT == DZ_KC) {
#pragma unroll
      for (int j = 0; j < NB; ++j) {
        const int idx = tid + j * 256;
        const int row = idx / (4 * CPS), rem = idx % (4 * CPS);
        if (B_FULL || idx < BT::SLOTS)
          *(float4*)(Bs + (rem >> 2) * BT::CHUNK + row * 20 + 4 * (rem & 3)) = rb[j];
      }
    } else {
#pragma unroll
      for (int j = 0; j < NB; ++j) {
        const int idx = tid + j * 256;
        const int kidx = idx / (BN / 4), rq = idx % (BN / 4);
        if (B_FULL || idx < BT::SLOTS)
          *(float4*)(Bs + (kidx >> 4) * BT::CHUNK + (kidx & 15) * BN + 4 * rq) = rb[j];
      }
    }
  };

  f32x16 acc[MI][NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[mi][ni][i] = 0.f;

  if (t.st_begin < t.st_end) load_stage(t.st_begin);
  DZ_GSTAMP(1);
  for (int st = t.st_begin; st < t.st_end; ++st) {
    __syncthreads();  // everyone finished reading the previous stage
    store_stage();
    __syncthreads();
    if (st == t.st_begin) DZ_GSTAMP(2);
    if (st + 1 < t.st_end) load_stage(st + 1);  // in flight under the MFMAs
    // (optionally pinned, Op::PIN_LOADS: in a fully unrolled stage loop the scheduler otherwise
    // sinks these loads BELOW the MFMA block, next to the LDS writes that consume them --
    // conv2 forward ISA, rounds 1-3 -- and the stage pays their whole latency behind its last
    // MFMA.  Per Op, by measurement: conv2's backward pair runs 2.5 us SLOWER pinned)
    if constexpr (DzPinLoads<Op>::v) __builtin_amdgcn_sched_barrier(0);

#pragma unroll
    for (int kt = 0; kt < KT; ++kt) {
      const int ch = wk * KT + kt;  // this wave's chunk of the stage
      float fa[MI][8], fb[NI][8];
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) {
        const int r0 = (wm * MI + mi) * 32 + l31;
        if constexpr (Op::A_LAYOUT == DZ_KC) {
          const float* src = As + ch * AT::CHUNK + r0 * 20 + half * 8;
          const float4 v0 = *(const float4*)src, v1 = *(const float4*)(src + 4);
          fa[mi][0] = v0.x; fa[mi][1] = v0.y; fa[mi][2] = v0.z; fa[mi][3] = v0.w;
          fa[mi][4] = v1.x; fa[mi][5] = v1.y; fa[mi][6] = v1.z; fa[mi][7] = v1.w;
        } else {
          const float* src = As + ch * AT::CHUNK + (half * 8) * BM + r0;
#pragma unroll
          for (int s = 0; s < 8; ++s) fa[mi][s] = src[s * BM];
        }
      }
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) {
        const int c0 = (wn * NI + ni) * 32 + l31;
        if constexpr (Op::B_LAYOUT == DZ_KC) {
          const float* src = Bs + ch * BT::CHUNK + c0 * 20 + half * 8;
          const float4 v0 = *(const float4*)src, v1 = *(const float4*)(src + 4);
          fb[ni][0] = v0.x; fb[ni][1] = v0.y; fb[ni][2] = v0.z; fb[ni][3] = v0.w;
          fb[ni][4] = v1.x; fb[ni][5] = v1.y; fb[ni][6] = v1.z; fb[ni][7] = v1.w;
        } else {
          const float* src = Bs + ch * BT::CHUNK + (half * 8) * BN + c0;
#pragma unroll
          for (int s = 0; s < 8; ++s) fb[ni][s] = src[s * BN];
        }
      }
#pragma unroll
      for (int s = 0; s < 8; ++s)
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
          for (int ni = 0; ni < NI; ++ni)
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[mi][s], fb[ni][s],
                                                               acc[mi][ni], 0, 0, 0);
    }
  }

  DZ_GSTAMP(3);
  if constexpr (WK > 1 && DzSplitStore<Op>::v && MI == 1 && NI == 1) {
    __syncthreads();
    float* red = smem;   // [WK][WM*WN][16][64]
    constexpr int PER = WM * WN;
    constexpr int RPW = 16 / WK;   // accumulator registers finished per wave
    {
      float* dst = red + ((wk * PER + wm * WN + wn) * 16) * 64 + lane;
#pragma unroll
      for (int i = 0; i < 16; ++i) dst[i * 64] = acc[0][0][i];
    }
    __syncthreads();
    f32x16 out;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      out[i] = 0.f;
      if (i / RPW == wk) {   // wave-uniform
        const float* src = red + ((wm * WN + wn) * 16 + i) * 64 + lane;
        float v = src[0];
#pragma unroll
        for (int k2 = 1; k2 < WK; ++k2) v += src[(long)k2 * PER * 16 * 64];
        out[i] = v;
      }
    }
    DZ_GSTAMP(4);
    if constexpr (DzHasPre<Op>::v) Op::store(p, t, wm, wn, lane, out, ((1u << RPW) - 1u) << (wk * RPW), pre);
    else Op::store(p, t, wm, wn, lane, out, ((1u << RPW) - 1u) << (wk * RPW));
    DZ_GSTAMP(5);
    return;
  }
  if constexpr (WK > 1) {
    __syncthreads();
    float* red = smem;
    constexpr int PER = WM * WN * MI * NI;  // accumulator blocks per k-group
    if (wk > 0) {
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
          float* dst = red + (((wk - 1) * PER + ((wm * WN + wn) * MI + mi) * NI + ni) * 16) * 64 + lane;
#pragma unroll
          for (int i = 0; i < 16; ++i) dst[i * 64] = acc[mi][ni][i];
        }
    }
    __syncthreads();
    if (wk > 0) return;
#pragma unroll
    for (int k2 = 1; k2 < WK; ++k2)
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
          const float* src = red + (((k2 - 1) * PER + ((wm * WN + wn) * MI + mi) * NI + ni) * 16) * 64 + lane;
#pragma unroll
          for (int i = 0; i < 16; ++i) acc[mi][ni][i] += src[i * 64];
          // one accumulator block at a time: without this the compiler hoists all
          // (WK-1)*MI*NI*16 LDS loads and pays for them in VGPRs (occupancy)
          asm volatile("" ::: "memory");
        }
  }
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
      if constexpr (DzHasPre<Op>::v && MI == 1 && NI == 1) Op::store(p, t, wm, wn, lane, acc[mi][ni], 0xffffu, pre);
      else Op::store(p, t, wm * MI + mi, wn * NI + ni, lane, acc[mi][ni]);
}

// C/D fragment coordinates of v_mfma_f32_32x32x2_f32 (cdna_hip_programming.md 3):
// acc[r] is element (row, col) = ((r&3) + 8*(r>>2) + 4*(lane>>5), lane&31).
__device__ __forceinline__ int dz_acc_row(int r, int lane) {
  return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
}

template <class Op>
__global__ __launch_bounds__(256) void dz_mfma_gemm(typename Op::Params p) {
  __shared__ __attribute__((aligned(16))) float smem[DzGemmSmem<Op>::ELEMS];
  dz_gemm_body<Op>(p, dim3(blockIdx.x, blockIdx.y, blockIdx.z), smem);
}

template <class Op>
static inline int dz_launch_gemm(const typename Op::Params& p, dim3 grid,
                                 hipStream_t s) {
  hipLaunchKernelGGL(dz_mfma_gemm<Op>, grid, dim3(256), 0, s, p);
  DZ_LAUNCH_CHECK();
  return DZ_OK;
}

// Horizontal fusion: up to three INDEPENDENT contractions in one launch (e.g. a
// layer's weight gradient and input gradient).  Each fills only part of the
// 256 CUs on its own; fused they overlap with no stream/event traffic (a
// cross-stream event hop measured 7-14 us on this stack) and one launch floor
// (~4.5 us) instead of three.  Blocks [0,na) run OpA, [na,na+nb) OpB, rest OpC.
__device__ __forceinline__ dim3 dz_unflatten(unsigned i, dim3 g) {
  return dim3(i % g.x, (i / g.x) % g.y, i / (g.x * g.y));
}
static inline unsigned dz_count(dim3 g) { return g.x * g.y * g.z; }

template <class OpA, class OpB>
__global__ __launch_bounds__(256) void dz_mfma_gemm2(typename OpA::Params pa, dim3 ga,
                                                     typename OpB::Params pb, dim3 gb) {
  constexpr int SM = DzGemmSmem<OpA>::ELEMS > DzGemmSmem<OpB>::ELEMS
                         ? DzGemmSmem<OpA>::ELEMS : DzGemmSmem<OpB>::ELEMS;
  __shared__ __attribute__((aligned(16))) float smem[SM];
  const unsigned na = ga.x * ga.y * ga.z;
  if (blockIdx.x < na) dz_gemm_body<OpA>(pa, dz_unflatten(blockIdx.x, ga), smem);
  else dz_gemm_body<OpB>(pb, dz_unflatten(blockIdx.x - na, gb), smem);
}

// dz_mfma_gemm2 compiled for OCC waves per SIMD (register budget 512 / OCC): for launches
// whose workgroup count exceeds the co-resident slots at the default allocation.
template <class OpA, class OpB, int OCC>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(OCC, OCC)))
void dz_mfma_gemm2_occ(typename OpA::Params pa, dim3 ga, typename OpB::Params pb, dim3 gb) {
  constexpr int SM = DzGemmSmem<OpA>::ELEMS > DzGemmSmem<OpB>::ELEMS
                         ? DzGemmSmem<OpA>::ELEMS : DzGemmSmem<OpB>::ELEMS;
  __shared__ __attribute__((aligned(16))) float smem[SM];
  const unsigned na = ga.x * ga.y * ga.z;
  if (blockIdx.x < na) dz_gemm_body<OpA>(pa, dz_unflatten(blockIdx.x, ga), smem);
  else dz_gemm_body<OpB>(pb, dz_unflatten(blockIdx.x - na, gb), smem);
}
template <class OpA, class OpB, int OCC>
static inline int dz_launch_gemm2_occ(const typename OpA::Params& pa, dim3 ga,
                                      const typename OpB::Params& pb, dim3 gb, hipStream_t s) {
  hipLaunchKernelGGL((dz_mfma_gemm2_occ<OpA, OpB, OCC>), dim3(dz_count(ga) + dz_count(gb)),
                     dim3(256), 0, s, pa, ga, pb, gb);
  DZ_LAUNCH_CHECK();
  return DZ_OK;
}

template <class OpA, class OpB, class OpC>
__global__ __launch_bounds__(256) void dz_mfma_gemm3(typename OpA::Params pa, dim3 ga,
                                                     typename OpB::Params pb, dim3 gb,
                                                     typename OpC::Params pc, dim3 gc) {
  constexpr int S1 = DzGemmSmem<OpA>::ELEMS > DzGemmSmem<OpB>::ELEMS
                         ? DzGemmSmem<OpA>::ELEMS : DzGemmSmem<OpB>::ELEMS;
  constexpr int SM = S1 > DzGemmSmem<OpC>::ELEMS ? S1 : DzGemmSmem<OpC>::ELEMS;
  __shared__ __attribute__((aligned(16))) float smem[SM];
  const unsigned na = ga.x * ga.y * ga.z, nb = gb.x * gb.y * gb.z;
  if (blockIdx.x < na) dz_gemm_body<OpA>(pa, dz_unflatten(blockIdx.x, ga), smem);
  else if (blockIdx.x < na + nb) dz_gemm_body<OpB>(pb, dz_unflatten(blockIdx.x - na, gb), smem);
  else dz_gemm_body<OpC>(pc, dz_unflatten(blockIdx.x - na - nb, gc), smem);
}

// XCD-aware tile order.  Workgroups are dealt to the 8 XCDs round-robin by linear
// id and every XCD has its own 4 MB L2, so with the natural order (x fastest) the
// g.x tiles that share an A-operand slab land on 8 different XCDs and the slab is
// fetched from the fabric 8 times (IQN fc1: 8 x 77 MB per launch).  Here block L
// runs tile  x = (L/8) % g.x,  (y,z) = ((L/8) / g.x) * 8 + L%8 : all g.x tiles of
// one (y,z) slab run on ONE XCD, back to back.  The 1-D grid is padded to
// 8 * g.x * ceil(g.y*g.z / 8) blocks; the surplus blocks exit.
__device__ __forceinline__ bool dz_xcd_tile(unsigned L, dim3 g, dim3& bid) {
  const unsigned T = g.y * g.z;
  const unsigned xcd = L & 7, j = L >> 3;
  const unsigned m = (j / g.x) * 8 + xcd;
  if (m >= T) return false;
  bid = dim3(j % g.x, m % g.y, m / g.y);
  return true;
}
static inline unsigned dz_xcd_blocks(dim3 g) { return 8 * g.x * ((g.y * g.z + 7) / 8); }

template <class Op>
__global__ __launch_bounds__(256) void dz_mfma_gemm_xcd(typename Op::Params p, dim3 g) {
  __shared__ __attribute__((aligned(16))) float smem[DzGemmSmem<Op>::ELEMS];
  dim3 bid;
  if (!dz_xcd_tile(blockIdx.x, g, bid)) return;
  dz_gemm_body<Op>(p, bid, smem);
}
template <class Op>
static inline int dz_launch_gemm_xcd(const typename Op::Params& p, dim3 g, hipStream_t s) {
  hipLaunchKernelGGL(dz_mfma_gemm_xcd<Op>, dim3(dz_xcd_blocks(g)), dim3(256), 0, s, p, g);
  DZ_LAUNCH_CHECK();
  return DZ_OK;
}
// ... compiled for OCC waves per SIMD (register budget 512 / OCC).  At the default the
// compiler aims for 8 waves and squeezes a 64-VGPR allocation out of the stage loop by parking
// freshly LOADED registers in others around the MFMA block -- a copy that waits for the
// prefetch in front of the MFMAs (IQN fc1 forward ISA, round 5); the LDS block admits 3-5
// workgroups per CU anyway.
template <class Op, int OCC>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(OCC, OCC)))
void dz_mfma_gemm_xcd_occ(typename Op::Params p, dim3 g) {
  __shared__ __attribute__((aligned(16))) float smem[DzGemmSmem<Op>::ELEMS];
  dim3 bid;
  if (!dz_xcd_tile(blockIdx.x, g, bid)) return;
  dz_gemm_body<Op>(p, bid, smem);
}
template <class Op, int OCC>
static inline int dz_launch_gemm_xcd_occ(const typename Op::Params& p, dim3 g, hipStream_t s) {
  hipLaunchKernelGGL((dz_mfma_gemm_xcd_occ<Op, OCC>), dim3(dz_xcd_blocks(g)), dim3(256), 0, s, p, g);
  DZ_LAUNCH_CHECK();
  return DZ_OK;
}
template <class OpA, class OpB>
__global__ __launch_bounds__(256) void dz_mfma_gemm2_xcd(typename OpA::Params pa, dim3 ga,
                                                         typename OpB::Params pb, dim3 gb) {
  constexpr int SM = DzGemmSmem<OpA>::ELEMS > DzGemmSmem<OpB>::ELEMS
                         ? DzGemmSmem<OpA>::ELEMS : DzGemmSmem<OpB>::ELEMS;
  __shared__ __attribute__((aligned(16))) float smem[SM];
  const unsigned na = 8 * ga.x * ((ga.y * ga.z + 7) / 8);  // multiple of 8: XCD = id % 8 holds for B too
  dim3 bid;
  if (blockIdx.x < na) {
    if (dz_xcd_tile(blockIdx.x, ga, bid)) dz_gemm_body<OpA>(pa, bid, smem);
  } else {
    if (dz_xcd_tile(blockIdx.x - na, gb, bid)) dz_gemm_body<OpB>(pb, bid, smem);
  }
}
template <class OpA, class OpB>
static inline int dz_launch_gemm2_xcd(const typename OpA::Params& pa, dim3 ga,
                                      const typename OpB::Params& pb, dim3 gb,
                                      hipStream_t s) {
  hipLaunchKernelGGL((dz_mfma_gemm2_xcd<OpA, OpB>), dim3(dz_xcd_blocks(ga) + dz_xcd_blocks(gb)),
                     dim3(256), 0, s, pa, ga, pb, gb);
  DZ_LAUNCH_CHECK();
  return DZ_OK;
}

// A contraction plus an unrelated small elementwise job in the same launch
// (blocks beyond the GEMM grid run Side::run): saves the ~5 us launch floor of a
// tiny kernel that nothing in the GEMM depends on.
template <class Op, class Side>
__global__ __launch_bounds__(256) void dz_mfma_gemm_side(typename Op::Params p, dim3 g,
                                                         typename Side::Params sp) {
  __shared__ __attribute__((aligned(16))) float smem[DzGemmSmem<Op>::ELEMS];
  const unsigned n = g.x * g.y * g.z;
  if (blockIdx.x < n) dz_gemm_body<Op>(p, dz_unflatten(blockIdx.x, g), smem);
  else Side::run(sp, blockIdx.x - n);
}
template <class Op, class Side>
static inline int dz_launch_gemm_side(const typename Op::Params& p, dim3 g,
                                      const typename Side::Params& sp, unsigned side_blocks,
                                      hipStream_t s) {
  hipLaunchKernelGGL((dz_mfma_gemm_side<Op, Side>), dim3(dz_count(g) + side_blocks),
                     dim3(256), 0, s, p, g, sp);
  DZ_LAUNCH_CHECK();
  return DZ_OK;
}

// A contraction with `side_blocks` side-job workgroups FIRST in the grid (they are
// dispatched before the contraction's and may use its LDS block as scratch).
template <class Op, class Side>
__global__ __launch_bounds__(256) void dz_mfma_gemm_side_first(typename Op::Params p, dim3 g,
                                                               typename Side::Params sp,
                                                               unsigned side_blocks) {
  __shared__ __attribute__((aligned(16))) float smem[DzGemmSmem<Op>::ELEMS];
  if (blockIdx.x < side_blocks) Side::run(sp, blockIdx.x, smem, (int)sizeof(smem));
  else dz_gemm_body<Op>(p, dz_unflatten(blockIdx.x - side_blocks, g), smem);
}
template <class Op, class Side>
static inline int dz_launch_gemm_side_first(const typename Op::Params& p, dim3 g,
                                            const typename Side::Params& sp,
                                            unsigned side_blocks, hipStream_t s) {
  hipLaunchKernelGGL((dz_mfma_gemm_side_first<Op, Side>), dim3(dz_count(g) + side_blocks),
                     dim3(256), 0, s, p, g, sp, side_blocks);
  DZ_LAUNCH_CHECK();
  return DZ_OK;
}

// Two contractions plus a side job (see dz_mfma_gemm_side).
template <class OpA, class OpB, class Side>
__global__ __launch_bounds__(256) void dz_mfma_gemm2_side(typename OpA::Params pa, dim3 ga,
                                                          typename OpB::Params pb, dim3 gb,
                                                          typename Side::Params sp) {
  constexpr int SM = DzGemmSmem<OpA>::ELEMS > DzGemmSmem<OpB>::ELEMS
                         ? DzGemmSmem<OpA>::ELEMS : DzGemmSmem<OpB>::ELEMS;
  __shared__ __attribute__((aligned(16))) float smem[SM];
  const unsigned na = ga.x * ga.y * ga.z, nb = gb.x * gb.y * gb.z;
  if (blockIdx.x < na) dz_gemm_body<OpA>(pa, dz_unflatten(blockIdx.x, ga), smem);
  else if (blockIdx.x < na + nb) dz_gemm_body<OpB>(pb, dz_unflatten(blockIdx.x - na, gb), smem);
  else Side::run(sp, blockIdx.x - na - nb, smem, (int)sizeof(smem));
}
template <class OpA, class OpB, class Side>
static inline int dz_launch_gemm2_side(const typename OpA::Params& pa, dim3 ga,
                                       const typename OpB::Params& pb, dim3 gb,
                                       const typename Side::Params& sp, unsigned side_blocks,
                                       hipStream_t s) {
  hipLaunchKernelGGL((dz_mfma_gemm2_side<OpA, OpB, Side>),
                     dim3(dz_count(ga) + dz_count(gb) + side_blocks), dim3(256), 0, s, pa, ga,
                     pb, gb, sp);
  DZ_LAUNCH_CHECK();
  return DZ_OK;
}

template <class OpA, class OpB>
static inline int dz_launch_gemm2(const typename OpA::Params& pa, dim3 ga,
                                  const typename OpB::Params& pb, dim3 gb, hipStream_t s) {
  hipLaunchKernelGGL((dz_mfma_gemm2<OpA, OpB>), dim3(dz_count(ga) + dz_count(gb)),
                     dim3(256), 0, s, pa, ga, pb, gb);
  DZ_LAUNCH_CHECK();
  return DZ_OK;
}
template <class OpA, class OpB, class OpC>
static inline int dz_launch_gemm3(const typename OpA::Params& pa, dim3 ga,
                                  const typename OpB::Params& pb, dim3 gb,
                                  const typename OpC::Params& pc, dim3 gc, hipStream_t s) {
  hipLaunchKernelGGL((dz_mfma_gemm3<OpA, OpB, OpC>),
                     dim3(dz_count(ga) + dz_count(gb) + dz_count(gc)), dim3(256), 0, s, pa,
                     ga, pb, gb, pc, gc);
  DZ_LAUNCH_CHECK();
  return DZ_OK;
}
