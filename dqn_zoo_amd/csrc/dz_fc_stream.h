// Weight-streaming kernel for the wide linear layer (fc1: 3136 -> 512 or 2x512,
// batch <= 32): the one layer of the DQN-family nets whose forward cost is
// reading its weights (25.7 MB for the noisy mu+sigma pair) rather than arithmetic.
//
// With M <= 32 rows the contraction is a "fat GEMV": the tile-GEMM skeleton
// (global -> registers -> LDS -> barrier -> MFMA) has only one stage of loads in
// flight per workgroup and measured 1.2 TB/s.  Here every wave streams its slice
// of W from HBM straight into MFMA operand registers: lane l loads the single
// float W[k + (l>>5)][n0 + (l&31)], which IS its B operand of
// v_mfma_f32_32x32x2_f32; the loads run as a software pipeline of small chunks
// (3 x 5 k-pairs in flight per wave) that the MFMAs drain in order behind counted
// vmcnt waits -- no LDS for weights, no barrier in the loop.  The A operand (x) is
// staged once per workgroup in LDS, k-major, so the per-MFMA ds_read_b32 is
// conflict-free.  (Measured alternatives, removed: 16-byte loads feeding 4
// column-interleaved accumulators, 26 -> 31 us; per-apply streams at depth 2K,
// 32 us)
#pragma once

#include "dz_qnet_ops.h"
#include "dz_glds.h"

namespace {  // internal linkage: this header is included by several .hip files

// ------------------ forward, shared weight stream across applies -------------- //
// The learner evaluates the layer for up to three applies, two of which use the
// SAME parameter set (Rainbow: online(s_tm1) and online(s_t); only the noise
// differs).  Streaming the weights once per apply costs 77 MB and, in the
// two-GEMM noisy form, a contraction of depth 2K.  Here a workgroup streams its slice
// of one parameter SET once (mu and sigma), builds each apply's effective weight
//     W_eff[k][n] = Wmu[k][n] + Wsig[k][n] * (eps_in[k] * eps_out[n])
// in registers (2 VALU per element -- nothing next to a 64-cycle MFMA) and feeds
// one depth-K MFMA chain per apply:  y = x . W_eff  ==  x.Wmu + ((x.eps_in).Wsig).eps_out
// (networks.py:168-176) up to float32 rounding order.  Rainbow fc1: weight bytes
// 77 MB -> 51 MB (online and target once each), MFMA work halved.
// grid = (strips of 128 columns over both heads, S k-splits, parameter sets).
struct FcStreamFwd3Params {
  const float* x; int ldx; int M;      // x: [groups*M][ldx]
  int noisy;
  const float* params[2];              // parameter sets
  int ng[2];                           // applies per set (1 or 2)
  int grp[2][2];                       // their group indices
  const float* noise[DZ_MAX_GROUPS];   // noise block per group
  int G;                               // total groups (row count of a partial slab / M)
  FcHead head[2];                      // N multiple of 128
  float* part;                         // [S][G*M][ldo]
  int ldo;
  int rows_per_split;                  // multiple of 4, <= 2*NL
  int xcd_order;                       // 1: the strips of one (split, set) row block run on one XCD
};

// Groups the G applies by parameter pointer into at most 2 sets of at most 2
// applies; returns the number of sets or -1 if the applies do not fit.
static inline int dz_fc3_assign_sets(FcStreamFwd3Params& q, int G, const float* const* prm,
                                     const float* const* nz) {
  int ns = 0;
  for (int g = 0; g < G; ++g) {
    q.noise[g] = nz[g];
    int st = -1;
    for (int j = 0; j < ns; ++j)
      if (q.params[j] == prm[g] && q.ng[j] < 2) st = j;
    if (st < 0) {
      if (ns == 2) return -1;
      st = ns++; q.params[st] = prm[g]; q.ng[st] = 0;
    }
    q.grp[st][q.ng[st]++] = g;
  }
  for (int j = 0; j < ns; ++j)
    if (q.ng[j] == 1) q.grp[j][1] = q.grp[j][0];
  for (int g = G; g < DZ_MAX_GROUPS; ++g) q.noise[g] = nz[0];
  if (ns == 1) {
    q.params[1] = q.params[0]; q.ng[1] = q.ng[0];
    q.grp[1][0] = q.grp[0][0]; q.grp[1][1] = q.grp[0][1];
  }
  return ns;
}

// NL = k-pairs per lane; CH = k-pairs per pipeline chunk; DEPTH chunks in flight.
//
// Software pipeline (round 2; measured with tools/micro/fc1_micro.hip, 51 MB of
// weights, back to back):  all 100 loads of a wave issued up front 18.2 us;
// chunks of 5 k-pairs with 3 chunks in flight and the LDS operands of a chunk
// fetched one chunk ahead 14.6 us, bit-identical slabs (the bare load stream takes
// 8.6 us, the 12.6 MB of slab stores 2.3 us).  Two things were wrong with the
// up-front form:
//   * a CU's memory pipeline serves its waves' load instructions in issue order, so
//     with 100 loads queued per wave the data arrives wave by wave and the last wave
//     starts its 100-150 MFMA chain when the stream ends; with 20-30 loads in
//     flight per wave all waves advance together (60 in flight: 16.2 us, 20: 14.5);
//   * x and eps_in were read from LDS right in front of each MFMA: an
//     `s_waitcnt lgkmcnt(0)` (~100 cycles) before every one of the 64-cycle MFMAs.
template <int NOISY, int NL, int CH = 5, int DEPTH = 3>
__global__ __launch_bounds__(256) void dz_fc_stream_fwd3(FcStreamFwd3Params p) {
  extern __shared__ __attribute__((aligned(16))) float lds3[];
  constexpr int NCH = NL / CH;
  static_assert(NL % CH == 0 && NCH >= DEPTH, "chunking");
  const int R = p.rows_per_split;
  float* xs = lds3;                  // [2][R][32]  x (batch-row minor)
  float* es = lds3 + 2 * R * 32;     // [2][R]      eps_in
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int half = lane >> 5, l31 = lane & 31;
  // Workgroups go to the 8 XCDs round-robin by linear id; in XCD order all strips
  // of one row block (one contiguous 4 KB weight row per k) are fetched through
  // ONE XCD's L2 instead of eight.
  unsigned bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
  if (p.xcd_order) {
    const unsigned L = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
    const unsigned j = L >> 3, yz = (j / gridDim.x) * 8 + (L & 7);
    bx = j % gridDim.x; by = yz % gridDim.y; bz = yz / gridDim.y;
  }
  const int strips0 = p.head[0].N / 128;
  const int h_idx = (int)bx >= strips0 ? 1 : 0;
  const FcHead hd = dz_pick_head(p.head, h_idx);
  const int n0 = ((int)bx - (h_idx ? strips0 : 0)) * 128 + 32 * wave;
  const int split = (int)by, set = (int)bz;
  const float* __restrict__ prm = set ? p.params[1] : p.params[0];
  const int ng = set ? p.ng[1] : p.ng[0];
  const int g0 = set ? p.grp[1][0] : p.grp[0][0];
  const int g1 = set ? p.grp[1][1] : p.grp[0][1];
  const float* __restrict__ nz0 = dz_pick3(p.noise, g0);
  const float* __restrict__ nz1 = dz_pick3(p.noise, g1);
  const int K = hd.K;
  const int r0 = split * R;
  const int nrows = max(min(K, r0 + R) - r0, 0);
  const int ncol = n0 + l31;

  // Issue order matters: vector loads return in order.  (1) the small operands
  // first -- x, eps_in, eps_out; (2) the first DEPTH weight chunks; (3) the LDS
  // staging waits only for (1) and runs while (2) is in flight.
  const float eo0 = NOISY ? nz0[hd.eps_out + ncol] : 0.f;
  const float eo1 = NOISY ? nz1[hd.eps_out + ncol] : 0.f;
  const int mm = threadIdx.x & 31, q0 = threadIdx.x >> 5;
  constexpr int NP = (2 * NL / 4 + 7) / 8;
  float4 v[2][NP];
  float e0 = 0.f, e1 = 0.f;
  const float ok = mm < p.M ? 1.f : 0.f;  // applied at the LDS write, not here (no use of a
  {                                        // loaded value before the weight loads are issued)
    const int mc = min(mm, p.M - 1);
#pragma unroll
    for (int j = 0; j < NP; ++j) {
      const int k = min(r0 + 4 * (q0 + 8 * j), K - 4);  // r0, K multiples of 4
      v[0][j] = dz_ld4(p.x + (long)(g0 * p.M + mc) * p.ldx + hd.x_off + k);
      v[1][j] = dz_ld4(p.x + (long)(g1 * p.M + mc) * p.ldx + hd.x_off + k);
    }
    if (NOISY) {
      const int k = min(r0 + (int)threadIdx.x, K - 1);
      e0 = nz0[hd.eps_in + k]; e1 = nz1[hd.eps_in + k];
    }
  }
  __builtin_amdgcn_sched_barrier(0);  // the scheduler otherwise sinks (1) below (2)
  float wm[DEPTH][CH], wg[DEPTH][NOISY ? CH : 1];
  const float* wmu = prm + hd.w_mu + ncol;
  const float* wsg = prm + hd.w_sig + ncol;
  // chunk c: lane l holds W[r0 + 2u + (l>>5)][n0 + (l&31)] for its CH k-pairs u,
  // which IS its B operand of v_mfma_f32_32x32x2_f32
  auto issue = [&](int c, float (&m)[CH], float (&g)[NOISY ? CH : 1]) {
#pragma unroll
    for (int j = 0; j < CH; ++j) {
      const int k = min(r0 + 2 * (c * CH + j) + half, K - 1);
      m[j] = wmu[(long)k * hd.ldw];
      if (NOISY) g[j] = wsg[(long)k * hd.ldw];
    }
  };
#pragma unroll
  for (int c = 0; c < DEPTH; ++c) issue(c, wm[c], wg[c]);
  __builtin_amdgcn_sched_barrier(0);

  // stage x[g][k] (k-major, 32 batch rows minor) and eps_in[g][k]
  {
#pragma unroll
    for (int j = 0; j < NP; ++j) {
      const int q = q0 + 8 * j;
      if (4 * q < nrows) {
        float* d0 = xs + (4 * q) * 32 + mm;
        const float4 a = dz_scale4(v[0][j], ok), b = dz_scale4(v[1][j], ok);
        d0[0] = a.x; d0[32] = a.y; d0[64] = a.z; d0[96] = a.w;
        float* d1 = d0 + R * 32;
        d1[0] = b.x; d1[32] = b.y; d1[64] = b.z; d1[96] = b.w;
      }
    }
    if (NOISY && (int)threadIdx.x < R) { es[threadIdx.x] = e0; es[R + threadIdx.x] = e1; }
  }
  __syncthreads();

  f32x16 acc0, acc1;
#pragma unroll
  for (int i = 0; i < 16; ++i) { acc0[i] = 0.f; acc1[i] = 0.f; }
  // the LDS operands of a chunk: x of both applies, and the per-element noise factor
  // eps_in[k] * eps_out[n] of W_eff = Wmu + Wsig * (eps_in (x) eps_out)
  struct Ops { float a0[CH], a1[CH], f0[CH], f1[CH]; };
  auto fetch = [&](int c, Ops& o, bool two) {
#pragma unroll
    for (int j = 0; j < CH; ++j) {
      const int rl = 2 * (c * CH + j) + half;
      const int rc = min(rl, R - 1);
      const bool live = rl < nrows;
      o.a0[j] = live ? xs[rc * 32 + l31] : 0.f;
      o.f0[j] = NOISY ? es[rc] * eo0 : 0.f;
      if (two) {
        o.a1[j] = live ? xs[(R + rc) * 32 + l31] : 0.f;
        o.f1[j] = NOISY ? es[R + rc] * eo1 : 0.f;
      }
    }
  };
  Ops ops[2];
  if (ng > 1) {
    fetch(0, ops[0], true);
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      float (&m)[CH] = wm[c % DEPTH];
      float (&g)[NOISY ? CH : 1] = wg[c % DEPTH];
      if (c + 1 < NCH) fetch(c + 1, ops[(c + 1) & 1], true);
      const Ops& o = ops[c & 1];
#pragma unroll
      for (int j = 0; j < CH; ++j) {
        const float w0 = NOISY ? __builtin_fmaf(g[j], o.f0[j], m[j]) : m[j];
        const float w1 = NOISY ? __builtin_fmaf(g[j], o.f1[j], m[j]) : m[j];
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(o.a0[j], w0, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(o.a1[j], w1, acc1, 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      if (c + DEPTH < NCH) issue(c + DEPTH, m, g);  // refill the registers just consumed
      __builtin_amdgcn_sched_barrier(0);
    }
  } else {
    fetch(0, ops[0], false);
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      float (&m)[CH] = wm[c % DEPTH];
      float (&g)[NOISY ? CH : 1] = wg[c % DEPTH];
      if (c + 1 < NCH) fetch(c + 1, ops[(c + 1) & 1], false);
      const Ops& o = ops[c & 1];
#pragma unroll
      for (int j = 0; j < CH; ++j) {
        const float w0 = NOISY ? __builtin_fmaf(g[j], o.f0[j], m[j]) : m[j];
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(o.a0[j], w0, acc0, 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      if (c + DEPTH < NCH) issue(c + DEPTH, m, g);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  float* base = p.part + (long)split * p.G * p.M * p.ldo + hd.out_off + ncol;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int mrow = dz_acc_row(r, lane);
    if (mrow < p.M) {
      base[(long)(g0 * p.M + mrow) * p.ldo] = acc0[r];
      if (ng > 1) base[(long)(g1 * p.M + mrow) * p.ldo] = acc1[r];
    }
  }
}

// Ordinary global loads the COMPILER does not know about (inline assembly): next to hand-counted
// LDS-DMA waits its own `s_waitcnt vmcnt` for a known load would be vmcnt(0) -- it cannot see the
// DMA instructions behind it -- and drain the ring's first slots in front of the x staging.  The
// values are usable only behind dz_asm_landed().
typedef float dz_f4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ dz_f4v dz_asm_ld4(const float* p) {
  dz_f4v v;
  asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v) : "v"(p) : "memory");
  return v;
}
__device__ __forceinline__ float dz_asm_ld1(const float* p) {
  float v;
  asm volatile("global_load_dword %0, %1, off" : "=v"(v) : "v"(p) : "memory");
  return v;
}

// ------------------ the same stream with the weights by LDS-DMA (round 6) -------------------- //
// The register pipeline above keeps 30 dword loads (7.7 KB) per wave in flight: at 8 waves per
// CU that is 61 KB per CU, under what the memory system needs to run at its rate once the
// latency under load passes 2 us.  Here every wave streams ITS OWN 32-column slice through a
// private LDS ring: one `global_load_lds_dwordx4` fetches 8 weight rows x 128 contiguous bytes
// (4 k-pairs of one matrix), a slot = that for mu (+ sigma), NSLOT slots in flight per wave
// (NSLOT x 2 KB, no registers held).  A wave waits only for its own instructions (counted
// `s_waitcnt vmcnt`, in order) and reads only its own ring: NO barrier in the loop.  Lane
// (half, l31) reads row 2 j + half of a slot as a conflict-free ds_read_b32 -- the same B
// operand, the same k order, the same W_eff arithmetic and MFMA chains as the kernel above:
// bit-identical slabs.  LDS: ring first (1 KB aligned), then x and eps_in as above.
template <int NOISY, int NCH, int NSLOT>
__global__ __launch_bounds__(256) void dz_fc_stream_dma(FcStreamFwd3Params p) {
  extern __shared__ __attribute__((aligned(1024))) float lds3[];
  constexpr int PER = NOISY ? 2 : 1;           // DMA instructions per slot
  constexpr int SLOTF = 256 * PER;             // floats per slot
  static_assert(NCH >= NSLOT, "ring");
  const int R = p.rows_per_split;              // <= 8 NCH
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  float* ring = lds3 + wave * NSLOT * SLOTF;
  float* xs = lds3 + 4 * NSLOT * SLOTF;        // [2][R][32]  x (batch-row minor)
  float* es = xs + 2 * R * 32;                 // [2][R]      eps_in
  const unsigned ring0 = (unsigned)(uintptr_t)ring;
  const int half = lane >> 5, l31 = lane & 31;
  unsigned bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
  if (p.xcd_order) {
    const unsigned L = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
    const unsigned j = L >> 3, yz = (j / gridDim.x) * 8 + (L & 7);
    bx = j % gridDim.x; by = yz % gridDim.y; bz = yz / gridDim.y;
  }
  const int strips0 = p.head[0].N / 128;
  const int h_idx = (int)bx >= strips0 ? 1 : 0;
  const FcHead hd = dz_pick_head(p.head, h_idx);
  const int n0 = ((int)bx - (h_idx ? strips0 : 0)) * 128 + 32 * wave;
  const int split = (int)by, set = (int)bz;
  const float* __restrict__ prm = set ? p.params[1] : p.params[0];
  const int ng = set ? p.ng[1] : p.ng[0];
  const int g0 = set ? p.grp[1][0] : p.grp[0][0];
  const int g1 = set ? p.grp[1][1] : p.grp[0][1];
  const float* __restrict__ nz0 = dz_pick3(p.noise, g0);
  const float* __restrict__ nz1 = dz_pick3(p.noise, g1);
  const int K = hd.K;
  const int r0 = split * R;
  const int nrows = max(min(K, r0 + R) - r0, 0);
  const int ncol = n0 + l31;

  // (1) the small operands: assembly loads (see dz_asm_ld4), oldest in the queue
  const int mm = threadIdx.x & 31, q0 = threadIdx.x >> 5;
  constexpr int NPmax = (2 * 4 * NCH / 4 + 7) / 8;
  dz_f4v v[2][NPmax];
  float eo0 = 0.f, eo1 = 0.f, e0 = 0.f, e1 = 0.f;
  const float ok = mm < p.M ? 1.f : 0.f;
  {
    const int mc = min(mm, p.M - 1);
#pragma unroll
    for (int j = 0; j < NPmax; ++j) {
      const int k = min(r0 + 4 * (q0 + 8 * j), K - 4);
      v[0][j] = dz_asm_ld4(p.x + (long)(g0 * p.M + mc) * p.ldx + hd.x_off + k);
      v[1][j] = dz_asm_ld4(p.x + (long)(g1 * p.M + mc) * p.ldx + hd.x_off + k);
    }
    if (NOISY) {
      const int k = min(r0 + (int)threadIdx.x, K - 1);
      e0 = dz_asm_ld1(nz0 + hd.eps_in + k); e1 = dz_asm_ld1(nz1 + hd.eps_in + k);
      eo0 = dz_asm_ld1(nz0 + hd.eps_out + ncol); eo1 = dz_asm_ld1(nz1 + hd.eps_out + ncol);
    }
  }
  // (2) the ring's first NSLOT slots.  Lane L of an instruction fetches unit L & 7 of row L >> 3
  const float* wmu = prm + hd.w_mu + n0 + 4 * (lane & 7);
  const float* wsg = prm + hd.w_sig + n0 + 4 * (lane & 7);
  auto issue = [&](int c) {
    const int k = min(r0 + 8 * c + (lane >> 3), K - 1);
    const unsigned dst = ring0 + 4u * (unsigned)((c % NSLOT) * SLOTF);
    dz_glds16<0>(wmu + (long)k * hd.ldw, dst);
    if (NOISY) dz_glds16<0>(wsg + (long)k * hd.ldw, dst + 1024u);
  };
#pragma unroll
  for (int c = 0; c < NSLOT; ++c) issue(c);
  // (1) has landed when only the ring's instructions are outstanding
  asm volatile("s_waitcnt vmcnt(%0)" : : "n"(NSLOT * PER) : "memory");
#pragma unroll
  for (int j = 0; j < NPmax; ++j) asm volatile("" : "+v"(v[0][j]), "+v"(v[1][j]));
  asm volatile("" : "+v"(e0), "+v"(e1), "+v"(eo0), "+v"(eo1));

  // (3) stage x[g][k] (k-major, 32 batch rows minor) and eps_in[g][k]
  {
#pragma unroll
    for (int j = 0; j < NPmax; ++j) {
      const int q = q0 + 8 * j;
      if (4 * q < nrows) {
        float* d0 = xs + (4 * q) * 32 + mm;
        d0[0] = v[0][j].x * ok; d0[32] = v[0][j].y * ok; d0[64] = v[0][j].z * ok; d0[96] = v[0][j].w * ok;
        float* d1 = d0 + R * 32;
        d1[0] = v[1][j].x * ok; d1[32] = v[1][j].y * ok; d1[64] = v[1][j].z * ok; d1[96] = v[1][j].w * ok;
      }
    }
    if (NOISY && (int)threadIdx.x < R) { es[threadIdx.x] = e0; es[R + threadIdx.x] = e1; }
  }
  __syncthreads();

  f32x16 acc0, acc1;
#pragma unroll
  for (int i = 0; i < 16; ++i) { acc0[i] = 0.f; acc1[i] = 0.f; }
  struct Ops { float a0[4], a1[4], f0[4], f1[4]; };
  auto fetch = [&](int c, Ops& o, bool two) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int rl = 8 * c + 2 * j + half;
      const int rc = min(rl, R - 1);
      const bool live = rl < nrows;
      o.a0[j] = live ? xs[rc * 32 + l31] : 0.f;
      o.f0[j] = NOISY ? es[rc] * eo0 : 0.f;
      if (two) {
        o.a1[j] = live ? xs[(R + rc) * 32 + l31] : 0.f;
        o.f1[j] = NOISY ? es[R + rc] * eo1 : 0.f;
      }
    }
  };
  Ops ops[2];
  const bool two = ng > 1;
  fetch(0, ops[0], two);
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    // slot c has landed when at most the younger slots' instructions are outstanding
    constexpr int kDummy = 0; (void)kDummy;
    {
      const int ahead = (NCH - 1 - c) < (NSLOT - 1) ? (NCH - 1 - c) : (NSLOT - 1);
      switch (ahead * PER) {
#define DZ_FCW(i) case i: asm volatile("s_waitcnt vmcnt(" #i ")" ::: "memory"); break;
        DZ_FCW(1) DZ_FCW(2) DZ_FCW(3) DZ_FCW(4) DZ_FCW(5) DZ_FCW(6) DZ_FCW(7) DZ_FCW(8) DZ_FCW(9) DZ_FCW(10)
        DZ_FCW(11) DZ_FCW(12) DZ_FCW(13) DZ_FCW(14) DZ_FCW(15) DZ_FCW(16) DZ_FCW(17) DZ_FCW(18) DZ_FCW(19) DZ_FCW(20)
        DZ_FCW(21) DZ_FCW(22) DZ_FCW(23) DZ_FCW(24) DZ_FCW(25) DZ_FCW(26) DZ_FCW(27) DZ_FCW(28) DZ_FCW(29) DZ_FCW(30)
#undef DZ_FCW
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
      }
    }
    const float* sl = ring + (c % NSLOT) * SLOTF + half * 32 + l31;
    float m[4], g[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) { m[j] = sl[j * 64]; g[j] = NOISY ? sl[256 + j * 64] : 0.f; }
    if (c + 1 < NCH) fetch(c + 1, ops[(c + 1) & 1], two);
    const Ops& o = ops[c & 1];
    if (two) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float w0 = NOISY ? __builtin_fmaf(g[j], o.f0[j], m[j]) : m[j];
        const float w1 = NOISY ? __builtin_fmaf(g[j], o.f1[j], m[j]) : m[j];
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(o.a0[j], w0, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(o.a1[j], w1, acc1, 0, 0, 0);
      }
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float w0 = NOISY ? __builtin_fmaf(g[j], o.f0[j], m[j]) : m[j];
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(o.a0[j], w0, acc0, 0, 0, 0);
      }
    }
    if (c + NSLOT < NCH) {
      // the slot's LDS reads have returned (their values fed the MFMAs above); refill it
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      issue(c + NSLOT);
    }
  }
  float* base = p.part + (long)split * p.G * p.M * p.ldo + hd.out_off + ncol;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int mrow = dz_acc_row(r, lane);
    if (mrow < p.M) {
      base[(long)(g0 * p.M + mrow) * p.ldo] = acc0[r];
      if (two) base[(long)(g1 * p.M + mrow) * p.ldo] = acc1[r];
    }
  }
}

}  // namespace
