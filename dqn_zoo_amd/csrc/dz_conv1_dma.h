// conv1 forward (uint8 frames, 8x8 stride 4, 4 -> 32 channels; ref: dqn_zoo/networks.py:193-195:
// `x / 255` then hk.Conv2D(32, [8, 8], 4)) with both operands by LDS-DMA and NO reduction across
// waves:
//   * a workgroup owns BMW = 32 NRB consecutive output pixels x all 32 channels x the whole depth
//     K = 256; its four waves = 2 row groups x 2 column blocks of 16, a wave = NRB accumulator
//     chains of `v_mfma_f32_16x16x4_f32` (NRB row blocks of 16 pixels sharing the wave's B
//     fragments).  No accumulator exchange, no split-K: 240 workgroups at NRB = 5 for the three
//     applies of a batch of 32 -- one round on 256 CUs, 1 280 MFMAs per workgroup.
//   * A = the im2col rows as RAW BYTES: a kernel row is 32 contiguous bytes (8 pixels x 4
//     channels), a DMA lane fetches 16 of them; LDS holds [4 stages][rows][64 B] with the 16-byte
//     units of a row XOR-swizzled by (row >> 2) & 3 so that the lanes (i = row, q = depth slot) of a
//     `ds_read_b64` (8 depth indices per lane: k = 32 c + 8 q + t) hit 64 distinct banks.  The conversion uint8 -> float -> exact `/ 255`
//     (dz_div255) happens at the fragment read: with 32 output channels = one column pair, every
//     input byte is converted twice (once per column block's wave) instead of being written to
//     LDS as 4 bytes and read back.
//   * B = the 32 KB filter bank [256][32], all of it resident, rows r and r + 8 stored in
//     opposite halves of the banks (source-side swap of the 16-float column blocks).
//   * all 4 K-stages (64 deep: two kernel rows) are requested at t = 0; stage s is consumed behind
//     a counted `s_waitcnt vmcnt` + one barrier; nothing is overwritten, so there is no second
//     barrier per stage.
//   * epilogue: bias + ReLU, the tile transposed through LDS (row pitch 36 floats), then 16-byte
//     fully coalesced stores (the workgroup's 32 NRB x 32 outputs are one contiguous block).
// Whole tiles only (B * 400 a multiple of BMW); other batch sizes keep ConvFwdOp.
#pragma once

#include "dz_dma_gemm.h"
#include "dz_qnet_ops.h"

namespace {

typedef float f32x4v __attribute__((ext_vector_type(4)));

// AHEAD_: K-stages requested up front; stage s + AHEAD_ is requested behind stage s's barrier.
// (All four at once queue every co-resident workgroup's 40 KB in front of anyone's SECOND stage:
// in-kernel stamps showed the launch's first MFMA 1.8 us after the workgroups start.)
// BRING_ = 1 (with AHEAD_ <= 2): only THREE 8 KB filter-bank stages are resident -- stage s + 2 is
// requested behind stage s's barrier into the buffer stage s - 1 was read from -- i.e. 32 KB of LDS
// per workgroup at NRB = 1, NRG = 2 instead of 40: FIVE workgroups per CU, so that all 1 200
// workgroups of the learner's launch are resident at once instead of 1 024 + a second round of 176
// (in-kernel stamps: the second round starts 5.8 us after the first).  MEASURED SLOWER, kept off:
// 13.0-13.3 us against 11.5 (bit-identical outputs) -- five workgroups sharing a CU's matrix pipe and
// memory queue lose more than the second round costs; three per CU (LDS padded): 11.5, two: 13.2.
template <int NRB_, int NRG_ = 2, int AHEAD_ = 4, int BRING_ = 0>
struct Conv1DmaCfg {
  static constexpr int AHEAD = AHEAD_, BRING = BRING_;
  static_assert(!BRING_ || AHEAD_ <= 2, "the ring's third buffer is the one stage s - 1 was read from");
  static constexpr int NRB = NRB_, NRG = NRG_, NW = 2 * NRG_, THREADS = 64 * NW, BMW = 16 * NRB_ * NRG_;
  static constexpr int H = 84, W = 84, OH = 20, OW = 20, CO = 32, K = 256, NSTG = 4;
  static constexpr int ROWB = W * 4;                       // bytes per input row
  static constexpr int A_STAGE_B = BMW * 64, A_BYTES = 4 * A_STAGE_B;
  static constexpr int AI = BMW / 16;                      // A instructions per stage
  static constexpr int BI = 8;                             // B instructions per stage (64 rows x 128 B)
  static constexpr int IPS = AI + BI;
  static constexpr int B_BYTES = (BRING_ ? 3 : 4) * 64 * CO * 4;
  static constexpr int OUT_PITCH = 36;
  static constexpr int LDS_BYTES = A_BYTES + B_BYTES;
  static_assert(BMW * OUT_PITCH * 4 <= A_BYTES, "the output tile fits in the A region");
  static int tiles_per_group(int B) { return B * OH * OW / BMW; }
  static bool fits(int B) { return (B * OH * OW) % BMW == 0; }
};

struct Conv1DmaParams {
  const uint8_t* in[DZ_MAX_GROUPS];   // per group: [B][84][84][4]
  const float* w[DZ_MAX_GROUPS];      // [256][32]
  const float* bias[DZ_MAX_GROUPS];
  float* out;                         // [G*B][20][20][32]
  int B, G;
  long long* dbg = nullptr;           // (DZ_GEMM_STAMPS builds) per-workgroup wall-clock stamps
};
#ifdef DZ_GEMM_STAMPS
#define DZ_C1_STAMP(i) do { if (p.dbg && threadIdx.x == 0) p.dbg[(long)(4 * mt) * 8 + (i)] = wall_clock64(); } while (0)
#else
#define DZ_C1_STAMP(i) do {} while (0)
#endif

// s_waitcnt vmcnt(n) for a wave-uniform run-time n (the immediate must be a constant)
__device__ __forceinline__ void dz_wait_vmcnt(int n) {
  switch (n) {
#define DZ_VMC(i) case i: asm volatile("s_waitcnt vmcnt(" #i ")" ::: "memory"); break;
    DZ_VMC(1) DZ_VMC(2) DZ_VMC(3) DZ_VMC(4) DZ_VMC(5) DZ_VMC(6) DZ_VMC(7) DZ_VMC(8) DZ_VMC(9) DZ_VMC(10)
    DZ_VMC(11) DZ_VMC(12) DZ_VMC(13) DZ_VMC(14) DZ_VMC(15) DZ_VMC(16) DZ_VMC(17) DZ_VMC(18) DZ_VMC(19)
    DZ_VMC(20) DZ_VMC(21) DZ_VMC(22) DZ_VMC(23) DZ_VMC(24)
#undef DZ_VMC
    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
  }
}

template <class C>
__device__ __forceinline__ void dz_conv1_dma_body(const Conv1DmaParams& p, int mt, unsigned char* lds) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int rg = wave >> 1, cb = wave & 1;   // rg < NRG
  const int li = lane & 15, kq = lane >> 4;
  const unsigned lds0 = (unsigned)(uintptr_t)lds;
  unsigned char* As = lds;
  const float* Bs0 = (const float*)(lds + C::A_BYTES);

  DZ_C1_STAMP(0);
  const int tpg = p.B * C::OH * C::OW / C::BMW;
  const int g = mt / tpg;
  const int m0 = (mt - g * tpg) * C::BMW;
  const uint8_t* in = dz_pick3(p.in, g);
  const float* wgt = dz_pick3(p.w, g);
  const float* bias = dz_pick3(p.bias, g);
  // (the only ordinary global load: issued before the first DMA, covered by the first wait)
  const float bcol = bias[cb * 16 + li];

  // ---- every DMA instruction of the tile, stage-major; wave w issues the A instructions
  // w, w + 4, ... (16 rows each) and the B instructions w, w + 4 (8 filter rows each) of a stage.
  // Lane addresses are formed ONCE (stage s adds a wave-uniform offset).
  constexpr int NAJ = (C::AI + C::NW - 1) / C::NW, NBJ = (C::BI + C::NW - 1) / C::NW;
  const int na = (C::AI - wave + C::NW - 1) / C::NW;      // this wave's A instructions per stage
  const int nb = (C::BI - wave + C::NW - 1) / C::NW;      // ... and B instructions
  const int nper = na + nb;
  const uint8_t* asrc[NAJ];
#pragma unroll
  for (int j = 0; j < NAJ; ++j) {
    const int row = min(16 * (C::NW * j + wave), C::BMW - 16) + (lane >> 2);
    const int ul = (lane & 3) ^ ((row >> 2) & 3);
    const int m = m0 + row;
    const int img = m / (C::OH * C::OW), pix = m - img * (C::OH * C::OW);
    const int oh = pix / C::OW, ow = pix - oh * C::OW;
    asrc[j] = in + ((long)img * C::H + 4 * oh + (ul >> 1)) * C::ROWB + 16 * ow + 16 * (ul & 1);
  }
  const float* bsrc[NBJ];
#pragma unroll
  for (int j = 0; j < NBJ; ++j) {
    const int k = min(8 * (C::NW * j + wave), 56) + (lane >> 3);
    bsrc[j] = wgt + (long)k * C::CO + 4 * ((lane & 7) ^ (4 * ((k >> 3) & 1)));
  }
  auto issue = [&](int s) {
#pragma unroll
    for (int j = 0; j < NAJ; ++j)
      if (j < na)
        dz_glds16<0>((const float*)(asrc[j] + s * 2 * C::ROWB),
                     lds0 + (unsigned)(s * C::A_STAGE_B) + (unsigned)((C::NW * j + wave) * 1024));
#pragma unroll
    for (int j = 0; j < NBJ; ++j)
      if (j < nb)
        dz_glds16<0>(bsrc[j] + s * 64 * C::CO,
                     lds0 + (unsigned)(C::A_BYTES + (C::BRING ? s % 3 : s) * 64 * 128) +
                         (unsigned)((C::NW * j + wave) * 1024));
  };
#pragma unroll
  for (int s = 0; s < C::AHEAD && s < C::NSTG; ++s) issue(s);

  DZ_C1_STAMP(1);
  f32x4v acc[C::NRB];
#pragma unroll
  for (int rb = 0; rb < C::NRB; ++rb) acc[rb] = f32x4v{0.f, 0.f, 0.f, 0.f};

  const int slot_x = (li >> 2) & 3;
#pragma unroll
  for (int s = 0; s < C::NSTG; ++s) {
    {  // stages s + 1 .. min(s + AHEAD, NSTG) - 1 may still be in flight
      const int hi = s + C::AHEAD < C::NSTG ? s + C::AHEAD : C::NSTG;
      dz_wait_vmcnt(nper * (hi - 1 - s));
    }
    __syncthreads();
    if (s == 0) DZ_C1_STAMP(2);
    if (s + C::AHEAD < C::NSTG) issue(s + C::AHEAD);
    // (indexed by the global k below: the ring's buffer of stage s starts 64 s rows earlier)
    const float* Bs = Bs0 + (C::BRING ? ((s % 3) - s) * 64 * C::CO : 0);
    // (the ring re-reads LDS addresses an earlier stage read, rewritten in between by DMA
    // instructions the compiler cannot see through -- it merged stage 3's fragment reads with stage
    // 0's in this fully unrolled loop, "memory" clobbers notwithstanding: the base goes opaque)
    if (C::BRING) asm volatile("" : "+v"(Bs));
#pragma unroll
    for (int c = 0; c < 2; ++c) {   // 32 depth indices: lane (i, q) holds k = 64 s + 32 c + 8 q + t, t < 8
      float bf[8];
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        const int k = 64 * s + 32 * c + 8 * kq + t;
        bf[t] = Bs[k * C::CO + ((cb * 16 + li) ^ (16 * (kq & 1)))];
      }
      uint2 aw[C::NRB];
#pragma unroll
      for (int rb = 0; rb < C::NRB; ++rb) {
        const int r = (rg * C::NRB + rb) * 16 + li;
        aw[rb] = *(const uint2*)(As + s * C::A_STAGE_B + r * 64 + (((2 * c + (kq >> 1)) ^ slot_x) << 4) + 8 * (kq & 1));
      }
#pragma unroll
      for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int rb = 0; rb < C::NRB; ++rb) {
          const unsigned w32 = t < 4 ? aw[rb].x : aw[rb].y;
          const float a = dz_div255((float)((w32 >> (8 * (t & 3))) & 0xffu));
          acc[rb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bf[t], acc[rb], 0, 0, 0);
        }
    }
  }
  DZ_C1_STAMP(3);
  // ---- bias + ReLU, transposed through LDS, 16-byte coalesced stores ---------------------
  __syncthreads();   // every wave has finished reading A
  float* Os = (float*)lds;
  float b = bcol;
  asm volatile("" : "+v"(b));
#pragma unroll
  for (int rb = 0; rb < C::NRB; ++rb)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float v = acc[rb][r] + b;
      Os[((rg * C::NRB + rb) * 16 + 4 * kq + r) * C::OUT_PITCH + cb * 16 + li] = v > 0.f ? v : 0.f;
    }
  __syncthreads();
  DZ_C1_STAMP(4);
  float* out = p.out + ((long)g * tpg * C::BMW + m0) * C::CO;
  static_assert((C::BMW * 8) % C::THREADS == 0, "whole float4 rounds");
#pragma unroll
  for (int j = 0; j < C::BMW * 8 / C::THREADS; ++j) {
    const int idx = tid + C::THREADS * j;     // float4 index in the tile: row = idx / 8, unit = idx % 8
    const float4 v = *(const float4*)(Os + (idx >> 3) * C::OUT_PITCH + 4 * (idx & 7));
    *(float4*)(out + 4 * (long)idx) = v;
  }
  DZ_C1_STAMP(5);
}

// the conv tiles first, then `Side` blocks (noise draw / seam clear / cosine table: jobs conv1
// does not depend on; they inherit this kernel's LDS allocation and must not use LDS)
template <class C, class Side>
__global__ __launch_bounds__(C::THREADS) void dz_conv1_dma_kernel(Conv1DmaParams p, unsigned ntiles,
                                                                 typename Side::Params sp) {
  __shared__ __attribute__((aligned(1024))) unsigned char lds[C::LDS_BYTES];
  if (blockIdx.x < ntiles) dz_conv1_dma_body<C>(p, (int)blockIdx.x, lds);
  else if (C::THREADS == 256 || threadIdx.x < 256) Side::run(sp, blockIdx.x - ntiles);   // (side jobs are 256-thread blocks)
}
struct DzNoSide {
  struct Params { int unused; };
  __device__ static void run(const Params&, unsigned) {}
};
template <class C, class Side>
static inline int dz_launch_conv1_dma(const Conv1DmaParams& p, const typename Side::Params& sp,
                                      unsigned side_blocks, hipStream_t s) {
  const unsigned nt = (unsigned)(p.G * C::tiles_per_group(p.B));
  static_assert(C::THREADS >= 256, "side jobs are 256-thread blocks");
  hipLaunchKernelGGL((dz_conv1_dma_kernel<C, Side>), dim3(nt + side_blocks), dim3(C::THREADS), 0, s, p, nt, sp);
  DZ_LAUNCH_CHECK();
  return DZ_OK;
}

}  // namespace
