// The IQN learner's three large contractions on the LDS-DMA GEMM mainloop (dz_dma_gemm.h), at
// the reference sizes (iqn/run_atari.py:98-100, batch 32 x 64 / 64 / 64 taus: 6 144 = 2 048 online + 4 096
// target rows, 3 136 -> 512; round 5 tuned 5 120 = 2 048 + 3 072 rows, a 32-tau policy, which stays covered):
//   forward   h1 = relu(head_in @ W1 + b1)        A = head_in (depth-contiguous), B = W1 (output-contiguous)
//   weight gradient  dW1[k][n] = sum_m head_in[m][k] dh1[m][n]      both operands output-contiguous
//   input gradient   dhin[m][k] = sum_n dh1[m][n] W1[k][n]          both operands depth-contiguous
//                    (+ the backward of the mix in its store: IqnDgradMixOp's arithmetic)
// Tile sets and what they replaced: dz_iqn.hip (iqn_head_forward, dz_iqn_learn).
#pragma once

#include "dz_dma_gemm.h"
#include "dz_iqn_ops.h"

namespace {

// out[i][j] = relu(acc + bias[j])
struct IqnFwdEpi {
  struct Params { const float* bias; float* out; long ldo; };
  __device__ static void store(const Params& q, int i0, int j0, int lane, const f32x16& acc) {
    const int col = j0 + (lane & 31);
    const float b = q.bias[col];
    float* o = q.out + (long)i0 * q.ldo + col;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float v = acc[r] + b;
      o[(long)dz_acc_row(r, lane) * q.ldo] = v > 0.f ? v : 0.f;
    }
  }
};

// out[i][j] = acc
struct IqnPlainEpi {
  struct Params { float* out; long ldo; };
  __device__ static void store(const Params& q, int i0, int j0, int lane, const f32x16& acc) {
    float* o = q.out + (long)i0 * q.ldo + j0 + (lane & 31);
#pragma unroll
    for (int r = 0; r < 16; ++r) o[(long)dz_acc_row(r, lane) * q.ldo] = acc[r];
  }
};

// The input gradient's store with the backward of the mix (head_in = temb * feat[b],
// networks.py:285) -- samples a multiple of 32, so that the 32 rows of a block belong to ONE batch
// element b = i0 / samples:
//   dzt[row][c]  = (head_in[row][c] > 0) ? dhin[row][c] * feat[b][c] : 0        -> dx
//   s1[blk][c]   = sum over the block's 32 rows of dhin * head_in                  (blk = row / 32)
//   s2[blk][c]   = sum over the block's 32 rows of dzt
// (dfeat[b][c] = (feat > 0) * sum_blk s1 / feat, folded by IqnBwdSide; the embedding bias
// gradient = sum_blk s2, folded by reduce_jobs_kernel.)  head_in and the feature factor are loaded
// here, behind the mainloop's last DMA wait.
struct IqnDgradMixEpi {
  struct Params { float* dx; long ldo; const float* hin; const float* feat; int samples; float* s1; float* s2; };
  __device__ static void store(const Params& q, int i0, int j0, int lane, const f32x16& acc) {
    const int col = j0 + (lane & 31);
    const float f = q.feat[(long)(i0 / q.samples) * q.ldo + col];
    float e[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) e[r] = q.hin[(long)(i0 + dz_acc_row(r, lane)) * q.ldo + col];
    float s1 = 0.f, s2 = 0.f;
    float* o = q.dx + (long)i0 * q.ldo + col;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float d = acc[r];
      s1 += d * e[r];
      const float dz = e[r] > 0.f ? d * f : 0.f;
      s2 += dz;
      o[(long)dz_acc_row(r, lane) * q.ldo] = dz;
    }
    s1 += __shfl_xor(s1, 32);   // the other 16 rows of the block live in lane ^ 32
    s2 += __shfl_xor(s2, 32);
    if (lane < 32) {
      q.s1[(long)(i0 >> 5) * q.ldo + col] = s1;
      q.s2[(long)(i0 >> 5) * q.ldo + col] = s2;
    }
  }
};

}  // namespace
