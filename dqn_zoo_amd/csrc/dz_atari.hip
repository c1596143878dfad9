// Atari observation preprocessing on the device (SURVEY.md 8f row f4;
// ref: dqn_zoo/processors.py:367-387 rgb2y / resize, :486-505 the observation
// branch of atari()): max-pool the last raw frames, grayscale, Pillow-BILINEAR
// resize, append to the frame stack, emit the stacked observation -- ONE launch,
// the frame stack and the observation never leave HBM.
//
// Bit-exactness (pinned by the reference's own sha256 golden,
// processors_test.py:472-475, through oracle/processors_oracle.py):
//   * rgb2y = (r*0.299 + g*0.587) + b*(1-(0.299+0.587)) in IEEE float64, left to
//     right, NO fused multiply-add (this file is compiled with -ffp-contract=off
//     like the rest of the library), truncated to uint8;
//   * the resize is Pillow's two-pass 8-bit resample: integer arithmetic on the
//     22-bit fixed-point coefficient tables the host computes once
//     (dqn_zoo_amd/processors.py::resample_coeffs), horizontal pass first, uint8
//     intermediate, out = clip8((2^21 + sum px*k) >> 22).
//
// One workgroup per output row: it grayscales the <= ksize input rows that row
// needs into LDS, resamples them horizontally, then vertically.  An input row is
// needed by ~out/in*ksize (2.8) output rows, so ~100 K pixels are converted per
// 84x84 frame: integer/byte work, latency-bound at this size (one launch ~5 us).
#include "dz_common.h"

namespace {

constexpr int kPrecisionBits = 32 - 8 - 2;
constexpr int kMaxTaps = 64;      // ksize limit (7 for 210 -> 84)
constexpr int kMaxPooled = 4;     // frames max-pooled per observation
constexpr int kMaxStack = 8;      // stacked frames per observation

struct AtariParams {
  const uint8_t* frames[kMaxPooled];  // [H][W][C] each
  int n_frames, H, W, C;              // C = 3 (RGB) or 1 (already gray)
  int planes;                         // 1: one output plane (C = 3: grayscaled); 3: RGB kept
                                      // (grayscaling=False: every band resampled on its own,
                                      // as Pillow does for mode "RGB")
  const int32_t* xb; const int32_t* xk; int xks;   // horizontal [OW][2], [OW][xks]
  const int32_t* yb; const int32_t* yk; int yks;   // vertical   [OH][2], [OH][yks]
  int OH, OW;
  uint8_t* ring;     // [stack][OH][OW][planes]
  int stack;         // ring depth = stacked frames
  int slot;          // ring slot the new frame goes to
  int count;         // frames in the stack INCLUDING the new one (1..stack)
  uint8_t* obs;      // [OH][OW][planes][stack] ([OH][OW][stack] for one plane), oldest first, trailing zeros
};

__device__ __forceinline__ uint8_t clip8(int v) {
  v >>= kPrecisionBits;  // arithmetic shift, as Pillow's clip8 lookup
  return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
}

__global__ __launch_bounds__(256) void atari_observation_kernel(AtariParams p) {
  extern __shared__ uint8_t lds[];
  uint8_t* gray = lds;                       // [yks][W]
  uint8_t* hres = lds + p.yks * p.W;         // [yks][OW]
  const int yy = blockIdx.x;
  const int band = blockIdx.y;               // < planes
  const bool to_gray = p.C == 3 && p.planes == 1;
  const int ymin = p.yb[2 * yy], ycnt = p.yb[2 * yy + 1];
  const int tid = threadIdx.x;
  // ---- max-pool + grayscale of the rows this output row needs ----
  for (int i = tid; i < ycnt * p.W; i += 256) {
    const int r = i / p.W, x = i - r * p.W;
    const long px = ((long)(ymin + r) * p.W + x) * p.C;
    int c0 = 0, c1 = 0, c2 = 0;
#pragma unroll
    for (int f = 0; f < kMaxPooled; ++f) {
      if (f < p.n_frames) {
        const uint8_t* s = p.frames[f] + px;
        c0 = max(c0, (int)s[to_gray ? 0 : band]);
        if (to_gray) { c1 = max(c1, (int)s[1]); c2 = max(c2, (int)s[2]); }
      }
    }
    uint8_t g = (uint8_t)c0;
    if (to_gray) {
      const double c2w = 1 - (0.299 + 0.587);   // processors.py:370, folded like Python
      double y = (double)c0 * 0.299;
      y = y + (double)c1 * 0.587;
      y = y + (double)c2 * c2w;
      g = (uint8_t)(int)y;                      // astype(uint8): truncation
    }
    gray[r * p.W + x] = g;
  }
  __syncthreads();
  // ---- horizontal pass ----
  for (int i = tid; i < ycnt * p.OW; i += 256) {
    const int r = i / p.OW, xx = i - r * p.OW;
    const int xmin = p.xb[2 * xx], n = p.xb[2 * xx + 1];
    const int32_t* k = p.xk + (long)xx * p.xks;
    int acc = 1 << (kPrecisionBits - 1);
    for (int x = 0; x < n; ++x) acc += (int)gray[r * p.W + xmin + x] * k[x];
    hres[r * p.OW + xx] = clip8(acc);
  }
  __syncthreads();
  // ---- vertical pass + ring write + stacked observation ----
  for (int xx = tid; xx < p.OW; xx += 256) {
    const int32_t* k = p.yk + (long)yy * p.yks;
    int acc = 1 << (kPrecisionBits - 1);
    for (int r = 0; r < ycnt; ++r) acc += (int)hres[r * p.OW + xx] * k[r];
    const uint8_t v = clip8(acc);
    const long o = ((long)yy * p.OW + xx) * p.planes + band;
    const long plane_sz = (long)p.OH * p.OW * p.planes;
    p.ring[(long)p.slot * plane_sz + o] = v;
    // stack position j holds the (count-1-j)-th newest frame; j >= count: zero pad
    for (int j = 0; j < p.stack; ++j) {
      uint8_t s = 0;
      if (j == p.count - 1) {
        s = v;
      } else if (j < p.count - 1) {
        int sl = p.slot - (p.count - 1 - j);
        sl = sl < 0 ? sl + p.stack : sl;
        s = p.ring[(long)sl * plane_sz + o];
      }
      p.obs[o * p.stack + j] = s;
    }
  }
}

}  // namespace

extern "C" int dz_atari_observation(const uint8_t* const* frames, int n_frames, int height,
                                    int width, int channels, int grayscale, const int32_t* xbounds,
                                    const int32_t* xcoeffs, int xksize, const int32_t* ybounds,
                                    const int32_t* ycoeffs, int yksize, int out_h, int out_w,
                                    uint8_t* ring, int stack, int slot, int count,
                                    uint8_t* obs, dz_stream_t stream) {
  DZ_REQUIRE(frames && n_frames >= 0 && n_frames <= kMaxPooled);  // 0: black frame
  DZ_REQUIRE(height > 0 && width > 0 && (channels == 3 || channels == 1));
  DZ_REQUIRE(xbounds && xcoeffs && ybounds && ycoeffs && ring && obs);
  DZ_REQUIRE(xksize >= 1 && xksize <= kMaxTaps && yksize >= 1 && yksize <= kMaxTaps);
  DZ_REQUIRE(out_h > 0 && out_w > 0 && stack >= 1 && stack <= kMaxStack);
  DZ_REQUIRE(slot >= 0 && slot < stack && count >= 1 && count <= stack);
  AtariParams p = {};
  for (int f = 0; f < n_frames; ++f) { DZ_REQUIRE(frames[f]); p.frames[f] = frames[f]; }
  p.n_frames = n_frames; p.H = height; p.W = width; p.C = channels;
  p.planes = (channels == 3 && !grayscale) ? 3 : 1;
  p.xb = xbounds; p.xk = xcoeffs; p.xks = xksize;
  p.yb = ybounds; p.yk = ycoeffs; p.yks = yksize;
  p.OH = out_h; p.OW = out_w; p.ring = ring; p.stack = stack; p.slot = slot; p.count = count;
  p.obs = obs;
  const size_t lds = (size_t)yksize * (size_t)(width + out_w);
  DZ_REQUIRE(lds <= 64 * 1024);
  hipLaunchKernelGGL(atari_observation_kernel, dim3((unsigned)out_h, (unsigned)p.planes), dim3(256), lds,
                     dz_s(stream), p);
  DZ_LAUNCH_CHECK();
  return DZ_OK;
}
