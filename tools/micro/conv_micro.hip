// Micro-benchmark / prototype bench for the weights-stationary forward convolution
// (tools only, not part of the library).  Builds against the library's headers so
// that the shipped ConvFwdOp kernels can be timed and used as the reference.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I include \
//         -I dqn_zoo_amd/csrc tools/micro/conv_micro.hip -o tools/micro/conv_micro.bin
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>
#include "dz_qnet_kernels.h"

int g_dz_last_hip_error = 0;
bool g_dz_prof_on = false;
void dz_prof_begin(hipStream_t) {}
void dz_prof_pair(int, int, hipStream_t) {}
void dz_prof_mark(hipStream_t, const char*) {}

#include "gemm_body_var.inc"
template <class Op, int MASK>
__global__ __launch_bounds__(256) void gemm_var(typename Op::Params p) {
  __shared__ __attribute__((aligned(16))) float smem[DzGemmSmem<Op>::ELEMS];
  gemm_body_var<Op, MASK>(p, dim3(blockIdx.x, blockIdx.y, blockIdx.z), smem);
}
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %d at %s:%d\n", (int)e, __FILE__, __LINE__); exit(1); } } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));

// ---------------------------------------------------------------------------
// Weights-stationary direct conv (f32 NHWC input):
//   one wave = one (16 output pixels) x (32 output channels) x full K item,
//   A fragments straight from global memory (one float4 per lane per 16-chunk),
//   the column group's weights resident in LDS for the whole workgroup,
//   v_mfma_f32_16x16x4_f32, two accumulator chains per wave, no barrier after
//   the weight staging.
// ---------------------------------------------------------------------------
template <int H, int W, int C, int KS, int S, int OH, int OW, int CO>
__global__ __launch_bounds__(256) void conv_ws_kernel(const float* __restrict__ in,
                                                      const float* __restrict__ w,
                                                      const float* __restrict__ bias,
                                                      float* __restrict__ out, int rows) {
  constexpr int K = KS * KS * C;
  constexpr int NCH = K / 16;          // 16-deep chunks
  constexpr int ROWK = KS * C;         // contiguous floats per kernel row
  static_assert(ROWK % 16 == 0 && CO % 32 == 0, "shape");
  constexpr int CG = CO / 32;
  extern __shared__ __attribute__((aligned(16))) float ws[];  // [NCH][32][16] swizzled
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int l15 = lane & 15, kq = lane >> 4;
  const int cg = blockIdx.x % CG;
  const int rt = (blockIdx.x / CG) * 4 + wave;   // 16-row tile of this wave
  const int r0 = rt * 16;
  // ---- A loads first (they are the long pole), all in flight ----
  const int m = min(r0 + l15, rows - 1);
  const int img = m / (OH * OW), pix = m % (OH * OW);
  const int oh = pix / OW, ow = pix % OW;
  const float* arow = in + (((long)img * H + oh * S) * W + ow * S) * C + 4 * kq;
  f32x4 a[NCH];
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int kh = (c * 16) / ROWK, o = (c * 16) % ROWK;
    a[c] = *(const f32x4*)(arow + (long)kh * W * C + o);
  }
  // ---- stage this column group's weights: LDS[(c*32 + n)*16 + 4*((kl>>2)^((n>>1)&3)) + (kl&3)]
  for (int i = threadIdx.x; i < K * 8; i += 256) {   // float4 along n: K rows x 8 quads
    const int k = i >> 3, nq = i & 7;
    const f32x4 v = *(const f32x4*)(w + (long)k * CO + cg * 32 + 4 * nq);
    const int c = k >> 4, kl = k & 15;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = 4 * nq + j;
      ws[(c * 32 + n) * 16 + 4 * ((kl >> 2) ^ ((n >> 1) & 3)) + (kl & 3)] = v[j];
    }
  }
  __syncthreads();
  f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
  const int swz = 4 * (kq ^ ((l15 >> 1) & 3));
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const f32x4 b0 = *(const f32x4*)(ws + (c * 32 + l15) * 16 + swz);
    const f32x4 b1 = *(const f32x4*)(ws + (c * 32 + 16 + l15) * 16 + swz);
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[c][s], b0[s], acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[c][s], b1[s], acc1, 0, 0, 0);
    }
  }
  if (r0 >= rows) return;
  const int col = cg * 32 + l15;
  const float bb0 = bias[col], bb1 = bias[col + 16];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = r0 + 4 * kq + r;
    if (row < rows) {
      const float v0 = acc0[r] + bb0, v1 = acc1[r] + bb1;
      out[(long)row * CO + col] = v0 > 0.f ? v0 : 0.f;
      out[(long)row * CO + col + 16] = v1 > 0.f ? v1 : 0.f;
    }
  }
}

template <class F>
float time_us(F f, int iters = 200) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 20; ++i) f();
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0, 0));
  for (int i = 0; i < iters; ++i) f();
  CK(hipEventRecord(e1, 0));
  CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  return ms * 1e3f / iters;
}

int main() {
  const int G = 3, B = 32;
  // conv2 geometry
  const int imgs = G * B, rows = imgs * 81;
  std::vector<float> h_in((size_t)imgs * 20 * 20 * 32), h_w(512 * 64), h_b(64);
  srand(1);
  for (auto& v : h_in) v = (rand() % 1000) / 1000.0f * ((rand() & 3) ? 1.f : 0.f);
  for (auto& v : h_w) v = ((rand() % 2000) - 1000) / 22627.0f;
  for (auto& v : h_b) v = ((rand() % 2000) - 1000) / 22627.0f;
  float *d_in, *d_w, *d_b, *d_ref, *d_new;
  CK(hipMalloc(&d_in, h_in.size() * 4)); CK(hipMalloc(&d_w, h_w.size() * 4));
  CK(hipMalloc(&d_b, 256)); CK(hipMalloc(&d_ref, (size_t)rows * 64 * 4));
  CK(hipMalloc(&d_new, (size_t)rows * 64 * 4));
  CK(hipMemcpy(d_in, h_in.data(), h_in.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_w, h_w.data(), h_w.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_b, h_b.data(), 256, hipMemcpyHostToDevice));
  ConvFwdParams p;
  for (int g = 0; g < G; ++g) { p.in[g] = d_in; p.in_img_base[g] = g * B; p.w[g] = d_w; p.bias[g] = d_b; }
  p.out = d_ref; p.B = B; p.G = G;
  auto run_ref = [&]() {
    dz_launch_gemm<Conv2Fwd>(p, dim3(64 / Conv2Fwd::BN, G * Conv2Fwd::tiles_per_group(B), 1), 0);
  };
  const int rtiles = (rows + 15) / 16;
  const int wgs = ((rtiles + 3) / 4) * 2;
  const size_t lds = 512 * 32 * 4;
  CK(hipFuncSetAttribute((const void*)conv_ws_kernel<20, 20, 32, 4, 2, 9, 9, 64>,
                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  auto run_new = [&]() {
    hipLaunchKernelGGL((conv_ws_kernel<20, 20, 32, 4, 2, 9, 9, 64>), dim3(wgs), dim3(256), lds, 0,
                       d_in, d_w, d_b, d_new, rows);
  };
  run_ref(); run_new();
  CK(hipDeviceSynchronize());
  std::vector<float> a((size_t)rows * 64), b((size_t)rows * 64);
  CK(hipMemcpy(a.data(), d_ref, a.size() * 4, hipMemcpyDeviceToHost));
  CK(hipMemcpy(b.data(), d_new, b.size() * 4, hipMemcpyDeviceToHost));
  double maxd = 0, maxv = 0; size_t nz = 0;
  for (size_t i = 0; i < a.size(); ++i) { maxd = fmax(maxd, fabs(a[i] - b[i])); maxv = fmax(maxv, fabs(a[i])); nz += a[i] != 0; }
  printf("conv2: max |ref-new| = %.3g (max |ref| %.3g, nonzero %.1f%%)\n", maxd, maxv, 100.0 * nz / a.size());
  printf("conv2 shipped kernel : %.2f us\n", time_us(run_ref));
  printf("conv2 weights-stationary (%d WGs): %.2f us\n", wgs, time_us(run_new));
  const dim3 g2(64 / Conv2Fwd::BN, G * Conv2Fwd::tiles_per_group(B), 1);
#define ABL(MASK, label) { auto f = [&]() { hipLaunchKernelGGL((gemm_var<Conv2Fwd, MASK>), g2, dim3(256), 0, 0, p); }; printf("conv2 %-40s %.2f us\n", label, time_us(f)); }
  ABL(0, "copy of shipped");
  ABL(1, "no global loads");
  ABL(8, "no output store");
  ABL(4, "no MFMA (VALU instead)");
  ABL(2, "no LDS / barriers");
  ABL(3, "no loads, no LDS");
  ABL(7, "no loads, no LDS, no MFMA");
  ABL(15, "nothing (launch + tile setup)");
  ABL(11, "MFMA only");
  { auto f = [&]() { hipLaunchKernelGGL((gemm_var<Conv2Fwd, 15>), dim3(1), dim3(64), 0, 0, p); }; printf("one-wave empty launch: %.2f us\n", time_us(f)); }
  return 0;
}
