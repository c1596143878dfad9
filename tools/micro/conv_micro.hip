#include <cstring>
#include <map>
#include <vector>
#include <algorithm>
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
__device__ unsigned long long g_stamp[8192 * 8];
#define DZ_PATCH_STAMP(i) do { if ((threadIdx.x & 63) == 0) g_stamp[((blockIdx.x + gridDim.x * blockIdx.y) * 4 + (threadIdx.x >> 6)) * 8 + (i)] = wall_clock64(); } while (0)
#include "dz_conv_patch.h"
#define DZ_C23_STAMP(i) do { if ((threadIdx.x & 63) == 0) g_stamp[((blockIdx.x + 2 * (blockIdx.y + gridDim.y * blockIdx.z)) * 8 + (threadIdx.x >> 6)) * 8 + (i)] = wall_clock64(); } while (0)
#include "dz_conv23.h"
#include "dz_conv1_persist.h"
using Conv1Patch = ConvPatchFwdOp<1, 84, 84, 4, 8, 4, 20, 20, 32, 2, 1, 2, 2>;
using Conv2Patch = ConvPatchFwdOp<0, 20, 20, 32, 4, 2, 9, 9, 64, 1, 1, 4, 2>;
using Conv3Patch = ConvPatchFwdOp<0, 9, 9, 64, 3, 1, 7, 7, 64, 1, 1, 4, 3>;
template <class Op>
__global__ __launch_bounds__(256) void conv_patch_kernel(typename Op::Params p) {
  __shared__ __attribute__((aligned(16))) float smem[Op::SMEM_ELEMS];
  Op::body(p, dim3(blockIdx.x, blockIdx.y, blockIdx.z), smem);
}

// conv3 forward, 32x64 per workgroup: K over 4 waves, two accumulators per wave sharing the
// A fragment (Op::NI = 2) -- 147 workgroups instead of 294
struct Conv3FwdNI2 : Conv3Fwd {
  static constexpr int NI = 2;
  static constexpr int BN = 64;
  __device__ static bool tile(const Params& p, const dim3& bid, Tile& t) {
    const bool ok = Conv3Fwd::tile(p, bid, t);
    t.n0 = bid.x * 64;
    return ok;
  }
};
// per-workgroup trace: where it ran and when (100 MHz wall clock)
__device__ unsigned long long g_trace[8192 * 4];
template <class Op>
__global__ __launch_bounds__(256) void gemm_trace(typename Op::Params p) {
  __shared__ __attribute__((aligned(16))) float smem[DzGemmSmem<Op>::ELEMS];
  const unsigned lin = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
  const unsigned long long t0 = wall_clock64();
  dz_gemm_body<Op>(p, dim3(blockIdx.x, blockIdx.y, blockIdx.z), smem);
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    g_trace[lin * 4 + 0] = hw; g_trace[lin * 4 + 1] = xcc;
    g_trace[lin * 4 + 2] = t0; g_trace[lin * 4 + 3] = wall_clock64();
  }
}
template <class Op>
static void show_trace(const char* name, dim3 g, typename Op::Params p) {
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((gemm_trace<Op>), g, dim3(256), 0, 0, p);
  (void)hipDeviceSynchronize();
  const int n = g.x * g.y * g.z;
  std::vector<unsigned long long> t(n * 4);
  (void)hipMemcpyFromSymbol(t.data(), HIP_SYMBOL(g_trace), n * 32);
  unsigned long long tmin = ~0ull, tmax = 0;
  for (int i = 0; i < n; ++i) { tmin = std::min(tmin, t[4 * i + 2]); tmax = std::max(tmax, t[4 * i + 3]); }
  std::map<unsigned, std::vector<int>> cus;
  double dsum = 0, smax = 0;
  for (int i = 0; i < n; ++i) {
    const unsigned hw = (unsigned)t[4 * i], xcc = (unsigned)t[4 * i + 1] & 15;
    const unsigned cu = (hw >> 8) & 15, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
    cus[(xcc << 12) | (se << 8) | (sh << 4) | cu].push_back(i);
    dsum += (t[4 * i + 3] - t[4 * i + 2]) * 0.01;
    smax = std::max(smax, (t[4 * i + 2] - tmin) * 0.01);
  }
  std::map<int, int> hist;
  for (auto& kv : cus) hist[(int)kv.second.size()]++;
  printf("%s: %d WGs on %zu CUs; first start -> last end %.2f us; mean WG duration %.2f us; latest start +%.2f us\n",
         name, n, cus.size(), (tmax - tmin) * 0.01, dsum / n, smax);
  printf("  WGs per CU histogram:");
  for (auto& kv : hist) printf("  %d:%d", kv.first, kv.second);
  printf("\n");
  int shown = 0;
  for (auto& kv : cus) {
    if (shown++ >= 6) break;
    printf("  cu %05x:", kv.first);
    for (int i : kv.second) printf("  wg%-4d [%5.2f..%5.2f]", i, (t[4 * i + 2] - tmin) * 0.01, (t[4 * i + 3] - tmin) * 0.01);
    printf("\n");
  }
}

template <class Op, int MASK>
__global__ __launch_bounds__(256) void gemm_var(typename Op::Params p) {
  __shared__ __attribute__((aligned(16))) float smem[DzGemmSmem<Op>::ELEMS];
  gemm_body_var<Op, MASK>(p, dim3(blockIdx.x, blockIdx.y, blockIdx.z), smem);
}
// staggered start: workgroups of the second "round" (linear id >= 256: the ones that
// share a CU with a first-round workgroup) begin SLEEP x 64 cycles later
template <class Op, int SLEEP>
__global__ __launch_bounds__(256) void gemm_stagger(typename Op::Params p) {
  __shared__ __attribute__((aligned(16))) float smem[DzGemmSmem<Op>::ELEMS];
  const unsigned lin = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
  if ((lin >> 8) & 1) { for (int i = 0; i < SLEEP; ++i) __builtin_amdgcn_s_sleep(15); }
  dz_gemm_body<Op>(p, dim3(blockIdx.x, blockIdx.y, blockIdx.z), smem);
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

static size_t count_diff(const float* d_a, const float* d_b, size_t n) {
  std::vector<float> a(n), b(n);
  CK(hipMemcpy(a.data(), d_a, n * 4, hipMemcpyDeviceToHost));
  CK(hipMemcpy(b.data(), d_b, n * 4, hipMemcpyDeviceToHost));
  size_t bad = 0, nz = 0;
  for (size_t i = 0; i < n; ++i) { bad += memcmp(&a[i], &b[i], 4) != 0; nz += a[i] != 0.f; }
  printf("    (%zu outputs, %.1f%% nonzero)", n, 100.0 * nz / n);
  return bad;
}
template <class T>
static void fill_dev(T* d, size_t n, int kind) {  // 0: activations >= 0 with zeros, 1: weights, 2: bytes
  std::vector<T> h(n);
  for (auto& v : h) v = kind == 2 ? (T)(rand() & 255) : kind == 1 ? (T)(((rand() % 2000) - 1000) / 22627.0f)
                                  : (T)((rand() % 1000) / 1000.0f * ((rand() & 3) ? 1.f : 0.f));
  CK(hipMemcpy(d, h.data(), n * sizeof(T), hipMemcpyHostToDevice));
}
template <class Ref, class New>
static void patch_vs_gemm(const char* name, dim3 g, ConvFwdParams p, size_t n_out) {
  float* d_alt; CK(hipMalloc(&d_alt, n_out * 4)); CK(hipMemset(d_alt, 0xff, n_out * 4));
  float* d_ref = p.out;
  hipLaunchKernelGGL((dz_mfma_gemm<Ref>), g, dim3(256), 0, 0, p);
  ConvFwdParams q = p; q.out = d_alt;
  hipLaunchKernelGGL((conv_patch_kernel<New>), g, dim3(256), 0, 0, q);
  CK(hipDeviceSynchronize());
  printf("%s patch kernel: LDS %d B", name, New::SMEM_ELEMS * 4);
  const size_t bad = count_diff(d_ref, d_alt, n_out);
  printf(" differing outputs %zu\n", bad);
  auto fr = [&]() { hipLaunchKernelGGL((dz_mfma_gemm<Ref>), g, dim3(256), 0, 0, p); };
  auto fn = [&]() { hipLaunchKernelGGL((conv_patch_kernel<New>), g, dim3(256), 0, 0, q); };
  printf("%s implicit GEMM %.2f us   patch-in-LDS %.2f us\n", name, time_us(fr), time_us(fn));
  CK(hipDeviceSynchronize());
  hipLaunchKernelGGL((conv_patch_kernel<New>), g, dim3(256), 0, 0, q);
  CK(hipDeviceSynchronize());
  {
    const int nw = g.x * g.y * 4;
    std::vector<unsigned long long> st(nw * 8);
    CK(hipMemcpyFromSymbol(st.data(), HIP_SYMBOL(g_stamp), (size_t)nw * 64));
    unsigned long long t0 = ~0ull;
    for (int i = 0; i < nw; ++i) t0 = std::min(t0, st[i * 8]);
    double mean[7] = {0};
    for (int i = 0; i < nw; ++i) for (int k = 0; k < 7; ++k) mean[k] += (st[i * 8 + k] - t0) * 0.01 / nw;
    printf("  mean stamps (us from first start): start %.2f | loads issued %.2f | patch in LDS %.2f | barrier %.2f | MFMA done %.2f | exchanged %.2f | stored %.2f\n",
           mean[0], mean[1], mean[2], mean[3], mean[4], mean[5], mean[6]);
    for (int w : {0, 1, nw / 2, nw - 1}) {
      printf("  wave %4d:", w);
      for (int k = 0; k < 7; ++k) printf(" %.2f", (st[w * 8 + k] - t0) * 0.01);
      printf("\n");
    }
  }
  CK(hipFree(d_alt));
}

__global__ void skew_probe(int spin) {
  extern __shared__ float sm[];
  const unsigned long long t0 = wall_clock64();
  if (threadIdx.x == 0) g_stamp[blockIdx.x] = t0;
  // keep the workgroup resident for a while (like a real kernel would)
  while (wall_clock64() - t0 < (unsigned long long)spin) {}
  if (spin < 0) sm[threadIdx.x] = 1.f;
}
static void probe(int wgs, int threads, int lds, int spin) {
  CK(hipFuncSetAttribute((const void*)skew_probe, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(skew_probe, dim3(wgs), dim3(threads), lds, 0, spin);
  CK(hipDeviceSynchronize());
  std::vector<unsigned long long> st(wgs);
  CK(hipMemcpyFromSymbol(st.data(), HIP_SYMBOL(g_stamp), (size_t)wgs * 8));
  unsigned long long t0 = ~0ull, t1 = 0; double mean = 0;
  for (auto v : st) { t0 = std::min(t0, v); t1 = std::max(t1, v); }
  for (auto v : st) mean += (v - t0) * 0.01 / wgs;
  printf("  start skew: %4d WGs x %4d threads, %6d B LDS, resident %d ns: mean +%.2f us, last +%.2f us\n",
         wgs, threads, lds, spin * 10, mean, (t1 - t0) * 0.01);
}

template <class Op>
static void sweep_one(const char* name, ConvFwdParams p, int CO, int G, int B) {
  const dim3 g(CO / Op::BN, G * Op::tiles_per_group(B), 1);
  auto f = [&]() { hipLaunchKernelGGL((dz_mfma_gemm<Op>), g, dim3(256), 0, 0, p); };
  printf("  %-28s %4d WGs  LDS %6d B  %.2f us\n", name, g.x * g.y, (int)DzGemmSmem<Op>::ELEMS * 4, time_us(f));
}

int main(int argc, char**) {
  const bool full = argc > 1;  // any argument: also the ablations and traces
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
  patch_vs_gemm<Conv2Fwd, Conv2Patch>("conv2", g2, p, (size_t)rows * 64);
  printf("conv2 tile sweep <WM,WN,WK,KT>:\n");
#define SW2(a, b, c_, d) sweep_one<ConvFwdOp<0, 20, 20, 32, 4, 2, 9, 9, 64, a, b, c_, d>>("<" #a "," #b "," #c_ "," #d ">", p, 64, G, B)
  SW2(1, 1, 4, 2); SW2(1, 1, 4, 1); SW2(1, 1, 4, 4); SW2(1, 2, 2, 2); SW2(1, 2, 2, 4); SW2(2, 1, 2, 2); SW2(2, 1, 2, 4); SW2(2, 2, 1, 2); SW2(2, 2, 1, 4); SW2(1, 2, 2, 1);
#define ABL(MASK, label) { auto f = [&]() { hipLaunchKernelGGL((gemm_var<Conv2Fwd, MASK>), g2, dim3(256), 0, 0, p); }; printf("conv2 %-40s %.2f us\n", label, time_us(f)); }
  if (full) ABL(0, "copy of shipped");
  if (full) ABL(1, "no global loads");
  if (full) ABL(8, "no output store");
  if (full) ABL(4, "no MFMA (VALU instead)");
  if (full) ABL(2, "no LDS / barriers");
  if (full) ABL(3, "no loads, no LDS");
  if (full) ABL(7, "no loads, no LDS, no MFMA");
  if (full) ABL(15, "nothing (launch + tile setup)");
  if (full) ABL(11, "MFMA only");
  if (full) ABL(16, "no A (im2col) loads");
  if (full) ABL(32, "no B (weight) loads");
#define STG2(S) { auto f = [&]() { hipLaunchKernelGGL((gemm_stagger<Conv2Fwd, S>), g2, dim3(256), 0, 0, p); }; printf("conv2 stagger %d x 960 cycles: %.2f us\n", S, time_us(f)); }
  if (full) { STG2(0); STG2(2); }
  if (full) show_trace<Conv2Fwd>("conv2", g2, p);
  { auto f = [&]() { hipLaunchKernelGGL((gemm_var<Conv2Fwd, 15>), dim3(1), dim3(64), 0, 0, p); }; printf("one-wave empty launch: %.2f us\n", time_us(f)); }
  {  // ---- conv1 forward (u8 input) ablation ----
    const int imgs1 = G * B;
    uint8_t* d_u8; float* d_w1; float* d_b1; float* d_o1;
    CK(hipMalloc(&d_u8, (size_t)imgs1 * 84 * 84 * 4)); CK(hipMalloc(&d_w1, 256 * 32 * 4));
    CK(hipMalloc(&d_b1, 128)); CK(hipMalloc(&d_o1, (size_t)imgs1 * 400 * 32 * 4));
    fill_dev(d_u8, (size_t)imgs1 * 84 * 84 * 4, 2); fill_dev(d_w1, 256 * 32, 1); fill_dev(d_b1, 32, 1);
    ConvFwdParams p1;
    for (int g = 0; g < G; ++g) { p1.in[g] = d_u8; p1.in_img_base[g] = g * B; p1.w[g] = d_w1; p1.bias[g] = d_b1; }
    p1.out = d_o1; p1.B = B; p1.G = G;
    const dim3 g1(32 / Conv1Fwd::BN, G * Conv1Fwd::tiles_per_group(B), 1);
    patch_vs_gemm<Conv1Fwd, Conv1Patch>("conv1", g1, p1, (size_t)imgs1 * 400 * 32);
    {
      float* d_p1; CK(hipMalloc(&d_p1, (size_t)imgs1 * 400 * 32 * 4)); CK(hipMemset(d_p1, 0xff, (size_t)imgs1 * 400 * 32 * 4));
      ConvFwdParams q1 = p1; q1.out = d_p1;
      constexpr int WPG = 85;
      CK(hipFuncSetAttribute((const void*)conv1_persist_kernel<WPG>, hipFuncAttributeMaxDynamicSharedMemorySize, conv1p::SMEM * 4));
      auto fp = [&]() { hipLaunchKernelGGL((conv1_persist_kernel<WPG>), dim3(WPG, G), dim3(256), conv1p::SMEM * 4, 0, q1); };
      auto fr = [&]() { hipLaunchKernelGGL((dz_mfma_gemm<Conv1Fwd>), g1, dim3(256), 0, 0, p1); };
      fr(); fp(); CK(hipDeviceSynchronize());
      const size_t n = (size_t)imgs1 * 400 * 32;
      std::vector<float> a(n), b(n);
      CK(hipMemcpy(a.data(), d_o1, n * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(b.data(), d_p1, n * 4, hipMemcpyDeviceToHost));
      double md = 0, mv = 0; size_t bad = 0;
      for (size_t i = 0; i < n; ++i) { const double d = fabs((double)a[i] - b[i]); if (!(d <= 1e30)) ++bad; else md = fmax(md, d); mv = fmax(mv, fabs(a[i])); }
      printf("conv1 persistent (weights in registers, %d WGs, %d B LDS): max |ref - new| = %.3g (max |ref| %.3g, non-finite %zu)\n", WPG * G, conv1p::SMEM * 4, md, mv, bad);
      printf("conv1 shipped %.2f us   persistent %.2f us\n", time_us(fr), time_us(fp));
#define PW(N) { CK(hipFuncSetAttribute((const void*)conv1_persist_kernel<N>, hipFuncAttributeMaxDynamicSharedMemorySize, conv1p::SMEM * 4)); auto f = [&]() { hipLaunchKernelGGL((conv1_persist_kernel<N>), dim3(N, G), dim3(256), conv1p::SMEM * 4, 0, q1); }; printf("  persistent, %d workgroups per group: %.2f us\n", N, time_us(f)); }
      PW(100); PW(134); PW(170); PW(200); PW(256); PW(400);
    }
    printf("conv1 tile sweep <WM,WN,WK,KT>:\n");
#define SW1(a, b, c_, d) sweep_one<ConvFwdOp<1, 84, 84, 4, 8, 4, 20, 20, 32, a, b, c_, d>>("<" #a "," #b "," #c_ "," #d ">", p1, 32, G, B)
    SW1(2, 1, 2, 2); SW1(2, 1, 2, 1); SW1(2, 1, 2, 4); SW1(1, 1, 4, 1); SW1(1, 1, 4, 2); SW1(1, 1, 4, 4); SW1(4, 1, 1, 2); SW1(4, 1, 1, 4); SW1(4, 1, 1, 1);
#define ABL1(MASK, label) { auto f = [&]() { hipLaunchKernelGGL((gemm_var<Conv1Fwd, MASK>), g1, dim3(256), 0, 0, p1); }; printf("conv1 %-40s %.2f us\n", label, time_us(f)); }
    if (full) ABL1(0, "copy of shipped");
    if (full) ABL1(1, "no global loads");
    if (full) ABL1(8, "no output store");
    if (full) ABL1(4, "no MFMA (VALU instead)");
    if (full) ABL1(2, "no LDS / barriers (no u8 conversion)");
    if (full) ABL1(11, "MFMA only");
    if (full) ABL1(15, "nothing");
    if (full) ABL1(16, "no A (u8 im2col) loads");
    if (full) ABL1(32, "no B (weight) loads");
#define STG1(S) { auto f = [&]() { hipLaunchKernelGGL((gemm_stagger<Conv1Fwd, S>), g1, dim3(256), 0, 0, p1); }; printf("conv1 stagger %d x 960 cycles: %.2f us\n", S, time_us(f)); }
    if (full) { STG1(0); STG1(2); }
    if (full) show_trace<Conv1Fwd>("conv1", g1, p1);
  }
  {  // ---- conv3 forward ablation ----
    const int rows3 = G * B * 49;
    float *d_i3, *d_w3, *d_b3, *d_o3;
    CK(hipMalloc(&d_i3, (size_t)G * B * 81 * 64 * 4)); CK(hipMalloc(&d_w3, 576 * 64 * 4));
    CK(hipMalloc(&d_b3, 256)); CK(hipMalloc(&d_o3, (size_t)rows3 * 64 * 4));
    fill_dev(d_i3, (size_t)G * B * 81 * 64, 0); fill_dev(d_w3, 576 * 64, 1); fill_dev(d_b3, 64, 1);
    ConvFwdParams p3;
    for (int g = 0; g < G; ++g) { p3.in[g] = d_i3; p3.in_img_base[g] = g * B; p3.w[g] = d_w3; p3.bias[g] = d_b3; }
    p3.out = d_o3; p3.B = B; p3.G = G;
    const dim3 g3(64 / Conv3Fwd::BN, G * Conv3Fwd::tiles_per_group(B), 1);
    patch_vs_gemm<Conv3Fwd, Conv3Patch>("conv3", g3, p3, (size_t)rows3 * 64);
    {
      float* d_alt; CK(hipMalloc(&d_alt, (size_t)rows3 * 64 * 4)); CK(hipMemset(d_alt, 0xff, (size_t)rows3 * 64 * 4));
      ConvFwdParams q3 = p3; q3.out = d_alt;
      const dim3 gn(1, G * Conv3Fwd::tiles_per_group(B), 1);
      hipLaunchKernelGGL((dz_mfma_gemm<Conv3Fwd>), g3, dim3(256), 0, 0, p3);
      hipLaunchKernelGGL((dz_mfma_gemm<Conv3FwdNI2>), gn, dim3(256), 0, 0, q3);
      CK(hipDeviceSynchronize());
      printf("conv3 NI=2 (147 WGs, LDS %d B):", (int)DzGemmSmem<Conv3FwdNI2>::ELEMS * 4);
      printf(" differing outputs %zu\n", count_diff(d_o3, d_alt, (size_t)rows3 * 64));
      auto fn = [&]() { hipLaunchKernelGGL((dz_mfma_gemm<Conv3FwdNI2>), gn, dim3(256), 0, 0, q3); };
      auto fr = [&]() { hipLaunchKernelGGL((dz_mfma_gemm<Conv3Fwd>), g3, dim3(256), 0, 0, p3); };
      printf("conv3 shipped %.2f us   NI=2 %.2f us\n", time_us(fr), time_us(fn));
    }
    printf("conv3 tile sweep <WM,WN,WK,KT>:\n");
#define SW3(a, b, c_, d) sweep_one<ConvFwdOp<0, 9, 9, 64, 3, 1, 7, 7, 64, a, b, c_, d>>("<" #a "," #b "," #c_ "," #d ">", p3, 64, G, B)
    SW3(1, 1, 4, 3); SW3(1, 1, 4, 1); SW3(1, 2, 2, 3); SW3(1, 2, 2, 2); SW3(1, 2, 2, 6); SW3(2, 1, 2, 3); SW3(2, 1, 2, 6); SW3(2, 2, 1, 3); SW3(2, 2, 1, 6); SW3(1, 2, 2, 1);
#define ABL3(MASK, label) { auto f = [&]() { hipLaunchKernelGGL((gemm_var<Conv3Fwd, MASK>), g3, dim3(256), 0, 0, p3); }; printf("conv3 %-40s %.2f us\n", label, time_us(f)); }
    if (full) ABL3(0, "copy of shipped");
    if (full) ABL3(1, "no global loads");
    if (full) ABL3(8, "no output store");
    if (full) ABL3(4, "no MFMA");
    if (full) ABL3(2, "no LDS / barriers");
    if (full) ABL3(11, "MFMA only");
    if (full) ABL3(16, "no A (im2col) loads");
    if (full) ABL3(32, "no B (weight) loads");
#define STG3(S) { auto f = [&]() { hipLaunchKernelGGL((gemm_stagger<Conv3Fwd, S>), g3, dim3(256), 0, 0, p3); }; printf("conv3 stagger %d x 960 cycles: %.2f us\n", S, time_us(f)); }
    if (full) { STG3(0); STG3(2); }
    if (full) show_trace<Conv3Fwd>("conv3", g3, p3);
  }
  printf("workgroup start skew:\n");
  probe(192, 512, 79584, 500); probe(192, 512, 16384, 500); probe(192, 256, 79584, 500); probe(192, 256, 16384, 500);
  probe(384, 256, 40000, 500); probe(486, 256, 37376, 500); probe(192, 1024, 79584, 500); probe(192, 512, 79584, 0);
  {  // ---- conv2 -> conv3 fused per (image, band) vs the two shipped launches ----
    float *a1, *w2, *b2, *w3, *b3, *act2_ref, *feat_ref, *act2_new, *feat_new;
    const size_t n1 = (size_t)G * B * 400 * 32, n2 = (size_t)G * B * 81 * 64, n3 = (size_t)G * B * 49 * 64;
    CK(hipMalloc(&a1, n1 * 4)); CK(hipMalloc(&w2, 512 * 64 * 4)); CK(hipMalloc(&b2, 256));
    CK(hipMalloc(&w3, 576 * 64 * 4)); CK(hipMalloc(&b3, 256));
    CK(hipMalloc(&act2_ref, n2 * 4)); CK(hipMalloc(&feat_ref, n3 * 4));
    CK(hipMalloc(&act2_new, n2 * 4)); CK(hipMalloc(&feat_new, n3 * 4));
    fill_dev(a1, n1, 0); fill_dev(w2, 512 * 64, 1); fill_dev(b2, 64, 1); fill_dev(w3, 576 * 64, 1); fill_dev(b3, 64, 1);
    CK(hipMemset(act2_new, 0xff, n2 * 4)); CK(hipMemset(feat_new, 0xff, n3 * 4));
    ConvFwdParams q2, q3;
    for (int g = 0; g < G; ++g) {
      q2.in[g] = a1; q2.in_img_base[g] = g * B; q2.w[g] = w2; q2.bias[g] = b2;
      q3.in[g] = act2_ref; q3.in_img_base[g] = g * B; q3.w[g] = w3; q3.bias[g] = b3;
    }
    q2.out = act2_ref; q2.B = B; q2.G = G; q3.out = feat_ref; q3.B = B; q3.G = G;
    const dim3 gg2(64 / Conv2Fwd::BN, G * Conv2Fwd::tiles_per_group(B), 1), gg3(64 / Conv3Fwd::BN, G * Conv3Fwd::tiles_per_group(B), 1);
    auto ref = [&]() {
      hipLaunchKernelGGL((dz_mfma_gemm<Conv2Fwd>), gg2, dim3(256), 0, 0, q2);
      hipLaunchKernelGGL((dz_mfma_gemm<Conv3Fwd>), gg3, dim3(256), 0, 0, q3);
    };
    Conv23Params f;
    for (int g = 0; g < G; ++g) { f.act1[g] = a1; f.img_base[g] = g * B; f.w2[g] = w2; f.b2[g] = b2; f.w3[g] = w3; f.b3[g] = b3; }
    f.act2 = act2_new; f.feat = feat_new; f.B = B; f.G = G; f.act2_groups = 7;
    CK(hipFuncSetAttribute((const void*)conv23_fused_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, conv23::SMEM * 4));
    auto fused = [&]() { hipLaunchKernelGGL(conv23_fused_kernel, dim3(2, B, G), dim3(512), conv23::SMEM * 4, 0, f); };
    ref(); fused();
    CK(hipDeviceSynchronize());
    auto maxdiff = [&](const float* da, const float* db, size_t n, const char* what) {
      std::vector<float> a(n), b(n);
      CK(hipMemcpy(a.data(), da, n * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(b.data(), db, n * 4, hipMemcpyDeviceToHost));
      double md = 0, mv = 0; size_t bad = 0;
      for (size_t i = 0; i < n; ++i) { const double d = fabs((double)a[i] - b[i]); if (!(d <= 1e30)) ++bad; else md = fmax(md, d); mv = fmax(mv, fabs(a[i])); }
      printf("  %s: max |ref - fused| = %.3g (max |ref| %.3g, non-finite %zu)\n", what, md, mv, bad);
    };
    maxdiff(act2_ref, act2_new, n2, "act2");
    maxdiff(feat_ref, feat_new, n3, "feat");
    {
      CK(hipDeviceSynchronize()); fused(); CK(hipDeviceSynchronize());
      const int nw = 2 * B * G * 8;
      std::vector<unsigned long long> st(nw * 8);
      CK(hipMemcpyFromSymbol(st.data(), HIP_SYMBOL(g_stamp), (size_t)nw * 64));
      unsigned long long t0 = ~0ull;
      for (int i = 0; i < nw; ++i) t0 = std::min(t0, st[i * 8]);
      double mean[6] = {0};
      for (int i = 0; i < nw; ++i) for (int k = 0; k < 6; ++k) mean[k] += (st[i * 8 + k] - t0) * 0.01 / nw;
      printf("  fused stamps (mean us): start %.2f | act1 in LDS %.2f | conv2 MFMAs done %.2f | act2 in LDS %.2f | conv3 MFMAs done %.2f | end %.2f\n",
             mean[0], mean[1], mean[2], mean[3], mean[4], mean[5]);
      for (int w : {0, 4, nw / 2, nw - 1}) { printf("  wave %4d:", w); for (int k = 0; k < 6; ++k) printf(" %.2f", (st[w * 8 + k] - t0) * 0.01); printf("\n"); }
    }
    printf("conv2 + conv3, two launches %.2f us   fused (192 WGs x 512 threads, %d B LDS) %.2f us\n",
           time_us(ref), conv23::SMEM * 4, time_us(fused));
  }
  return 0;
}