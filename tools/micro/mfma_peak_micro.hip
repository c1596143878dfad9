// What the fp32 matrix pipe sustains on this chip: back-to-back v_mfma_f32_32x32x2_f32 with
// nothing else in the loop (no memory, no LDS), 1-8 waves per SIMD, 1-4 accumulator chains per
// wave.  Peak by the guide: 256 CUs x 4 SIMDs x 64 FLOP/cycle x 2.4 GHz = 157.3 TFLOP/s.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_peak_micro.bin mfma_peak_micro.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int CH>
__global__ __launch_bounds__(256) void mfma_loop(float* out, int iters, float a0, float b0) {
  f32x16 acc[CH];
#pragma unroll
  for (int c = 0; c < CH; ++c)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[c][i] = 0.f;
  float a = a0 + threadIdx.x * 1e-6f, b = b0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int c = 0; c < CH; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[c], 0, 0, 0);
  }
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < CH; ++c)
#pragma unroll
    for (int i = 0; i < 16; ++i) s += acc[c][i];
  if (s == 12345.678f) out[0] = s;
}

template <int CH>
static void run(int wg_per_cu, int iters) {
  float* out; hipMalloc(&out, 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int grid = 256 * wg_per_cu;
  hipLaunchKernelGGL(mfma_loop<CH>, dim3(grid), dim3(256), 0, 0, out, iters, 1.f, 1.f);
  hipDeviceSynchronize();
  float best = 1e9f;
  for (int r = 0; r < 5; ++r) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(mfma_loop<CH>, dim3(grid), dim3(256), 0, 0, out, iters, 1.f, 1.f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  const double flop = (double)grid * 4 * iters * 8 * CH * 4096.0;
  printf("chains %d  workgroups/CU %d (waves/SIMD %d)  %8.3f ms  %7.1f TFLOP/s\n", CH, wg_per_cu,
         wg_per_cu, best, flop / best / 1e9);
  hipFree(out);
}

int main() {
  for (int w : {1, 2, 4, 8}) run<1>(w, 20000 / w);
  for (int w : {1, 2, 4}) run<2>(w, 10000 / w);
  for (int w : {1, 2}) run<4>(w, 5000 / w);
  // a longer run: does the rate hold (clocks under sustained matrix load)?
  run<2>(2, 200000);
  return 0;
}
