// How fast can the fc1 weight-stream access patterns pull 51 MB?  (tools only)
#include <cstdio>
#include <cstdlib>
#include <hip/hip_runtime.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %d line %d\n", (int)e, __LINE__); exit(1); } } while (0)

// (a) the shipped pattern: grid (8 strips, 32 splits, 2 sets), wave = 32 columns,
// lane loads W[k + half][n0 + l31] (dword) for NL k-pairs of mu and sigma.
template <int NL>
__global__ __launch_bounds__(256) void stream_dword(const float* __restrict__ w, int ld, long set_stride,
                                                    long sig_off, float* __restrict__ out) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, half = lane >> 5, l31 = lane & 31;
  const float* p = w + blockIdx.z * set_stride + (long)(blockIdx.y * 2 * NL + half) * ld +
                   blockIdx.x * 128 + wave * 32 + l31;
  float a[NL], b[NL];
#pragma unroll
  for (int u = 0; u < NL; ++u) { a[u] = p[(long)2 * u * ld]; b[u] = p[sig_off + (long)2 * u * ld]; }
  float s = 0.f;
#pragma unroll
  for (int u = 0; u < NL; ++u) s += a[u] * b[u];
  if (s == 123.456f) out[0] = s;
}
// (b) float4 per lane: wave = 2 k-rows x 128 columns per instruction
template <int NL>
__global__ __launch_bounds__(256) void stream_x4(const float* __restrict__ w, int ld, long set_stride,
                                                 long sig_off, float* __restrict__ out) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, half = lane >> 5, l31 = lane & 31;
  // same bytes per workgroup: 128 columns x (2*NL) rows; wave w takes rows [w*NL/2 ...)
  const float* p = w + blockIdx.z * set_stride + (long)(blockIdx.y * 2 * NL + wave * (NL / 2) + half) * ld +
                   blockIdx.x * 128 + 4 * l31;
  float4 a[NL / 4], b[NL / 4];
#pragma unroll
  for (int u = 0; u < NL / 4; ++u) {
    a[u] = *(const float4*)(p + (long)2 * u * ld);
    b[u] = *(const float4*)(p + sig_off + (long)2 * u * ld);
  }
  float s = 0.f;
#pragma unroll
  for (int u = 0; u < NL / 4; ++u) s += a[u].x * b[u].x + a[u].y * b[u].y + a[u].z * b[u].z + a[u].w * b[u].w;
  if (s == 123.456f) out[0] = s;
}
// (c) flat float4 grid-stride copy-like read of the same bytes (Adam-style)
__global__ __launch_bounds__(256) void stream_flat(const float4* __restrict__ w, long n4, float* __restrict__ out) {
  float s = 0.f;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    const float4 v = w[i]; s += v.x + v.y + v.z + v.w;
  }
  if (s == 123.456f) out[0] = s;
}
template <class F> float time_us(F f, int iters = 100) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 10; ++i) f();
  CK(hipDeviceSynchronize()); CK(hipEventRecord(e0, 0));
  for (int i = 0; i < iters; ++i) f();
  CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1)); return ms * 1e3f / iters;
}
int main() {
  const int ld = 1056, K = 3200;                 // 32 splits x 100 rows
  const long mat = (long)K * ld;                 // one matrix (mu or sigma)
  const long set_stride = 2 * mat + 4096;        // online / target sets
  float* w; float* out; float* junk;
  CK(hipMalloc(&w, (2 * set_stride) * 4)); CK(hipMalloc(&out, 64));
  CK(hipMemset(w, 0, (2 * set_stride) * 4));
  const size_t junk_bytes = 600u << 20;          // evict L2 + MALL between runs
  CK(hipMalloc(&junk, junk_bytes));
  const double mb = 2.0 * 2 * 3200 * 1024 * 4 / 1e6;
  auto flush = [&]() { CK(hipMemsetAsync(junk, 1, junk_bytes, 0)); };
  for (int cold = 0; cold < 2; ++cold) {
    auto A = [&]() { if (cold) flush(); hipLaunchKernelGGL(stream_dword<50>, dim3(8, 32, 2), dim3(256), 0, 0, w, ld, set_stride, mat, out); };
    auto B = [&]() { if (cold) flush(); hipLaunchKernelGGL(stream_x4<50 - 2>, dim3(8, 32, 2), dim3(256), 0, 0, w, ld, set_stride, mat, out); };
    auto C = [&]() { if (cold) flush(); hipLaunchKernelGGL(stream_flat, dim3(2048), dim3(256), 0, 0, (const float4*)w, (long)(mb * 1e6 / 16), out); };
    auto Z = [&]() { if (cold) flush(); };
    const float tz = cold ? time_us(Z, 30) : 0.f;
    const float ta = time_us(A, cold ? 30 : 100) - tz, tb = time_us(B, cold ? 30 : 100) - tz, tc = time_us(C, cold ? 30 : 100) - tz;
    printf("%s  %.1f MB: dword/lane %.2f us (%.2f TB/s) | float4/lane %.2f us (%.2f TB/s) | flat float4 %.2f us (%.2f TB/s)\n",
           cold ? "cold (600 MB memset between)" : "warm (back to back)", mb, ta, mb / ta, tb, mb * 0.96 / tb, tc, mb / tc);
  }
  return 0;
}
