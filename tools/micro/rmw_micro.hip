// What does the memory system give a read-modify-write stream of the optimiser's shape?
// Three arrays of P floats (p, m, v) read and written in place, optionally a fourth (g)
// read only; flat float4 grid-stride, 2048 workgroups (tools only).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %d line %d\n", (int)e, __LINE__); exit(1); } } while (0)
template <int READ_G, int WRITES>
__global__ __launch_bounds__(256) void rmw(float4* __restrict__ p, float4* __restrict__ m, float4* __restrict__ v,
                                           const float4* __restrict__ g, long n4) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    float4 a = p[i], b = m[i], c = v[i];
    float4 d = READ_G ? g[i] : make_float4(1.f, 1.f, 1.f, 1.f);
    a.x += d.x * 1e-3f; b.y = b.y * 0.9f + d.y; c.z = c.z * 0.999f + d.z * d.z; a.w += b.y * c.z;
    if (WRITES >= 1) p[i] = a;
    if (WRITES >= 2) m[i] = b;
    if (WRITES >= 3) v[i] = c;
    if (WRITES == 0 && a.x == 123.f) p[i] = a;
  }
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
  const long P = 6868480, n4 = P / 4;
  float4 *p, *m, *v, *g;
  CK(hipMalloc(&p, P * 4)); CK(hipMalloc(&m, P * 4)); CK(hipMalloc(&v, P * 4)); CK(hipMalloc(&g, P * 4));
  CK(hipMemset(p, 0, P * 4)); CK(hipMemset(m, 0, P * 4)); CK(hipMemset(v, 0, P * 4)); CK(hipMemset(g, 0, P * 4));
  const double MB = P * 4 / 1e6;
#define RUN(RG, W, label) { auto f = [&]() { hipLaunchKernelGGL((rmw<RG, W>), dim3(2048), dim3(256), 0, 0, p, m, v, g, n4); }; \
    const float t = time_us(f); const double mb = (3 + RG + W) * MB; \
    printf("%-34s %6.1f MB  %6.2f us  %.2f TB/s\n", label, mb, t, mb / t); }
  RUN(0, 0, "read p,m,v");
  RUN(1, 0, "read g,p,m,v");
  RUN(0, 1, "read p,m,v  write p");
  RUN(0, 3, "read p,m,v  write p,m,v");
  RUN(1, 3, "read g,p,m,v  write p,m,v");
  return 0;
}
