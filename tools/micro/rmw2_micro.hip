// Variants of the optimiser-shaped read-modify-write stream (tools/micro/rmw_micro.hip): does the
// kind of load / store instruction, the grid size or the number of units in flight per lane move
// the 30.3 us it takes to read and write back p, m, v (165 MB)?   (tools only)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %d line %d\n", (int)e, __LINE__); exit(1); } } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));
template <int LD, int ST, int U>
__global__ __launch_bounds__(256) void rmw(f4* __restrict__ p, f4* __restrict__ m, f4* __restrict__ v, long n4) {
  const long stride = (long)gridDim.x * 256;
  for (long i0 = (long)blockIdx.x * 256 + threadIdx.x; i0 < n4; i0 += U * stride) {
    f4 a[U], b[U], c[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long i = i0 + u * stride < n4 ? i0 + u * stride : n4 - 1;
      if (LD == 1) { a[u] = __builtin_nontemporal_load(p + i); b[u] = __builtin_nontemporal_load(m + i); c[u] = __builtin_nontemporal_load(v + i); }
      else { a[u] = p[i]; b[u] = m[i]; c[u] = v[i]; }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long i = i0 + u * stride;
      a[u].x += 1e-3f; b[u].y = b[u].y * 0.9f + 1.f; c[u].z = c[u].z * 0.999f + 1.f; a[u].w += b[u].y * c[u].z;
      if (i < n4) {
        if (ST == 1) { __builtin_nontemporal_store(a[u], p + i); __builtin_nontemporal_store(b[u], m + i); __builtin_nontemporal_store(c[u], v + i); }
        else { p[i] = a[u]; m[i] = b[u]; v[i] = c[u]; }
      }
    }
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
  f4 *p, *m, *v;
  CK(hipMalloc(&p, P * 4)); CK(hipMalloc(&m, P * 4)); CK(hipMalloc(&v, P * 4));
  CK(hipMemset(p, 0, P * 4)); CK(hipMemset(m, 0, P * 4)); CK(hipMemset(v, 0, P * 4));
#define RUN(LD, ST, U, G) { auto f = [&]() { hipLaunchKernelGGL((rmw<LD, ST, U>), dim3(G), dim3(256), 0, 0, p, m, v, n4); }; \
    printf("loads %-3s stores %-3s  %d units per lane  %5d workgroups: %6.2f us\n", LD ? "nt" : "", ST ? "nt" : "", U, G, time_us(f)); }
  RUN(0, 0, 1, 2048); RUN(0, 0, 1, 4096); RUN(0, 0, 1, 1024); RUN(0, 0, 2, 2048); RUN(0, 0, 2, 1024); RUN(0, 0, 4, 1024);
  RUN(1, 0, 1, 2048); RUN(0, 1, 1, 2048); RUN(1, 1, 1, 2048); RUN(1, 1, 2, 2048); RUN(0, 1, 2, 1024);
  return 0;
}
