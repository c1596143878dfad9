// What does a grid-wide barrier cost on this chip (8 XCDs, per-XCD L2)?  One persistent
// launch of G workgroups runs N barriers; each barrier = agent-scope release, one atomic
// arrival on a counter, spin on the generation word (bounded: a stuck barrier aborts
// instead of hanging the box), agent-scope acquire.  Between barriers every workgroup
// writes a few KB and reads what ANOTHER workgroup wrote before the barrier (so that the
// fences have real work to do and correctness is checked).  (tools only)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %d line %d\n", (int)e, __LINE__); exit(1); } } while (0)
struct Bar { unsigned count; unsigned gen; unsigned fail; };
__device__ __forceinline__ bool grid_barrier(Bar* b, unsigned nblocks, unsigned& my_gen) {
  __syncthreads();
  bool ok = true;
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    const unsigned target = my_gen + 1;
    if (__hip_atomic_fetch_add(&b->count, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == nblocks - 1) {
      __hip_atomic_store(&b->count, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(&b->gen, target, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      long spins = 0;
      while (__hip_atomic_load(&b->gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != target) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > 2000000) { ok = false; __hip_atomic_store(&b->fail, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  }
  my_gen += 1;
  __syncthreads();
  return ok;
}
template <int WORK>
__global__ __launch_bounds__(256) void k(Bar* b, float* buf, int n, unsigned* bad) {
  unsigned gen = 0;
  const unsigned nb = gridDim.x;
  for (int it = 0; it < n; ++it) {
    if (WORK) {
      // write my 4 KB, barrier, read the neighbour's 4 KB written in this iteration
      for (int j = threadIdx.x; j < 1024; j += 256) buf[(size_t)blockIdx.x * 1024 + j] = (float)(it * 7 + j);
    }
    if (!grid_barrier(b, nb, gen)) return;
    if (WORK) {
      const unsigned other = (blockIdx.x + 37) % nb;
      float s = 0.f;
      for (int j = threadIdx.x; j < 1024; j += 256) s += buf[(size_t)other * 1024 + j] - (float)(it * 7 + j);
      if (s != 0.f) atomicAdd(bad, 1u);
      if (!grid_barrier(b, nb, gen)) return;   // (before the next iteration overwrites)
    }
  }
}
int main(int argc, char** argv) {
  const int G = argc > 1 ? atoi(argv[1]) : 256, N = 200;
  Bar* b; float* buf; unsigned* bad;
  CK(hipMalloc(&b, sizeof(Bar))); CK(hipMalloc(&buf, (size_t)G * 4096)); CK(hipMalloc(&bad, 4));
  for (int work = 0; work < 2; ++work) {
    CK(hipMemset(b, 0, sizeof(Bar))); CK(hipMemset(bad, 0, 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0, 0));
    if (work) hipLaunchKernelGGL(k<1>, dim3(G), dim3(256), 0, 0, b, buf, N, bad);
    else hipLaunchKernelGGL(k<0>, dim3(G), dim3(256), 0, 0, b, buf, N, bad);
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    Bar hb; unsigned hbad; CK(hipMemcpy(&hb, b, sizeof(Bar), hipMemcpyDeviceToHost)); CK(hipMemcpy(&hbad, bad, 4, hipMemcpyDeviceToHost));
    const int nbar = work ? 2 * N : N;
    printf("%d workgroups, %s: %.2f us per barrier (%d barriers)  fail=%u  stale reads=%u\n", G,
           work ? "4 KB written + neighbour's 4 KB read between barriers" : "barriers only", ms * 1e3 / nbar, nbar, hb.fail, hbad);
  }
  return 0;
}
