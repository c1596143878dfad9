// What does a grid-wide barrier cost on this chip (8 XCDs, per-XCD L2)?  One persistent
// launch of G workgroups runs N barriers; between barriers every workgroup writes 4 KB and
// reads what ANOTHER workgroup wrote before the barrier (so that the publish has real work
// to do and staleness is detected).  Forms (tools only):
//   flat/fence : plain stores, agent-scope release fence, ONE counter + generation word,
//                agent-scope acquire fence                        (rounds 1-3: 11.4-12.6 us)
//   xcd/fence  : the same publish, XCD-hierarchical arrival (MI355X_MICROARCH.md
//                "barrier-xcd"): per-XCC counter -> the XCC's last arriver bumps the top
//                counter -> the top's last arriver bumps the generation of every XCC
//   xcd/sc1+acq: payload stored write-through (sc1), every wave drains vmcnt, NO release fence;
//                readers issue an agent-scope ACQUIRE fence and then PLAIN loads
//   xcd/sc1    : payload stored write-through (sc1), every wave drains vmcnt, NO release
//                fence; readers use sc1 loads, NO acquire fence (Guideline 16 R1)
// Every spin is bounded: a stuck barrier aborts instead of hanging the box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %d line %d\n", (int)e, __LINE__); exit(1); } } while (0)
#define RLX __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT
struct Bar { unsigned count; unsigned gen; unsigned fail; unsigned pad[13];
             unsigned xcount[8][16]; unsigned xgen[8][16]; unsigned xn[8][16]; };
__device__ __forceinline__ unsigned xcc_id() {
  unsigned v; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v)); return v & 15u;
}
__device__ __forceinline__ bool spin_until(unsigned* w, unsigned target, Bar* b) {
  long spins = 0;
  while (__hip_atomic_load(w, RLX) != target) {
    __builtin_amdgcn_s_sleep(1);
    if (++spins > 2000000) { __hip_atomic_store(&b->fail, 1u, RLX); return false; }
  }
  return true;
}
// MODE 0: flat + fences; 1: hierarchical + fences; 2: hierarchical, no fences (sc1 payload)
template <int MODE>
__device__ __forceinline__ bool grid_barrier(Bar* b, unsigned nblocks, unsigned& my_gen, unsigned xcc) {
  if (MODE >= 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every storing wave drains
  __syncthreads();
  bool ok = true;
  if (threadIdx.x == 0) {
    if (MODE < 2) { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
    const unsigned target = my_gen + 1;
    if (MODE == 0) {
      if (__hip_atomic_fetch_add(&b->count, 1u, RLX) == nblocks - 1) {
        __hip_atomic_store(&b->count, 0u, RLX);
        __hip_atomic_store(&b->gen, target, RLX);
      } else ok = spin_until(&b->gen, target, b);
    } else {
      const unsigned mine = __hip_atomic_load(&b->xn[xcc][0], RLX);   // blocks on my XCC (census)
      if (__hip_atomic_fetch_add(&b->xcount[xcc][0], 1u, RLX) == mine - 1) {
        __hip_atomic_store(&b->xcount[xcc][0], 0u, RLX);
        if (__hip_atomic_fetch_add(&b->count, mine, RLX) == nblocks - mine) {
          __hip_atomic_store(&b->count, 0u, RLX);
          for (int x = 0; x < 8; ++x) __hip_atomic_store(&b->xgen[x][0], target, RLX);
        } else ok = spin_until(&b->xgen[xcc][0], target, b);
      } else ok = spin_until(&b->xgen[xcc][0], target, b);
    }
    if (MODE < 2) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  }
  my_gen += 1;
  __syncthreads();
  if (MODE == 3) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");   // every wave, then plain loads
  return ok;
}
__global__ void census(Bar* b) { if (threadIdx.x == 0) atomicAdd(&b->xn[xcc_id() & 7][0], 1u); }
template <int MODE, int WORK>
__global__ __launch_bounds__(256) void k(Bar* b, float* buf, int n, unsigned* bad) {
  unsigned gen = 0;
  const unsigned nb = gridDim.x, xcc = xcc_id() & 7;
  for (int it = 0; it < n; ++it) {
    if (WORK) {   // write my 4 KB
      for (int j = threadIdx.x; j < 1024; j += 256) {
        float* p = buf + (size_t)blockIdx.x * 1024 + j;
        if (MODE >= 2) __hip_atomic_store(p, (float)(it * 7 + j), RLX); else *p = (float)(it * 7 + j);
      }
    }
    if (!grid_barrier<MODE>(b, nb, gen, xcc)) return;
    if (WORK) {   // read the neighbour's 4 KB written in this iteration
      const unsigned other = (blockIdx.x + 37) % nb;
      float s = 0.f;
      for (int j = threadIdx.x; j < 1024; j += 256) {
        const float* p = buf + (size_t)other * 1024 + j;
        s += (MODE == 2 ? __hip_atomic_load(p, RLX) : *p) - (float)(it * 7 + j);
      }
      if (s != 0.f) atomicAdd(bad, 1u);
      if (!grid_barrier<MODE>(b, nb, gen, xcc)) return;   // (before the next iteration overwrites)
    }
  }
}
template <int MODE> static void run(int G, Bar* b, float* buf, unsigned* bad, const char* name) {
  const int N = 200;
  for (int work = 0; work < 2; ++work) {
    CK(hipMemset(b, 0, sizeof(Bar))); CK(hipMemset(bad, 0, 4));
    // census: how many blocks of a G-wide grid land on each XCC (dispatch is round-robin by
    // block id; the hierarchical barrier only needs the COUNTS to be those of its own launch,
    // so a mismatch shows up as fail=1, never as a hang)
    hipLaunchKernelGGL(census, dim3(G), dim3(256), 0, 0, b);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0, 0));
    if (work) hipLaunchKernelGGL((k<MODE, 1>), dim3(G), dim3(256), 0, 0, b, buf, N, bad);
    else hipLaunchKernelGGL((k<MODE, 0>), dim3(G), dim3(256), 0, 0, b, buf, N, bad);
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    Bar hb; unsigned hbad; CK(hipMemcpy(&hb, b, sizeof(Bar), hipMemcpyDeviceToHost)); CK(hipMemcpy(&hbad, bad, 4, hipMemcpyDeviceToHost));
    const int nbar = work ? 2 * N : N;
    printf("%-10s %4d workgroups, %-28s: %6.2f us per barrier  fail=%u stale=%u\n", name, G,
           work ? "4 KB out + 4 KB in per phase" : "barriers only", ms * 1e3 / nbar, hb.fail, hbad);
  }
}
int main(int argc, char** argv) {
  const int G = argc > 1 ? atoi(argv[1]) : 256;
  Bar* b; float* buf; unsigned* bad;
  CK(hipMalloc(&b, sizeof(Bar))); CK(hipMalloc(&buf, (size_t)G * 4096)); CK(hipMalloc(&bad, 4));
  run<0>(G, b, buf, bad, "flat/fence");
  run<1>(G, b, buf, bad, "xcd/fence");
  run<2>(G, b, buf, bad, "xcd/sc1");
  run<3>(G, b, buf, bad, "xcd/sc1+acq");
  return 0;
}
