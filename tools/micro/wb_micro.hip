// Priority write-back walk: level-by-level (set_leaves_and_ancestors) vs the LDS walk
// (set_leaves_and_ancestors_fast), standalone timing + in-kernel stamps + equality.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I include tools/micro/wb_micro.hip -o tools/micro/wb_micro.bin
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <hip/hip_runtime.h>
__device__ long long* g_stamps;
#define DZ_WB_STAMP(k) do { if (threadIdx.x == 0 && g_stamps) g_stamps[k] = wall_clock64(); } while (0)
#include "../../dqn_zoo_amd/csrc/dz_sumtree_dev.h"
int g_dz_last_hip_error = 0;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %d line %d\n", (int)e, __LINE__); exit(1); } } while (0)

template <int FAST>
__global__ __launch_bounds__(256) void wb_kernel(PrioUpdateParams q) {
  __shared__ WbScratch wb;
  PrioUpdateSideT<FAST>::run(q, 0, &wb, (int)sizeof(wb));
}

int main() {
  const int64_t cap = 1 << 20, N = 1000000;
  const int n = 32, R = 200;
  std::vector<double> tree(2 * cap, 0.0);
  srand(1);
  for (int64_t i = 0; i < N; ++i) tree[cap + i] = 0.5 + (rand() % 1000) / 500.0;
  for (int64_t i = cap - 1; i >= 1; --i) tree[i] = tree[2 * i] + tree[2 * i + 1];
  double *d_a, *d_b, *d_max; int64_t* d_ids; float* d_p; uint32_t* d_st; long long* d_stamps;
  CK(hipMalloc(&d_a, 2 * cap * 8)); CK(hipMalloc(&d_b, 2 * cap * 8)); CK(hipMalloc(&d_max, 8));
  CK(hipMalloc(&d_ids, R * n * 8)); CK(hipMalloc(&d_p, R * n * 4)); CK(hipMalloc(&d_st, 4));
  CK(hipMalloc(&d_stamps, 64 * 8));
  CK(hipMemcpy(d_a, tree.data(), 2 * cap * 8, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_b, tree.data(), 2 * cap * 8, hipMemcpyHostToDevice));
  CK(hipMemset(d_max, 0, 8)); CK(hipMemset(d_st, 0, 4)); CK(hipMemset(d_stamps, 0, 64 * 8));
  std::vector<int64_t> ids(R * n); std::vector<float> pr(R * n);
  for (int i = 0; i < R * n; ++i) { ids[i] = (i % 7 == 3) ? ids[i - 1] : rand() % N; pr[i] = (rand() % 10000) / 100.0f; }
  CK(hipMemcpy(d_ids, ids.data(), R * n * 8, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_p, pr.data(), R * n * 4, hipMemcpyHostToDevice));
  long long* null_stamps = nullptr;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int fast = 0; fast < 2; ++fast) {
    double* tr = fast ? d_b : d_a;
    CK(hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), &null_stamps, 8));
    for (int rep = 0; rep < 2; ++rep) {  // second pass timed (same data order both kernels)
      if (rep == 1) CK(hipMemcpy(tr, tree.data(), 2 * cap * 8, hipMemcpyHostToDevice));
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0));
      for (int r = 0; r < R; ++r) {
        PrioUpdateParams q = {tr, cap, N, N, 5 * N, d_ids + r * n, d_p + r * n, 1, 0.5, n, d_max, d_st, 0};
        if (fast) hipLaunchKernelGGL(wb_kernel<1>, dim3(1), dim3(256), 0, 0, q);
        else hipLaunchKernelGGL(wb_kernel<0>, dim3(1), dim3(256), 0, 0, q);
      }
      CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (rep == 1) printf("%s: %.2f us per launch (back to back)\n", fast ? "fast" : "slow", ms * 1e3 / R);
    }
    if (fast) {
      CK(hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), &d_stamps, 8));
      PrioUpdateParams q = {tr, cap, N, N, 5 * N, d_ids, d_p, 1, 0.5, n, d_max, d_st, 0};
      hipLaunchKernelGGL(wb_kernel<1>, dim3(1), dim3(256), 0, 0, q);
      CK(hipDeviceSynchronize());
      long long st[64]; CK(hipMemcpy(st, d_stamps, 64 * 8, hipMemcpyDeviceToHost));
      for (int k = 1; k < 64 && st[k]; ++k) printf("  stamp %d: +%.2f us\n", k, (st[k] - st[0]) / 100.0);
    }
  }
  std::vector<double> ta(2 * cap), tb(2 * cap);
  CK(hipMemcpy(ta.data(), d_a, 2 * cap * 8, hipMemcpyDeviceToHost));
  CK(hipMemcpy(tb.data(), d_b, 2 * cap * 8, hipMemcpyDeviceToHost));
  // the fast tree got one extra (stamped) launch of batch 0: apply it to the slow tree too
  { PrioUpdateParams q = {d_a, cap, N, N, 5 * N, d_ids, d_p, 1, 0.5, n, d_max, d_st, 0};
    hipLaunchKernelGGL(wb_kernel<0>, dim3(1), dim3(256), 0, 0, q); CK(hipDeviceSynchronize());
    CK(hipMemcpy(ta.data(), d_a, 2 * cap * 8, hipMemcpyDeviceToHost)); }
  printf("trees %s\n", memcmp(ta.data(), tb.data(), 2 * cap * 8) == 0 ? "IDENTICAL" : "DIFFER");
  return 0;
}
