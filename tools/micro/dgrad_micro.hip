// Row-owning weight stream for fc1's input gradient (dz_fc1_dgrad.h): correctness against
// a host double sum and time per launch (tools only).
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include "dz_row_dgrad.h"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %d line %d\n", (int)e, __LINE__); exit(1); } } while (0)
template <class F> float time_us(F f, int iters = 200) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 20; ++i) f();
  CK(hipDeviceSynchronize()); CK(hipEventRecord(e0, 0));
  for (int i = 0; i < iters; ++i) f();
  CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1)); return ms * 1e3f / iters;
}
template <int P>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2)))
void dgrad_kernel(RowDgrad q) {
  __shared__ __attribute__((aligned(16))) float lds[kRdLdsFloats];
  row_dgrad_block<2, 2, true, P>(q, blockIdx.x, lds);
}
int main(int argc, char** argv) {
  const int K = 3136, ld = 1056, M = argc > 1 ? atoi(argv[1]) : 32;
  std::vector<float> prm((size_t)2 * K * ld), dh1(32 * 1024), ein(2 * K), eout(1024), feat((size_t)32 * K);
  srand(5);
  auto rnd = []() { return ((rand() % 2001) - 1000) / 1000.f; };
  for (auto& v : prm) v = rnd() * 0.05f;
  for (auto& v : dh1) v = rnd();
  for (auto& v : ein) v = rnd(); for (auto& v : eout) v = rnd();
  for (auto& v : feat) v = rnd();
  if (getenv("DBG2")) {
    for (auto& v : prm) v = 0.f;
    for (size_t i = 0; i < (size_t)K * ld; ++i) prm[i] = 1.f;
    for (int b = 0; b < 32; ++b) for (int n = 0; n < 1024; ++n) dh1[b * 1024 + n] = (float)(b + 1);
    for (auto& v : feat) v = 1.f;
  }
  if (getenv("DBG3")) for (size_t i = (size_t)K * ld; i < prm.size(); ++i) prm[i] = 0.f;
  if (getenv("DBG4")) for (int b = 0; b < 32; ++b) for (int n = 0; n < 1024; ++n) dh1[b * 1024 + n] = (float)(b + 1);
  if (getenv("DBG5")) for (size_t i = 0; i < (size_t)K * ld; ++i) prm[i] = 1.f;
  float *dprm, *ddh1, *dein, *deout, *dfeat, *dout;
  CK(hipMalloc(&dprm, prm.size() * 4)); CK(hipMalloc(&ddh1, dh1.size() * 4)); CK(hipMalloc(&dein, ein.size() * 4));
  CK(hipMalloc(&deout, 4096)); CK(hipMalloc(&dfeat, feat.size() * 4)); CK(hipMalloc(&dout, feat.size() * 4));
  CK(hipMemcpy(dprm, prm.data(), prm.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(ddh1, dh1.data(), dh1.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dein, ein.data(), ein.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(deout, eout.data(), 4096, hipMemcpyHostToDevice));
  CK(hipMemcpy(dfeat, feat.data(), feat.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemset(dout, 0xff, feat.size() * 4));
  // noise block: [eps_in0 | eps_in1 | eps_out]
  float* dnz; CK(hipMalloc(&dnz, (2 * K + 1024) * 4));
  CK(hipMemcpy(dnz, dein, 2 * K * 4, hipMemcpyDeviceToDevice));
  CK(hipMemcpy(dnz + 2 * K, deout, 4096, hipMemcpyDeviceToDevice));
  RowDgrad q = {};
  q.params = dprm; q.noise = dnz;
  for (int h = 0; h < 2; ++h) {
    q.head[h].w_mu = 512 * h; q.head[h].w_sig = (long)K * ld + 512 * h; q.head[h].ldw = ld;
    q.head[h].N = 512; q.head[h].eps_in = K * h; q.head[h].eps_out = 2 * K + 512 * h; q.head[h].out_off = 512 * h;
  }
  q.dy = ddh1; q.ldy = 1024; q.mask = dfeat; q.out = dout; q.ldo = K; q.out_col[0] = 0; q.out_col[1] = 0;
  q.same_out = 1; q.M = M; q.K = K;
  for (int v = 0; v < 6; ++v) {
    const int nb = (v % 2) ? 512 : 448, P = v < 2 ? 2 : (v < 4 ? 4 : 7);
    q.nblocks = nb;
    auto f = [&]() {
      if (P == 2) hipLaunchKernelGGL(dgrad_kernel<2>, dim3(nb), dim3(256), 0, 0, q);
      else if (P == 4) hipLaunchKernelGGL(dgrad_kernel<4>, dim3(nb), dim3(256), 0, 0, q);
      else hipLaunchKernelGGL(dgrad_kernel<7>, dim3(nb), dim3(256), 0, 0, q);
    };
    f(); CK(hipDeviceSynchronize());
    std::vector<float> got((size_t)32 * K);
    CK(hipMemcpy(got.data(), dout, got.size() * 4, hipMemcpyDeviceToHost));
    double worst = 0;
    for (int b = 0; b < M; b += 5) for (int k = 0; k < K; k += 97) {
      double s = 0;
      for (int n = 0; n < 1024; ++n) {
        const float e = ein[(n < 512 ? 0 : K) + k] * eout[n];
        const float w = fmaf(prm[(size_t)K * ld + (size_t)k * ld + n], e, prm[(size_t)k * ld + n]);
        s += (double)dh1[b * 1024 + n] * w;
      }
      if (!(feat[(size_t)b * K + k] > 0.f)) s = 0;
      worst = fmax(worst, fabs(s - got[(size_t)b * K + k]));
    }
    if (nb == 448 && getenv("DBG")) {
      for (int b : {0, 1, 5, 17, 31}) { printf("b=%d:", b); for (int k : {0, 1, 2, 7, 8, 100}) {
        double s = 0;
        for (int n = 0; n < 1024; ++n) {
          const float e = ein[(n < 512 ? 0 : K) + k] * eout[n];
          s += (double)dh1[b * 1024 + n] * fmaf(prm[(size_t)K * ld + (size_t)k * ld + n], e, prm[(size_t)k * ld + n]);
        }
        printf("  [%d] %.4f/%.4f(m%d)", k, got[(size_t)b * K + k], s, feat[(size_t)b * K + k] > 0.f);
      } printf("\n"); }
    }
    printf("%4d workgroups P=%d  max |err| %.2e   %.2f us\n", nb, P, worst, time_us(f));
  }
  return 0;
}
