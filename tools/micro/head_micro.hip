// Ablation of rainbow_head_loss_kernel<1> (tools only).
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "dz_qnet_kernels.h"
int g_dz_last_hip_error = 0;
bool g_dz_prof_on = false;
void dz_prof_begin(hipStream_t) {}
void dz_prof_pair(int, int, hipStream_t) {}
void dz_prof_mark(hipStream_t, const char*) {}
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %d line %d\n", (int)e, __LINE__); exit(1); } } while (0)
template <class F> float time_us(F f, int iters = 300) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 20; ++i) f();
  CK(hipDeviceSynchronize()); CK(hipEventRecord(e0, 0));
  for (int i = 0; i < iters; ++i) f();
  CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1)); return ms * 1e3f / iters;
}
__global__ void empty32() {}
int main() {
  const int B = 32, A = 6, K = 51, NAp = 308, ld = 360, S = 8;
  float *part, *prm, *nz, *out, *dout, *losses, *prio, *qsel, *tprob, *support, *w;
  double *r, *d; int64_t* a;
  CK(hipMalloc(&part, (size_t)S * 96 * ld * 4)); CK(hipMalloc(&prm, 4096 * 4)); CK(hipMalloc(&nz, 4096 * 4));
  CK(hipMalloc(&out, 96 * ld * 4)); CK(hipMalloc(&dout, 32 * ld * 4)); CK(hipMalloc(&losses, 128));
  CK(hipMalloc(&prio, 128)); CK(hipMalloc(&qsel, 32 * 6 * 4)); CK(hipMalloc(&tprob, 32 * 51 * 4));
  CK(hipMalloc(&support, 256)); CK(hipMalloc(&w, 128)); CK(hipMalloc(&r, 256)); CK(hipMalloc(&d, 256));
  CK(hipMalloc(&a, 256));
  std::vector<float> hp((size_t)S * 96 * ld);
  srand(1); for (auto& v : hp) v = ((rand() % 2001) - 1000) / 4000.f;
  CK(hipMemcpy(part, hp.data(), hp.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemset(prm, 0, 4096 * 4)); CK(hipMemset(nz, 0, 4096 * 4));
  std::vector<float> hs(64); for (int i = 0; i < 51; ++i) hs[i] = -10.f + 0.4f * i;
  CK(hipMemcpy(support, hs.data(), 256, hipMemcpyHostToDevice));
  std::vector<float> hw(32, 0.5f); CK(hipMemcpy(w, hw.data(), 128, hipMemcpyHostToDevice));
  std::vector<double> hr(32, 1.0), hd(32, 0.97); std::vector<int64_t> ha(32, 2);
  CK(hipMemcpy(r, hr.data(), 256, hipMemcpyHostToDevice)); CK(hipMemcpy(d, hd.data(), 256, hipMemcpyHostToDevice));
  CK(hipMemcpy(a, ha.data(), 256, hipMemcpyHostToDevice));
  HeadPre pre = {};
  pre.part = part; pre.S = S; pre.rows = 96;
  for (int g = 0; g < 3; ++g) { pre.prm[g] = prm; pre.nz[g] = nz; }
  pre.b_sig = 0; pre.eps_out = 0;
  auto full = [&]() { hipLaunchKernelGGL(rainbow_head_loss_kernel<1>, dim3(B), dim3(256), (size_t)3 * ld * 4, 0,
      out, ld, NAp, B, A, K, 1, 1, 2, a, r, d, w, support, dout, losses, prio, qsel, tprob, pre); };
  auto nopre = [&]() { hipLaunchKernelGGL(rainbow_head_loss_kernel<0>, dim3(B), dim3(256), 0, 0,
      out, ld, NAp, B, A, K, 1, 1, 2, a, r, d, w, support, dout, losses, prio, qsel, tprob, pre); };
  auto empty = [&]() { hipLaunchKernelGGL(empty32, dim3(B), dim3(256), 0, 0); };
  // the wide layer's input Grams as extra workgroups of the same launch (dz_gram.h)
  float *feat, *ein; double* gpart;
  CK(hipMalloc(&feat, 32 * 3136 * 4)); CK(hipMalloc(&ein, 2 * 3136 * 4));
  CK(hipMalloc(&gpart, (size_t)3 * 7 * 4 * 64 * 4 * 8));
  std::vector<float> hf(32 * 3136); for (auto& v : hf) v = ((rand() % 2001) - 1000) / 1000.f;
  CK(hipMemcpy(feat, hf.data(), hf.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(ein, hf.data(), 2 * 3136 * 4, hipMemcpyHostToDevice));
  GramX gx; gx.x = feat; gx.M = 32; gx.K = 3136; gx.eps_in[0] = ein; gx.eps_in[1] = ein + 3136; gx.part = gpart;
  auto with_gram = [&]() { hipLaunchKernelGGL(rainbow_head_loss_kernel<1>, dim3(B + kGramXBlocks), dim3(256), (size_t)3 * ld * 4, 0,
      out, ld, NAp, B, A, K, 1, 1, 2, a, r, d, w, support, dout, losses, prio, qsel, tprob, pre, gx); };
  auto gram_only = [&]() { hipLaunchKernelGGL(rainbow_head_loss_kernel<1>, dim3(kGramXBlocks), dim3(256), (size_t)3 * ld * 4, 0,
      out, ld, NAp, 0, A, K, 1, 1, 2, a, r, d, w, support, dout, losses, prio, qsel, tprob, pre, gx); };
  printf("head_loss<1> + 84 Gram WGs %.2f us\n", time_us(with_gram));
  printf("84 Gram WGs alone          %.2f us\n", time_us(gram_only));
  printf("empty 32-WG launch         %.2f us\n", time_us(empty));
  printf("head_loss<1> (fold slabs)  %.2f us\n", time_us(full));
  printf("head_loss<0> (no fold)     %.2f us\n", time_us(nopre));
  return 0;
}
