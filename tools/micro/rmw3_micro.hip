// The optimiser's read-modify-write of fc1's mu and sigma matrices ([3136][1056] each, inside the
// p / m / v vectors) in the TILE order adam_onfly_kernel walks them (a workgroup owns R rows x C
// columns of both matrices, 256 / (C / 4) rows per iteration, the next rows requested after the
// current rows' stores) against the flat grid-stride stream of rmw2_micro: does the access pattern
// cost anything by itself?   (tools only)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %d line %d\n", (int)e, __LINE__); exit(1); } } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));
constexpr int kRows = 3136, kLd = 1056;
__device__ __forceinline__ void upd(f4& a, f4& b, f4& c) {
  a.x += 1e-3f; b.y = b.y * 0.9f + 1.f; c.z = c.z * 0.999f + 1.f; a.w += b.y * c.z;
}
template <int R, int C, int AHEAD>
__global__ __launch_bounds__(256) void tiles(float* __restrict__ p, float* __restrict__ m, float* __restrict__ v,
                                             unsigned mu, unsigned sig) {
  constexpr int TPR = C / 4, RP = 256 / TPR, IT = R / RP, STRIPS = 1024 / C;
  const int strip = blockIdx.x % STRIPS, rg = blockIdx.x / STRIPS;
  const int tid = threadIdx.x, rl = tid / TPR, c4 = tid % TPR;
  unsigned om = mu + ((unsigned)(rg * R + rl) * kLd + strip * C + 4 * c4);
  unsigned os = sig + ((unsigned)(rg * R + rl) * kLd + strip * C + 4 * c4);
  f4 pm = *(f4*)(p + om), mm = *(f4*)(m + om), vm = *(f4*)(v + om);
  f4 ps = *(f4*)(p + os), ms = *(f4*)(m + os), vs = *(f4*)(v + os);
#pragma unroll 1
  for (int it = 0; it < IT; ++it) {
    f4 npm, nmm, nvm, nps, nms, nvs;
    const unsigned nm = om + RP * kLd, ns = os + RP * kLd;
    if (AHEAD && it + 1 < IT) {
      npm = *(f4*)(p + nm); nmm = *(f4*)(m + nm); nvm = *(f4*)(v + nm);
      nps = *(f4*)(p + ns); nms = *(f4*)(m + ns); nvs = *(f4*)(v + ns);
    }
    upd(pm, mm, vm); upd(ps, ms, vs);
    *(f4*)(m + om) = mm; *(f4*)(v + om) = vm; *(f4*)(p + om) = pm;
    *(f4*)(m + os) = ms; *(f4*)(v + os) = vs; *(f4*)(p + os) = ps;
    om = nm; os = ns;
    if (it + 1 < IT) {
      if (AHEAD) { pm = npm; mm = nmm; vm = nvm; ps = nps; ms = nms; vs = nvs; }
      else {
        pm = *(f4*)(p + om); mm = *(f4*)(m + om); vm = *(f4*)(v + om);
        ps = *(f4*)(p + os); ms = *(f4*)(m + os); vs = *(f4*)(v + os);
      }
    }
  }
}
__global__ __launch_bounds__(256) void flat(f4* __restrict__ p, f4* __restrict__ m, f4* __restrict__ v, long n4) {
  const long stride = (long)gridDim.x * 256;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
    f4 a = p[i], b = m[i], c = v[i];
    upd(a, b, c);
    p[i] = a; m[i] = b; v[i] = c;
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
  const long P = 6868480;
  const unsigned mu = 77824, sig = mu + kRows * kLd + 1024;   // (roughly where the matrices sit)
  float *p, *m, *v, *junk;
  CK(hipMalloc(&p, P * 4)); CK(hipMalloc(&m, P * 4)); CK(hipMalloc(&v, P * 4)); CK(hipMalloc(&junk, 320l << 20));
  CK(hipMemset(p, 0, P * 4)); CK(hipMemset(m, 0, P * 4)); CK(hipMemset(v, 0, P * 4));
  const long n4 = 2l * kRows * kLd / 4;   // the same bytes as the tiles (pads included)
  { auto f = [&]() { hipLaunchKernelGGL(flat, dim3(2048), dim3(256), 0, 0, (f4*)(p + mu), (f4*)(m + mu), (f4*)(v + mu), n4); };
    printf("flat grid-stride stream over the two matrices (%.1f MB moved): %6.2f us\n", n4 * 16 * 6 / 1e6, time_us(f)); }
#define RUN(R, C, A) { auto f = [&]() { hipLaunchKernelGGL((tiles<R, C, A>), dim3((kRows / R) * (1024 / C)), dim3(256), 0, 0, p, m, v, mu, sig); }; \
    printf("tiles %3d rows x %4d columns%s, %4d workgroups: %6.2f us\n", R, C, A ? " (next rows requested first)" : "", (kRows / R) * (1024 / C), time_us(f)); }
  RUN(112, 64, 0); RUN(112, 64, 1); RUN(56, 128, 0); RUN(56, 128, 1); RUN(28, 256, 0); RUN(28, 256, 1);
  RUN(14, 512, 0); RUN(14, 512, 1); RUN(7, 1024, 0); RUN(7, 1024, 1); RUN(14, 1024, 1); RUN(28, 1024, 1);
  // the same with N MB of other traffic between two launches (what the rest of a step does to the
  // 256 MB Infinity Cache): written (memset) or read (a flat read-only pass)
  for (long mb : {25l, 50l, 100l, 150l, 200l, 300l}) {
    auto g = [&]() { hipMemsetAsync(junk, 1, mb << 20, 0); };
    auto f = [&]() { hipMemsetAsync(junk, 1, mb << 20, 0);
                     hipLaunchKernelGGL((tiles<112, 64, 0>), dim3((kRows / 112) * 16), dim3(256), 0, 0, p, m, v, mu, sig); };
    const float tg = time_us(g), tf = time_us(f);
    printf("tiles 112 x 64 behind a %3ld MB memset: %6.2f us (%6.2f - %6.2f)\n", mb, tf - tg, tf, tg);
  }
  return 0;
}
