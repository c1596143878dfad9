// Shared host-side helpers for libdqnzoo_hip.so (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/dqnzoo_hip.h"

// The kernels are written for ONE target: 160 KB of LDS per workgroup (fc1's input gradient
// holds 135 KB), `global_load_lds_dwordx4`, the gfx950 lane-swap instructions, hand-counted
// `s_waitcnt` around LDS-DMA.  A device pass for anything else must not compile.
#if defined(__HIP_DEVICE_COMPILE__) && __HIP_DEVICE_COMPILE__ && !defined(__gfx950__)
#error "libdqnzoo_hip is gfx950 (MI355X) only: build with --offload-arch=gfx950"
#endif

extern int g_dz_last_hip_error;

#define DZ_HIP_CHECK(expr)                         \
  do {                                             \
    hipError_t _e = (expr);                        \
    if (_e != hipSuccess) {                        \
      g_dz_last_hip_error = (int)_e;               \
      return DZ_ERR_HIP;                           \
    }                                              \
  } while (0)

#define DZ_LAUNCH_CHECK() DZ_HIP_CHECK(hipGetLastError())

#define DZ_REQUIRE(cond)                  \
  do {                                    \
    if (!(cond)) return DZ_ERR_INVALID_ARG; \
  } while (0)

static inline hipStream_t dz_s(dz_stream_t s) { return (hipStream_t)s; }

static inline bool dz_is_pow2(int64_t v) { return v > 0 && (v & (v - 1)) == 0; }

// Euclidean modulo for possibly negative a (b > 0).
__host__ __device__ static inline int64_t dz_mod(int64_t a, int64_t b) {
  int64_t r = a % b;
  return r < 0 ? r + b : r;
}

// Polling rounds per seam of the one-launch decision kernels before a workgroup gives up
// (dz_act_one.h; dz_act_debug_spin_limit).
extern int g_dz_act_spin_limit;

// ---- optional event profiler (dz_prof_*) ------------------------------------
extern bool g_dz_prof_on;
void dz_prof_begin(hipStream_t s);
void dz_prof_pair(int which, int end, hipStream_t s);
void dz_prof_mark(hipStream_t s, const char* name);
#define DZ_PROF(stream, name)                         \
  do {                                                \
    if (g_dz_prof_on) dz_prof_mark((stream), (name)); \
  } while (0)
