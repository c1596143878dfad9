// In-launch seams between workgroup roles: THE DATA IS ITS OWN FLAG.
//
// An intermediate that one role of a launch produces and another role of the SAME launch consumes
// lives in a buffer that is all-zero bits before its producers write it.  Producers store with
// write-through, agent-coherent stores (`sc1`: the value is at the coherence point -- memory-side
// cache / HBM -- not in the producing XCD's L2) and store -0.0f for a zero; a consumer re-reads the
// words it needs with `sc1` loads until none of them is +0.0f.  No arrival counter, no fence, no
// drained `vmcnt`, no second round trip for the payload: 2.2-2.5 us from the producer's store to
// the consumer's registers on MI355X (the counter + flag + load form measured
// 23.0 against 21.5 us in the decision kernel).  -0.0f behaves as 0 in every sum and product
// downstream, and `x > 0` is false for it like for +0.0f.
//
// Every spin is bounded (dz_act_debug_spin_limit rounds): the callers give up, set a sticky word
// and poison their results instead of hanging the device.
#pragma once

#include "dz_common.h"

namespace {

#define DZ_ACT_RLX __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT
__device__ __forceinline__ void act_store(float* p, float v) { __hip_atomic_store(p, v, DZ_ACT_RLX); }
__device__ __forceinline__ float act_load(const float* p) {
  return __hip_atomic_load(p, DZ_ACT_RLX);
}
__device__ __forceinline__ float2 act_load2(const float* p) {   // 8-byte aligned
  const unsigned long long v = __hip_atomic_load((const unsigned long long*)p, DZ_ACT_RLX);
  return make_float2(__builtin_bit_cast(float, (unsigned)(v & 0xffffffffull)),
                     __builtin_bit_cast(float, (unsigned)(v >> 32)));
}
__device__ __forceinline__ void act_store2(float* p, float a, float b) {   // 8-byte aligned
  const unsigned long long v = (unsigned long long)__builtin_bit_cast(unsigned, a) |
                               ((unsigned long long)__builtin_bit_cast(unsigned, b) << 32);
  __hip_atomic_store((unsigned long long*)p, v, DZ_ACT_RLX);
}
// 16-byte seam accesses (MI355X_MICROARCH.md price list: a dword `sc1` store costs ~6x, an 8-byte
// one 2.7x the time per byte of a 16-byte one; 8-byte loads run at 0.54-0.70x the 16-byte rate):
// raw buffer loads / stores with aux = sc1 through a descriptor of the seam buffer.  `off` in BYTES,
// 16-byte aligned.  (A relaxed agent-scope __hip_atomic lowers to sc1 only up to 8 bytes.)
typedef unsigned dz_u4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t act_rsrc(const void* base) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, 0x7ffffff0, 0x00020000);
}
__device__ __forceinline__ float4 act_load4(__amdgpu_buffer_rsrc_t r, unsigned off) {
  // (the whole vector is bit-cast at once: with a per-element __builtin_bit_cast(float, v.x) this
  // compiler -- ROCm 7.2 -- narrows the load to ONE dword and the other three lanes are garbage)
  typedef float dz_f4v __attribute__((ext_vector_type(4)));
  const dz_f4v v = __builtin_bit_cast(dz_f4v, __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, /*sc1*/ 16));
  return make_float4(v[0], v[1], v[2], v[3]);
}
__device__ __forceinline__ void act_store4(__amdgpu_buffer_rsrc_t r, unsigned off, float4 v) {
  dz_u4 u;
  u.x = __builtin_bit_cast(unsigned, v.x); u.y = __builtin_bit_cast(unsigned, v.y);
  u.z = __builtin_bit_cast(unsigned, v.z); u.w = __builtin_bit_cast(unsigned, v.w);
  __builtin_amdgcn_raw_buffer_store_b128(u, r, off, 0, /*sc1*/ 16);
}
// a zero is stored as -0.0f: all-zero bits mean "not written yet"
__device__ __forceinline__ float act_mark(float v) { return v == 0.f ? -0.f : v; }
__device__ __forceinline__ bool act_missing(float v) { return __builtin_bit_cast(unsigned, v) == 0u; }
__device__ __forceinline__ float4 act_mark4(float4 v) {
  return make_float4(act_mark(v.x), act_mark(v.y), act_mark(v.z), act_mark(v.w));
}
// 1 if all four words are present (branch-free)
__device__ __forceinline__ unsigned act_have4(float4 v) {
  return ((__builtin_bit_cast(unsigned, v.x) != 0u) & (__builtin_bit_cast(unsigned, v.y) != 0u) &
          (__builtin_bit_cast(unsigned, v.z) != 0u) & (__builtin_bit_cast(unsigned, v.w) != 0u)) ? 1u : 0u;
}

// Before the polling rounds: ONE thread watches ONE of the words the workgroup needs (224 x 256
// threads re-reading 14 words each starved every other access of the chip: 38 us per decision).
// The rounds that follow see the rest, written within a microsecond of it.
__device__ __forceinline__ void act_watch(const float* word, int limit) {
  if (threadIdx.x == 0)
    for (int i = 0; i < limit && act_missing(act_load(word)); ++i) __builtin_amdgcn_s_sleep(2);
  __syncthreads();
}
// The same with one word PER PRODUCER: every thread may watch one word (nullptr: none) -- e.g. the
// last word each of the 32 producers of a tile stores -- in rounds of ONE load per thread, until
// none is missing.  The payload round that follows then finds everything (a payload round costs
// 16-64 loads per thread: repeating it for stragglers was 2-8 us per seam in the head chain).
// Returns true if the spin limit was hit (the sticky word is set).
__device__ __forceinline__ bool act_again(bool miss, int round, unsigned* fail, bool* give_up, int limit);
// NAP: extra s_sleep units (64 cycles each) between rounds, for consumers whose producers are
// several phases away -- thousands of pollers re-reading a few lines slow every other seam of the
// launch down (MI355X_MICROARCH.md polling-cost).
template <int NAP = 0>
__device__ __forceinline__ bool act_watch_each(const float* word, unsigned* fail, int limit) {
  int round = 0;
  bool miss, give_up;
  do {
    miss = word != nullptr && act_missing(act_load(word));
    if (NAP > 0 && __syncthreads_or(miss ? 1 : 0)) __builtin_amdgcn_s_sleep(NAP);
  } while (act_again(miss, round++, fail, &give_up, limit));
  return give_up;
}
// One polling round ends here: true = some thread still saw a missing value (go round again);
// after `limit` rounds the sticky failure word is set and *give_up becomes true.
__device__ __forceinline__ bool act_again(bool miss, int round, unsigned* fail, bool* give_up,
                                          int limit) {
  const bool again = __syncthreads_or(miss ? 1 : 0) != 0;
  *give_up = again && round >= limit;
  if (*give_up && threadIdx.x == 0) __hip_atomic_store(fail, 1u, DZ_ACT_RLX);
  if (again) __builtin_amdgcn_s_sleep(1);
  return again && !*give_up;
}

}  // namespace
