// `global_load_lds_dwordx4` as one statement (shared by every LDS-DMA kernel of the library).
#pragma once

#include "dz_common.h"

namespace {

// One LDS-DMA instruction: 64 lanes x 16 bytes from each lane's `src` to LDS bytes
// [lds_byte, lds_byte + 1024) in lane order.  Inline assembly on purpose: hipcc ties every later
// LDS read to an outstanding __builtin_amdgcn_global_load_lds with `s_waitcnt vmcnt(0)`, which
// drains the whole stream in front of the first MFMA; an asm load is absent from its
// bookkeeping, so the counted waits below (dm_chunk) are the only ones -- and therefore no
// ordinary global load may be in flight between the first DMA and the last wait.  M0 (the
// DMA's LDS base) is compiler-reserved: saved, set and restored inside the one statement
// (cdna_hip_programming.md 5.6, "LDS-DMA recipe").
// (NT = 1, non-temporal: measured 11.0 vs 10.75 us for this launch and a slower step -- the
// optimiser re-reads these weights 30 us later and finds them in the Infinity Cache.)
template <int NT>
__device__ __forceinline__ void dz_glds16(const float* src, unsigned lds_byte) {
  unsigned keep;
  if constexpr (NT)
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(src), "s"(lds_byte) : "memory");
  else
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(src), "s"(lds_byte) : "memory");
}

}  // namespace
