// Device-side pieces of the sum tree shared by dz_sumtree.hip and by learner
// launches that carry the priority write-back as a side job (dz_rainbow.hip).
// Same bit-exactness contract as dz_sumtree.hip: compile with -ffp-contract=off.
#pragma once

#include "dz_common.h"

#ifndef DZ_WB_STAMP
#define DZ_WB_STAMP(k)   // tools/micro/wb_micro.hip: in-kernel time stamps
#endif

namespace {

constexpr int kMaxBatch = 1024;

__device__ __forceinline__ bool finite_nonneg(double v) {
  return (v >= 0.0) && (v < __builtin_inf());  // false for NaN, -x, +inf
}

// (the word may live in pinned host memory -- the Python layer polls it there for free --
// hence the system scope; nothing on the device ever reads it)
__device__ __forceinline__ void raise(uint32_t* status, uint32_t bit) {
  if (status) __hip_atomic_fetch_or(status, bit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// Shared body of SumTree.set for one workgroup.  `leaf[i]` are tree indices in
// [0, cap), `val[i]` the new leaf values; n <= blockDim.x.
// ref: replay.py:283-290.  After all leaves are assigned (last duplicate wins),
// the sequential per-index root walks of the reference leave every touched node
// equal to fl(left+right) of its final children; recomputing the touched nodes
// level by level gives the identical final array.
__device__ void set_leaves_and_ancestors(double* node, int64_t cap, int64_t my_leaf,
                                         double my_val, bool active,
                                         const int64_t* s_leaf, int n) {
  const int i = threadIdx.x;
  if (active) {
    bool last = true;
    for (int j = i + 1; j < n; ++j) last &= (s_leaf[j] != my_leaf);
    if (last) node[cap + my_leaf] = my_val;
  }
  __syncthreads();
  int64_t p = (cap + my_leaf) >> 1;
  for (int64_t level = cap >> 1; level >= 1; level >>= 1) {
    if (active) node[p] = node[2 * p] + node[2 * p + 1];
    p >>= 1;
    __syncthreads();
  }
}

// The same result with TWO dependent trips to memory instead of one per level (a batch of
// n <= 255 leaves -- partner indices are bytes and 0xFF means "none", so batch element
// 255 must not exist -- one workgroup of 256 threads, cap <= 2^31).
//   1. every thread requests the sibling of each node on its leaf's path -- up to 31
//      independent loads, one round trip -- before anything is stored;
//   2. one scan over the batch finds, per thread, the duplicates of its leaf (the LAST
//      one's value wins) and its PARTNERS: thread j's path runs through the sibling of
//      thread i's path node at exactly one level, l = floor(log2(x_i ^ x_j)) (the
//      highest bit in which the leaves' node indices differ); partner[l][i] = any such j
//      (all of them carry the same node at level l, hence the same value);
//   3. the walk: at level l a thread takes its sibling's value from its partner's LDS
//      slot (fresh) or, without a partner, from the prefetch (that node is on nobody's
//      path: untouched by this batch) -- two LDS round trips per level, no global access;
//   4. all path nodes are stored at the end, back to back.
// Each touched node ends up exactly fl(left + right) of its final children (IEEE addition
// is commutative: operand order is immaterial), every duplicate of a leaf carries the last
// duplicate's value: the tree is bit-identical to set_leaves_and_ancestors (tools/micro/
// wb_micro.hip, tests/test_fused_step_gpu.py).  As a side block of a busy launch the
// level-by-level form's 21 dependent round trips stretch to 15-30 us; this one does not.
// (First version: an O(n) LDS scan for the sibling at EVERY level -- 1.3 us per level of
// 64-bit compares, 33 us per call; the partner table makes the scan a one-off.)
constexpr int kWbMaxLevels = 31;
constexpr int kWbMaxBatch = 255;         // partner[][] holds j as a byte; 0xFF is the "no partner" mark
struct WbScratch {                       // 14.3 KB; may alias a host kernel's idle LDS
  uint32_t x[256];                       // node index of the leaf (cap + leaf < 2^32)
  double v0[256];                        // leaf values (scan), then level values, ping ...
  double v1[256];                        //   ... pong (one barrier per level)
  unsigned char partner[kWbMaxLevels + 1][256];   // [l][i]: thread i's partner at level l (0xFF: none)
};
// The scan of the batch (steps 2 above): this thread's value with duplicates resolved, and its
// partners' batch indices in S.partner[.][i].  S.x / S.v0 are filled here.
__device__ __forceinline__ double wb_scan(uint32_t x32, double my_val, bool active, int n, WbScratch& S) {
  const int i = threadIdx.x;
  S.x[i] = x32; S.v0[i] = my_val;
#pragma unroll
  for (int l = 0; l <= kWbMaxLevels; ++l) S.partner[l][i] = 0xFF;
  __syncthreads();
  DZ_WB_STAMP(1);
  // the one scan, in batches of 8 (all LDS reads of a batch issued before any use)
  double val = my_val;
  if (i < ((n + 63) & ~63)) {            // waves without a batch element have nothing to find
    for (int j0 = 0; j0 < n; j0 += 8) {
      uint32_t xs[8]; double vs[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) { const int j = min(j0 + u, n - 1); xs[u] = S.x[j]; vs[u] = S.v0[j]; }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int j = j0 + u;
        const bool live = active && j < n;
        val = (live && xs[u] == x32 && j > i) ? vs[u] : val;   // ascending j: the last match stays
        const uint32_t d = xs[u] ^ x32;
        const int bb = 31 - __builtin_clz(d | 1u);             // level of the sibling relation (< lg)
        if (live && d != 0) S.partner[bb][i] = (unsigned char)j;
      }
    }
  }
  return val;
}
// The walk (steps 3 and 4): `val` = this thread's leaf value, `sib` its prefetched siblings,
// S.partner its partners.
__device__ __forceinline__ void wb_walk(double* node, int lg, int64_t x0, double val, bool active,
                                        const double (&sib)[kWbMaxLevels], WbScratch& S) {
  constexpr int kMaxLevels = kWbMaxLevels;
  const int i = threadIdx.x;
  double pv[kMaxLevels + 1];
  int64_t x = x0;
  DZ_WB_STAMP(2);
#pragma unroll
  for (int l = 0; l <= kMaxLevels; ++l) {
    pv[l] = val;
    DZ_WB_STAMP(3 + l);
    if (l < lg) {                         // uniform
      double* buf = (l & 1) ? S.v1 : S.v0;
      if (l == 0) __syncthreads();        // every scan read of v0 is done before it is reused
      buf[i] = val;
      __syncthreads();
      const uint32_t pj = S.partner[l < kMaxLevels ? l : 0][i];   // (own writes: ordered by the barrier)
      const double fresh = buf[pj == 0xFFu ? (uint32_t)i : pj];
      const double sv = pj == 0xFFu ? sib[l < kMaxLevels ? l : 0] : fresh;
      val = (x & 1) ? sv + val : val + sv;   // fl(left + right)
      x >>= 1;
    }
  }
  // all stores of the path at the end, back to back
#pragma unroll
  for (int l = 0; l <= kMaxLevels; ++l)
    if (active && l <= lg) node[x0 >> l] = pv[l];
  DZ_WB_STAMP(40);
}
// ... for a batch of at most 64 (one wave holds it: thread index = batch index): the partner's
// value comes over the lane crossbar, no LDS buffer, no barrier.  `pb` = this thread's partner
// bytes (level l in byte l).  Only wave 0 may call it.
__device__ __forceinline__ void wb_walk_wave(double* node, int lg, int64_t x0, double val, bool active,
                                             const double (&sib)[kWbMaxLevels], const unsigned char (&pb)[32]) {
  constexpr int kMaxLevels = kWbMaxLevels;
  double pv[kMaxLevels + 1];
  int64_t x = x0;
#pragma unroll
  for (int l = 0; l <= kMaxLevels; ++l) {
    pv[l] = val;
    if (l < lg) {                         // uniform
      const uint32_t pj = pb[l < kMaxLevels ? l : 0];
      const double fresh = __shfl(val, (int)(pj & 63u), 64);
      const double sv = pj == 0xFFu ? sib[l < kMaxLevels ? l : 0] : fresh;
      val = (x & 1) ? sv + val : val + sv;   // fl(left + right)
      x >>= 1;
    }
  }
#pragma unroll
  for (int l = 0; l <= kMaxLevels; ++l)
    if (active && l <= lg) node[x0 >> l] = pv[l];
}
__device__ __forceinline__ void set_leaves_and_ancestors_fast(
    double* node, int64_t cap, int64_t my_leaf, double my_val, bool active, int n,
    WbScratch& S) {
  constexpr int kMaxLevels = kWbMaxLevels;
  const int lg = 63 - __builtin_clzll((unsigned long long)cap);   // levels above the leaves
  const int64_t x0 = active ? cap + my_leaf : 0;
  DZ_WB_STAMP(0);
  // (unconditional loads from clamped indices: a conditional load is an exec-mask block
  // with its own full wait, i.e. one round trip per level again)
  double sib[kMaxLevels];
#pragma unroll
  for (int l = 0; l < kMaxLevels; ++l) {
    const int64_t xl = x0 >> l;
    sib[l] = node[(active && xl > 1) ? (xl ^ 1) : 1];
  }
  const double val = wb_scan((uint32_t)x0, my_val, active, n, S);
  wb_walk(node, lg, x0, val, active, sib, S);
}

// id -> tree index and back for the fixed-capacity distribution
// (ref: replay.py:457,499,533: the free stack is popped from its END, and an
// evicted index is pushed and popped straight back).
__device__ __forceinline__ int64_t tree_index_of_id(int64_t id, int64_t N) {
  return N - 1 - dz_mod(id, N);
}
// power_zero_safe in the dtype NumPy would use (replay.py:203-208).
__device__ __forceinline__ double leaf_from_priority_f64(double p, double e) {
  if (p == 0.0) return 0.0;
  if (e == 0.5) return sqrt(p);
  if (e == 1.0) return p;
  if (e == 2.0) return p * p;
  if (e == 0.0) return 1.0;
  return pow(p, e);
}
__device__ __forceinline__ double leaf_from_priority_f32(float p, double e) {
  if (p == 0.0f) return 0.0;
  if (e == 0.5) return (double)sqrtf(p);
  if (e == 1.0) return (double)p;
  if (e == 2.0) return (double)(p * p);
  if (e == 0.0) return 1.0;
  return (double)powf(p, (float)e);
}

// dz_prioritized_update for one workgroup (n <= blockDim.x threads active).
struct PrioUpdateParams {
  double* node; int64_t cap; int64_t N; int64_t size; int64_t t;
  const int64_t* ids; const void* prio; int is_f32; double exponent; int n;
  double* max_seen; uint32_t* status;
  int check_ids;   // 0: ids come straight from this replay's sampler (live by construction)
  const unsigned* abort = nullptr;   // non-zero word: the step that produced `prio` is void, write nothing
  // The write-back as TWO blocks in two consecutive launches (one workgroup's chain of three
  // dependent trips to memory is ~9.5 us -- longer than either host launch): phase 1 checks the
  // batch, updates the running maximum, and leaves the leaf values, node indices and the 31
  // siblings of every path in `scratch` (kPrioScratchDoubles doubles, nobody else's); phase 2 reads
  // them back (one trip, L2) and does the walk and the stores.  The tree must not change in between.
  int phase = 0;                     // 0: everything in one block
  double* scratch = nullptr;
};
// flag | values (duplicates resolved) | node indices | siblings [31][256] | partners [256][32 bytes]
constexpr int kPrioScratchDoubles = 1 + 2 * 256 + kWbMaxLevels * 256 + 4 * 256;
constexpr int kPrioScrSib = 1 + 512, kPrioScrPartner = kPrioScrSib + kWbMaxLevels * 256;
__device__ __forceinline__ void prio_update_body(const PrioUpdateParams& q, int64_t* s_leaf,
                                                 double* s_red, WbScratch* wb = nullptr) {
  const int i = threadIdx.x;
  double* node = q.node;
  const int64_t cap = q.cap, N = q.N, size = q.size, t = q.t;
  const int n = q.n;
  if (q.phase == 2) {   // ---- second half: the walk, from what phase 1 left in the scratch ----
    // (phase 1 found the step void or the batch bad: nothing to do; all threads look BEFORE the clear)
    if (!__syncthreads_or(q.scratch[0] == 1.0 ? 1 : 0)) return;
    if (i == 0) q.scratch[0] = 0.0;    // consumed (a phase 2 without its phase 1 does nothing)
    if (i >= 64) return;               // (phase 1 took the split only for batches of <= 64: one wave)
    const bool active = i < n;
    const int lg = 63 - __builtin_clzll((unsigned long long)cap);
    const double val = q.scratch[1 + i];
    const int64_t x0 = __builtin_bit_cast(int64_t, q.scratch[1 + 256 + i]);
    double sib[kWbMaxLevels];
#pragma unroll
    for (int l = 0; l < kWbMaxLevels; ++l) sib[l] = q.scratch[kPrioScrSib + l * 256 + i];
    unsigned char pb[32];
    {
      const uint4* pp = reinterpret_cast<const uint4*>(q.scratch + kPrioScrPartner + 4 * i);
      const uint4 a = pp[0], b = pp[1];
      const unsigned w[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
      for (int l = 0; l < 32; ++l) pb[l] = (unsigned char)(w[l >> 2] >> (8 * (l & 3)));
    }
    wb_walk_wave(node, lg, x0, val, active, sib, pb);
    return;
  }
  if (q.phase == 1 && i == 0) q.scratch[0] = 0.0;
  if (q.abort && *q.abort) return;   // (launch-uniform)
  const int64_t* __restrict__ ids = q.ids;
  const void* __restrict__ prio = q.prio;
  const int is_f32 = q.is_f32;
  const double exponent = q.exponent;
  double* max_seen = q.max_seen;
  uint32_t* status = q.status;
  const bool active = i < n;
  int64_t leaf = 0;
  double v = 0.0, p64 = 0.0;
  bool bad_id = false;
  if (active) {
    const int64_t id = ids[i];
    bad_id = q.check_ids && ((id < t - size) || (id >= t));  // replay.py:541-543
    leaf = tree_index_of_id(id, N);
    if (is_f32) {
      const float p = ((const float*)prio)[i];
      p64 = (double)p;
      v = leaf_from_priority_f32(p, exponent);
    } else {
      p64 = ((const double*)prio)[i];
      v = leaf_from_priority_f64(p64, exponent);
    }
    s_leaf[i] = leaf;
  }
  const bool bad_v = active && !finite_nonneg(v);
  const int any_bad_i = __syncthreads_or(bad_id);
  const int any_bad_v = __syncthreads_or(bad_v);
  if (any_bad_i || any_bad_v) {
    if (i == 0) raise(status, (any_bad_v ? DZ_ST_BAD_VALUE : 0u) |
                                  (any_bad_i ? DZ_ST_BAD_INDEX : 0u));
    return;
  }
  if (q.phase == 1) {   // ---- first half: the siblings' trip starts now, next to the maximum's ----
    const int64_t x0 = active ? cap + leaf : 0;
    double sib[kWbMaxLevels];
#pragma unroll
    for (int l = 0; l < kWbMaxLevels; ++l) {
      const int64_t xl = x0 >> l;
      sib[l] = node[(active && xl > 1) ? (xl ^ 1) : 1];
    }
    // (the scan too: duplicates resolved and partners found here, under this launch)
    const double val = wb_scan((uint32_t)x0, v, active, n, *wb);
    __syncthreads();
    q.scratch[1 + i] = val;
    q.scratch[1 + 256 + i] = __builtin_bit_cast(double, x0);
#pragma unroll
    for (int l = 0; l < kWbMaxLevels; ++l) q.scratch[kPrioScrSib + l * 256 + i] = sib[l];
    {
      unsigned w[8];
#pragma unroll
      for (int k = 0; k < 8; ++k)
        w[k] = (unsigned)wb->partner[4 * k][i] | ((unsigned)wb->partner[4 * k + 1][i] << 8) |
               ((unsigned)wb->partner[4 * k + 2][i] << 16) | ((unsigned)wb->partner[4 * k + 3][i] << 24);
      uint4* pp = reinterpret_cast<uint4*>(q.scratch + kPrioScrPartner + 4 * i);
      pp[0] = make_uint4(w[0], w[1], w[2], w[3]); pp[1] = make_uint4(w[4], w[5], w[6], w[7]);
    }
  }
  if (max_seen) {  // rainbow/agent.py:196-197
    double m = active ? p64 : -__builtin_inf();
    for (int off = 32; off >= 1; off >>= 1) {
      const double o = __shfl_xor(m, off);
      m = o > m ? o : m;
    }
    if ((i & 63) == 0) s_red[i >> 6] = m;
    __syncthreads();
    if (i == 0) {
      double mm = *max_seen;
      for (int k = 0; k < (int)((blockDim.x + 63) / 64); ++k)
        mm = s_red[k] > mm ? s_red[k] : mm;
      *max_seen = mm;
    }
  }
  if (q.phase == 1) {
    if (i == 0) q.scratch[0] = 1.0;   // (same block, same launch as the stores above: the NEXT launch reads it)
    return;
  }
  if (wb && n <= kWbMaxBatch && blockDim.x == 256 && cap <= ((int64_t)1 << 31)) {
    __syncthreads();
    set_leaves_and_ancestors_fast(node, cap, leaf, v, active, n, *wb);
  } else {
    set_leaves_and_ancestors(node, cap, leaf, v, active, s_leaf, n);
  }
}

// The same write-back as a side job of a learner launch (dz_mfma_gemm_side): one
// extra 256-thread block, so the 10 us single-workgroup kernel disappears inside a
// contraction that does not depend on the tree.
// FAST: the two-round-trip walk (set_leaves_and_ancestors_fast) -- for hosts whose
// register budget is a GEMM's anyway: its prefetched siblings and path values cost
// ~130 VGPRs, which would take adam_kernel from 8 to 3 waves per SIMD (Adam's block 0
// keeps the level-by-level form).  `scratch`: the host kernel's LDS, idle in this block.
template <int FAST>
struct PrioUpdateSideT {
  typedef PrioUpdateParams Params;
  __device__ static void run(const Params& q, unsigned block, void* scratch = nullptr,
                             int scratch_bytes = 0) {
    __shared__ int64_t s_leaf[256];
    __shared__ double s_red[4];
    if (block != 0) return;
    if constexpr (FAST) {   // (no scratch offered: the level-by-level form)
      WbScratch* wb = (scratch && scratch_bytes >= (int)sizeof(WbScratch))
                          ? reinterpret_cast<WbScratch*>(scratch) : nullptr;
      prio_update_body(q, s_leaf, s_red, wb);
    } else {
      prio_update_body(q, s_leaf, s_red, nullptr);
    }
  }
};
typedef PrioUpdateSideT<0> PrioUpdateSide;
typedef PrioUpdateSideT<1> PrioUpdateSideFast;

__global__ __launch_bounds__(256) void prio_update_side_kernel(PrioUpdateParams q) {
  PrioUpdateSide::run(q, 0);
}


// --------------------------------------------------------------------------- //
//  Sampling (moved here from dz_sumtree.hip so that a learner launch can carry the
//  NEXT step's sample + gather as side blocks).
// --------------------------------------------------------------------------- //
// Wave-wide NaN-propagating max of doubles without ds_bpermute round trips: quad permutes,
// row_half_mirror and row_mirror on both 32-bit halves give every lane of a 16-lane row
// the row's result, the four rows are combined through v_readlane.  max is exact, so
// the order of the combination does not matter.  All 64 lanes must be active.
template <int CTRL>
__device__ __forceinline__ double dz_dpp_f64(double v) {
  const uint64_t u = __builtin_bit_cast(uint64_t, v);
  const uint32_t lo = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)u, CTRL, 0xf, 0xf, true);
  const uint32_t hi = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)(u >> 32), CTRL, 0xf, 0xf, true);
  return __builtin_bit_cast(double, ((uint64_t)hi << 32) | lo);
}
__device__ __forceinline__ double dz_lane_f64(double v, int l) {
  const uint64_t u = __builtin_bit_cast(uint64_t, v);
  const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)u, l);
  const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(u >> 32), l);
  return __builtin_bit_cast(double, ((uint64_t)hi << 32) | lo);
}
__device__ __forceinline__ double dz_nanmax(double a, double b) {
  return (a != a || b != b) ? __builtin_nan("") : (b > a ? b : a);
}
__device__ __forceinline__ double dz_wave_nanmax_f64(double m) {
  m = dz_nanmax(m, dz_dpp_f64<0xB1>(m));
  m = dz_nanmax(m, dz_dpp_f64<0x4E>(m));
  m = dz_nanmax(m, dz_dpp_f64<0x141>(m));
  m = dz_nanmax(m, dz_dpp_f64<0x140>(m));
  return dz_nanmax(dz_nanmax(dz_lane_f64(m, 0), dz_lane_f64(m, 16)),
                   dz_nanmax(dz_lane_f64(m, 32), dz_lane_f64(m, 48)));
}

// ref: replay.py:406-426.
__device__ __forceinline__ int64_t descend(const double* __restrict__ node,
                                           int64_t cap, double target) {
  int64_t i = 1;
  while (i < cap) {
    const double left = node[2 * i];
    if (target < left) {
      i = 2 * i;
    } else {
      target -= left;
      i = 2 * i + 1;
    }
  }
  return i - cap;
}

// The same descent by LANES (8 or 16) consecutive lanes (sub = lane % LANES, all with the
// same target), L = log2(LANES) levels per memory round trip: lane sub >= 1 loads the
// left-child sum of one of the LANES-1 nodes that can be the current node within the next
// L steps
//   sub 1: 2i     sub 2, 3: 4i, 4i+2     sub 4..7: 8i, 8i+2, 8i+4, 8i+6     sub 8..15: 16i + 2(sub-8)
// and the group then takes the L decisions from registers (shuffles).  Same comparisons
// and subtractions on the same node values as descend(): the result is identical; the
// chain is ceil(levels/L) dependent loads instead of `levels` (20 levels: 7 round trips
// with 8 lanes, 5 with 16).
template <int LANES>
__device__ __forceinline__ int64_t descend_coop(const double* __restrict__ node, int64_t cap,
                                                double target, int sub) {
  constexpr int L = LANES == 16 ? 4 : 3;
  static_assert(LANES == 8 || LANES == 16, "group size");
  const int base = (int)(threadIdx.x & 63) & ~(LANES - 1);  // first lane of this group
  int64_t i = 1;
  while (i < cap) {
    // sub in [2^j, 2^(j+1)): node 2^(j+1) i + 2 (sub - 2^j)
    const int j = sub < 2 ? 0 : (sub < 4 ? 1 : (sub < 8 ? 2 : 3));
    int64_t idx = (i << (j + 1)) + 2 * (sub - (1 << j));
    idx = (sub >= 1 && idx < 2 * cap) ? idx : 2 * cap - 1;  // past the leaves / lane 0: unused
    const double v = node[idx];
    int pick = 1;
#pragma unroll
    for (int step = 0; step < L; ++step) {
      const double left = __shfl(v, base + pick);
      if (i < cap) {  // group-uniform
        int d = 0;
        if (target < left) {
          i = 2 * i;
        } else {
          target -= left;
          i = 2 * i + 1;
          d = 1;
        }
        pick = 2 * pick + d;  // 1 -> 2|3 -> 4..7 -> 8..15
      }
    }
  }
  return i - cap;
}

// The live id whose slot is N-1-ti; live ids are [t-size, t).
__device__ __forceinline__ int64_t id_of_tree_index(int64_t ti, int64_t N,
                                                    int64_t t, int64_t size) {
  const int64_t slot = N - 1 - ti;
  const int64_t base = t - size;
  return base + dz_mod(slot - base, N);
}
// ref: replay.py:52-82 applied to _active_indices (positions hold tree indices
// of the ids of the uniform swap-remove list).
__device__ __forceinline__ int64_t id_at_position(int64_t j, int64_t N, int64_t t) {
  if (t <= N || N == 1) return (N == 1) ? t - 1 : j;
  if (j == N - 1) return t - 1;
  const int64_t base = t - N;
  return base + dz_mod(j - base, N - 1);
}

// RNG draws of a small batch passed BY VALUE in the kernel arguments (1.5 KB of
// the 4 KB kernarg segment): no staging buffer, no H2D copy, no blit kernel in
// front of the sample (that copy was a 4 us launch of its own per step).
constexpr int kMaxHostDraws = 64;
struct HostDraws {
  int64_t pos[kMaxHostDraws];
  double u_target[kMaxHostDraws];
  double u_mix[kMaxHostDraws];
};

// Tree index drawn for batch element i (replay.py:551-567): uniform candidate,
// prioritized candidate (descent), mix.  `bad` reports a target outside [0, root).
// COOP = 8 or 16: called by COOP consecutive lanes with the same i; `sub` = lane % COOP
// (descend_coop).
template <int HOST_DRAWS, int COOP = 0>
__device__ __forceinline__ int64_t sample_tree_index(const dz_prio_sample_args_t& a,
                                                     const HostDraws& hd, int i, double root,
                                                     bool zero_root, bool& bad, int sub = 0) {
  const int64_t N = a.capacity;
  const int64_t pos_i = HOST_DRAWS ? hd.pos[i & (kMaxHostDraws - 1)] : a.pos[i];
  const double ut_i = HOST_DRAWS ? hd.u_target[i & (kMaxHostDraws - 1)] : a.u_target[i];
  const double um_i = HOST_DRAWS ? hd.u_mix[i & (kMaxHostDraws - 1)] : a.u_mix[i];
  const int64_t uni_ti = tree_index_of_id(id_at_position(pos_i, N, a.t), N);
  int64_t pri_ti = uni_ti;
  bad = false;
  if (!zero_root) {
    const double target = ut_i * root;
    if (!(0.0 <= target && target < root)) bad = true;
    else if constexpr (COOP != 0) pri_ti = descend_coop<COOP>(a.node, a.cap_pow2, target, sub);
    else pri_ti = descend(a.node, a.cap_pow2, target);
  }
  return (um_i < a.usp) ? uni_ti : pri_ti;
}

// s_ti != null (n <= 64): the descents are done first, COOP lanes per batch element
// (descend_coop), and parked in s_ti[]; otherwise one thread per element.
template <int HOST_DRAWS, int COOP = 8>
__device__ __forceinline__ void prioritized_sample_body(
    const dz_prio_sample_args_t& a, const HostDraws& hd, int n, int64_t* __restrict__ ids_out,
    int64_t* __restrict__ tree_idx_out, double* __restrict__ probs_out,
    double* __restrict__ weights_out, float* __restrict__ weights32_out,
    uint32_t* status, double* s_red, double& s_max, int64_t* s_ti = nullptr) {
  const int i = threadIdx.x;
  const bool active = i < n;
  const double* __restrict__ node = a.node;
  const int64_t N = a.capacity, cap = a.cap_pow2;
  const double root = node[1];
  const bool zero_root = (root == 0.0);
  if (zero_root && a.assume_nonzero_root && i == 0) raise(status, DZ_ST_ZERO_ROOT);
  if (s_ti) {
    for (int q = i / COOP; q < n; q += (int)blockDim.x / COOP) {
      bool bad;
      const int64_t ti =
          sample_tree_index<HOST_DRAWS, COOP>(a, hd, q, root, zero_root, bad, i % COOP);
      if ((i % COOP) == 0) {
        if (bad) raise(status, DZ_ST_BAD_TARGET);
        s_ti[q] = ti;
      }
    }
    __syncthreads();
  }

  double w = 0.0;
  if (active) {
    bool bad = false;
    const int64_t ti = s_ti ? s_ti[i]
                            : sample_tree_index<HOST_DRAWS>(a, hd, i, root, zero_root, bad);
    if (bad) raise(status, DZ_ST_BAD_TARGET);
    // probabilities: replay.py:569-577 (separate mul, mul, add: no FMA)
    const double leaf = node[cap + ti];
    const double pp = zero_root ? a.uniform_prob : leaf / root;
    const double m1 = a.one_minus_usp * pp;
    const double prob = m1 + a.usp_times_up;
    if (ids_out) ids_out[i] = id_of_tree_index(ti, N, a.t, a.size);
    if (tree_idx_out) tree_idx_out[i] = ti;
    if (probs_out) probs_out[i] = prob;
    if (a.compute_weights) {
      // replay.py:238: (uniform_probability / probabilities) ** exponent
      const double ratio = a.uniform_prob / prob;
      if (a.beta == 1.0) w = ratio;            // NumPy scalar fast path
      else if (a.beta == 0.5) w = sqrt(ratio); //   "
      else w = pow(ratio, a.beta);
    }
  }
  if (!a.compute_weights) return;

  if (a.normalize) {  // replay.py:239-240: weights /= max(weights)
    double m = active ? w : -__builtin_inf();
    // NaN-propagating max like np.max (order-independent: exact), on the DPP crossbar
    m = dz_wave_nanmax_f64(m);
    if ((i & 63) == 0) s_red[i >> 6] = m;
    __syncthreads();
    if (i == 0) {
      double mm = s_red[0];
      for (int k = 1; k < (int)((blockDim.x + 63) / 64); ++k) {
        const double o = s_red[k];
        mm = (mm != mm || o != o) ? __builtin_nan("") : (o > mm ? o : mm);
      }
      s_max = mm;
    }
    __syncthreads();
    w = w / s_max;
  }
  if (active) {
    if (!(w - w == 0.0)) raise(status, DZ_ST_NONFINITE_WEIGHT);  // replay.py:241
    if (weights_out) weights_out[i] = w;
    if (weights32_out) weights32_out[i] = (float)w;  // the jit-boundary cast
  }
}

// Sample AND gather as the blocks of ONE launch (batch <= 64, draws in the kernel
// arguments).  Block 0 is the sampler proper (ids, probabilities, IS weights: exactly
// prioritized_sample_body); every other block copies one chunk of one field of one
// batch element and re-derives ITS element's tree index with the same arithmetic (16
// lanes, 5 dependent round trips for 20 levels: descend_coop<16>) instead of waiting
// for a second launch to read ids[]: the descent and the copy overlap.  The block
// list is compact: field f owns blocks [first[f], first[f+1]) = n elements x chunks[f]
// (a chunk = THREADS 16-byte vectors, or THREADS bytes on the byte path -- a row that
// is not a multiple of 16 bytes or not 16-byte aligned; at most 64 chunks per row,
// longer rows loop).
struct SampleGatherParams {
  dz_prio_sample_args_t a;
  HostDraws hd;
  dz_field_t f[DZ_MAX_FIELDS];
  int first[DZ_MAX_FIELDS + 1];
  int chunks[DZ_MAX_FIELDS];
  int num_fields, n;
  int64_t* ids_out; double* probs_out; double* weights_out; float* weights32_out;
  uint32_t* status;
};
__host__ __device__ static inline bool dz_field_vec_ok(const dz_field_t& f) {
  return ((f.row_bytes & 15) == 0) && ((((uintptr_t)f.src) & 15) == 0) &&
         ((((uintptr_t)f.dst) & 15) == 0);
}
// Fills q.f / first / chunks for `threads`-wide blocks; returns the block count.
static inline unsigned sample_gather_plan(SampleGatherParams& q, const dz_field_t* fields,
                                          int num_fields, int n, int threads,
                                          int max_chunks = 64) {
  q.num_fields = num_fields; q.n = n;
  int next = 1;
  for (int i = 0; i < DZ_MAX_FIELDS; ++i) {
    if (i < num_fields) {
      q.f[i] = fields[i];
      const int64_t units = dz_field_vec_ok(fields[i]) ? fields[i].row_bytes >> 4
                                                        : fields[i].row_bytes;
      int64_t c = (units + threads - 1) / threads;
      c = c < 1 ? 1 : (c > max_chunks ? max_chunks : c);
      q.first[i] = next; q.chunks[i] = (int)c;
      next += n * (int)c;
    } else {
      q.f[i] = fields[0]; q.first[i] = next; q.chunks[i] = 1;
    }
  }
  q.first[DZ_MAX_FIELDS] = next;
  return (unsigned)next;
}
template <int THREADS>
__device__ __forceinline__ void sample_gather_block(const SampleGatherParams& q, unsigned blk) {
  __shared__ double s_red[THREADS / 64];
  __shared__ double s_max;
  __shared__ int64_t s_slot;
  __shared__ int64_t s_ti[kMaxHostDraws];
  // q.a.node == null: UNIFORM replay (replay.py:44-117, 141-175): element b is the id at
  // position hd.pos[b] of the swap-remove list (closed form), no tree, no weights
  const bool uniform = q.a.node == nullptr;
  if (blk == 0) {
    if (uniform) {
      if ((int)threadIdx.x < q.n && q.ids_out)
        q.ids_out[threadIdx.x] = id_at_position(q.hd.pos[threadIdx.x & (kMaxHostDraws - 1)],
                                                q.a.capacity, q.a.t);
      return;
    }
    prioritized_sample_body<1, 16>(q.a, q.hd, q.n, q.ids_out, nullptr, q.probs_out,
                                   q.weights_out, q.weights32_out, q.status, s_red, s_max, s_ti);
    return;
  }
  int fi = 0;
#pragma unroll
  for (int g = 1; g < DZ_MAX_FIELDS; ++g) fi = (g < q.num_fields && (int)blk >= q.first[g]) ? g : fi;
  if ((int)blk >= q.first[DZ_MAX_FIELDS]) return;
  const int rel = (int)blk - q.first[fi], chunks = q.chunks[fi];
  const int b = rel / chunks, c = rel - b * chunks;
  if (uniform) {
    if (threadIdx.x == 0)
      s_slot = dz_mod(id_at_position(q.hd.pos[b & (kMaxHostDraws - 1)], q.a.capacity, q.a.t),
                      q.a.capacity);
  } else if (threadIdx.x < 16) {  // 16 lanes walk the element's descent, 4 levels per round trip
    const double root = q.a.node[1];
    bool bad;
    const int64_t ti = sample_tree_index<1, 16>(q.a, q.hd, b, root, root == 0.0, bad, threadIdx.x);
    if (threadIdx.x == 0)
      s_slot = dz_mod(id_of_tree_index(ti, q.a.capacity, q.a.t, q.a.size), q.a.capacity);
  }
  __syncthreads();
  const dz_field_t fd = q.f[fi];
  const int64_t rb = fd.row_bytes;
  const char* src = (const char*)fd.src + s_slot * rb;
  char* dst = (char*)fd.dst + (int64_t)b * rb;
  if (dz_field_vec_ok(fd)) {
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    const int64_t nvec = rb >> 4, step = (int64_t)chunks * THREADS;
    // four 16-byte units per lane in flight (clamped, unconditional loads; stores behind them):
    // a block of the side-job form (few fat chunks per row) copies its 16 KB in one round trip
    for (int64_t i0 = (int64_t)c * THREADS + threadIdx.x; i0 < nvec; i0 += 4 * step) {
      u32x4 v[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int64_t i = i0 + k * step;
        v[k] = __builtin_nontemporal_load((const u32x4*)src + (i < nvec ? i : nvec - 1));
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int64_t i = i0 + k * step;
        if (i < nvec) ((u32x4*)dst)[i] = v[k];
      }
    }
  } else {
    for (int64_t i = (int64_t)c * THREADS + threadIdx.x; i < rb; i += (int64_t)chunks * THREADS)
      dst[i] = src[i];
  }
}
// dz_next_sample_t (a learner step's `next_sample`) -> the 256-thread block list.
// args.node == NULL selects the uniform replay (positions only).
static inline int sample_gather_from_desc(const dz_next_sample_t* ns, SampleGatherParams& q,
                                          unsigned* blocks) {
  DZ_REQUIRE(ns && ns->ids_out && ns->n > 0 && ns->n <= kMaxHostDraws && ns->pos_h &&
             ns->fields && ns->num_fields > 0 && ns->num_fields <= DZ_MAX_FIELDS);
  const bool uniform = ns->args.node == nullptr;
  DZ_REQUIRE(uniform || (ns->u_target_h && ns->u_mix_h && dz_is_pow2(ns->args.cap_pow2) &&
                         ns->args.capacity <= ns->args.cap_pow2));
  DZ_REQUIRE(ns->args.capacity > 0 && ns->args.size > 0 && ns->args.size <= ns->args.capacity &&
             ns->args.t >= ns->args.size);
  q.a = ns->args;
  for (int i = 0; i < kMaxHostDraws; ++i) {
    const int j = i < ns->n ? i : 0;
    q.hd.pos[i] = ns->pos_h[j];
    DZ_REQUIRE(q.hd.pos[i] >= 0 && q.hd.pos[i] < ns->args.size);
    q.hd.u_target[i] = uniform ? 0.0 : ns->u_target_h[j];
    q.hd.u_mix[i] = uniform ? 0.0 : ns->u_mix_h[j];
  }
  for (int i = 0; i < ns->num_fields; ++i)
    DZ_REQUIRE(ns->fields[i].src && ns->fields[i].dst && ns->fields[i].row_bytes > 0);
  // side-job form: at most 4 chunks per row, four 16-byte units per lane (545 -> 289 blocks for a
  // Rainbow batch).  Measured same-box: 64 / 4 / 2 / 1 chunks per row = 10.90 / 10.96 / 10.97 /
  // 10.97 k steps/s on BASELINE config 2, 9.50-9.53 k on config 3, Rainbow 6.63 / 6.64 / 6.62 /
  // 6.58 k: the block count is not what the side job costs its host (the descent's five dependent
  // round trips are).
#ifndef DZ_SG_SIDE_CHUNKS
#define DZ_SG_SIDE_CHUNKS 4
#endif
  *blocks = sample_gather_plan(q, ns->fields, ns->num_fields, ns->n, 256, DZ_SG_SIDE_CHUNKS);
  q.ids_out = ns->ids_out; q.probs_out = ns->probs_out; q.weights_out = ns->weights_out;
  q.weights32_out = ns->weights32_out; q.status = ns->status;
  return DZ_OK;
}

// The same blocks as extra blocks of a learner launch (256-thread workgroups).
struct SampleGatherSide {
  typedef SampleGatherParams Params;
  __device__ static void run(const Params& q, unsigned block) { sample_gather_block<256>(q, block); }
};

}  // namespace
