// Device-side pieces of the sum tree shared by dz_sumtree.hip and by learner
// launches that carry the priority write-back as a side job (dz_rainbow.hip).
// Same bit-exactness contract as dz_sumtree.hip: compile with -ffp-contract=off.
#pragma once

#include "dz_common.h"

namespace {

constexpr int kMaxBatch = 1024;

__device__ __forceinline__ bool finite_nonneg(double v) {
  return (v >= 0.0) && (v < __builtin_inf());  // false for NaN, -x, +inf
}

__device__ __forceinline__ void raise(uint32_t* status, uint32_t bit) {
  if (status) atomicOr(status, bit);
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
};
__device__ __forceinline__ void prio_update_body(const PrioUpdateParams& q, int64_t* s_leaf,
                                                 double* s_red) {
  double* node = q.node;
  const int64_t cap = q.cap, N = q.N, size = q.size, t = q.t;
  const int64_t* __restrict__ ids = q.ids;
  const void* __restrict__ prio = q.prio;
  const int is_f32 = q.is_f32, n = q.n;
  const double exponent = q.exponent;
  double* max_seen = q.max_seen;
  uint32_t* status = q.status;
  const int i = threadIdx.x;
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
  set_leaves_and_ancestors(node, cap, leaf, v, active, s_leaf, n);
}

// The same write-back as a side job of a learner launch (dz_mfma_gemm_side): one
// extra 256-thread block, so the 10 us single-workgroup kernel disappears inside a
// contraction that does not depend on the tree.
struct PrioUpdateSide {
  typedef PrioUpdateParams Params;
  __device__ static void run(const Params& q, unsigned block) {
    __shared__ int64_t s_leaf[256];
    __shared__ double s_red[4];
    if (block == 0) prio_update_body(q, s_leaf, s_red);
  }
};

__global__ __launch_bounds__(256) void prio_update_side_kernel(PrioUpdateParams q) {
  PrioUpdateSide::run(q, 0);
}

}  // namespace
