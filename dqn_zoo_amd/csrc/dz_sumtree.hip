// Sum-tree kernels (float64 implicit heap in HBM) and the fused prioritized
// sample / update / add built on them.
//
// Bit-exactness contract (tests/test_replay_gpu.py): every internal node is
// exactly fl(left + right) in IEEE float64, the descent compares and subtracts
// in float64, and products/sums that NumPy evaluates as separate operations
// are kept separate here -- this file MUST be compiled with -ffp-contract=off.
//
// Roofline: dependent-load latency (log2(cap) levels), not bandwidth: a batch of
// 32 touches 32*20*2*8 B = 10 KiB when sampling and 15 KiB when updating
// (SURVEY.md 8d).  All batch-sized kernels run as ONE workgroup so that the
// level-by-level recompute can use workgroup barriers.
#include "dz_sumtree_dev.h"

namespace {

__global__ __launch_bounds__(kMaxBatch) void sumtree_set_kernel(
    double* node, int64_t cap, int64_t size, const int64_t* __restrict__ idx,
    const double* __restrict__ val, int n, uint32_t* status) {
  __shared__ int64_t s_leaf[kMaxBatch];
  const int i = threadIdx.x;
  const bool active = i < n;
  int64_t leaf = active ? idx[i] : 0;
  const double v = active ? val[i] : 0.0;
  const bool bad_v = active && !finite_nonneg(v);
  if (leaf < 0) leaf += size;  // NumPy negative indexing on the values view.
  const bool bad_i = active && (leaf < 0 || leaf >= size);
  if (active) s_leaf[i] = leaf;
  const int any_bad_v = __syncthreads_or(bad_v);
  const int any_bad_i = __syncthreads_or(bad_i);
  if (any_bad_v || any_bad_i) {  // the reference raises before writing anything
    if (i == 0) raise(status, (any_bad_v ? DZ_ST_BAD_VALUE : 0u) |
                                  (any_bad_i ? DZ_ST_BAD_INDEX : 0u));
    return;
  }
  set_leaves_and_ancestors(node, cap, leaf, v, active, s_leaf, n);
}

__global__ void sumtree_get_kernel(const double* __restrict__ node, int64_t cap,
                                   int64_t size, const int64_t* __restrict__ idx,
                                   int n, double* __restrict__ out,
                                   uint32_t* status) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int64_t k = idx[i];
  if (k < 0 || k >= size) {
    raise(status, DZ_ST_BAD_INDEX);
    out[i] = __builtin_nan("");
    return;
  }
  out[i] = node[cap + k];
}

// Zero the tail leaves [size, cap) and node[0].
__global__ void sumtree_zero_tail_kernel(double* node, int64_t cap, int64_t size) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i == 0) node[0] = 0.0;
  const int64_t k = size + i;
  if (k < cap) node[cap + k] = 0.0;
}

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

// One launch per level: node[i] = node[2i] + node[2i+1] for i in [first, 2*first).
__global__ void sumtree_level_kernel(double* node, int64_t first) {
  const int64_t i = first + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < 2 * first) node[i] = node[2 * i] + node[2 * i + 1];
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

__global__ void sumtree_query_kernel(const double* __restrict__ node, int64_t cap,
                                     const double* __restrict__ targets, int n,
                                     int64_t* __restrict__ out, uint32_t* status) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double t = targets[i];
  const double root = node[1];
  if (!(0.0 <= t && t < root)) {
    raise(status, DZ_ST_BAD_TARGET);
    out[i] = -1;
    return;
  }
  out[i] = descend(node, cap, t);
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

template <int HOST_DRAWS>
__global__ __launch_bounds__(kMaxBatch) void prioritized_sample_kernel(
    dz_prio_sample_args_t a, HostDraws hd, int n, int64_t* __restrict__ ids_out,
    int64_t* __restrict__ tree_idx_out, double* __restrict__ probs_out,
    double* __restrict__ weights_out, float* __restrict__ weights32_out,
    uint32_t* status) {
  __shared__ double s_red[kMaxBatch / 64];
  __shared__ double s_max;
  prioritized_sample_body<HOST_DRAWS>(a, hd, n, ids_out, tree_idx_out, probs_out, weights_out,
                                      weights32_out, status, s_red, s_max);
}

// Sample AND gather in one launch (batch <= 64, draws in the kernel arguments).
// Block row y == n is the sampler proper (ids, probabilities, IS weights: exactly
// prioritized_sample_body); every gather block (x = chunk, y = batch element,
// z = field) re-derives ITS element's tree index with the same arithmetic (16
// lanes, 5 dependent round trips for 20 levels: descend_coop<16>) instead of
// waiting for a second launch to read ids[]: the descent and the copy overlap.
// 512 threads: the sampler block walks 32 elements x 16 lanes in one pass.
struct SampleGatherFields { dz_field_t f[DZ_MAX_FIELDS]; int num_fields; };
constexpr int kSampleGatherThreads = 512;
__global__ __launch_bounds__(kSampleGatherThreads) void prioritized_sample_gather_kernel(
    dz_prio_sample_args_t a, HostDraws hd, int n, SampleGatherFields gf,
    int64_t* __restrict__ ids_out, double* __restrict__ probs_out,
    double* __restrict__ weights_out, float* __restrict__ weights32_out, uint32_t* status) {
  __shared__ double s_red[kSampleGatherThreads / 64];
  __shared__ double s_max;
  __shared__ int64_t s_slot;
  __shared__ int64_t s_ti[kMaxHostDraws];
  if ((int)blockIdx.y == n) {
    if (blockIdx.x == 0 && blockIdx.z == 0)
      prioritized_sample_body<1, 16>(a, hd, n, ids_out, nullptr, probs_out, weights_out,
                                 weights32_out, status, s_red, s_max, s_ti);
    return;
  }
  const int b = blockIdx.y;
  // chunk blocks beyond this field's row have nothing to copy (the grid is sized for the
  // widest field; the scalar fields need one block): leave before walking the tree
  if ((int64_t)blockIdx.x * kSampleGatherThreads * 16 >= gf.f[blockIdx.z].row_bytes && blockIdx.x > 0)
    return;
  if (threadIdx.x < 16) {  // 16 lanes walk the element's descent, 4 levels per round trip
    const double root = a.node[1];
    bool bad;
    const int64_t ti = sample_tree_index<1, 16>(a, hd, b, root, root == 0.0, bad, threadIdx.x);
    if (threadIdx.x == 0)
      s_slot = dz_mod(id_of_tree_index(ti, a.capacity, a.t, a.size), a.capacity);
  }
  __syncthreads();
  const dz_field_t fd = gf.f[blockIdx.z];
  const int64_t rb = fd.row_bytes;
  const char* src = (const char*)fd.src + s_slot * rb;
  char* dst = (char*)fd.dst + (int64_t)b * rb;
  const bool vec_ok = ((rb & 15) == 0) && ((((uintptr_t)fd.src) & 15) == 0) &&
                      ((((uintptr_t)fd.dst) & 15) == 0);
  if (vec_ok) {
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    const int64_t nvec = rb >> 4;
    for (int64_t i = (int64_t)blockIdx.x * kSampleGatherThreads + threadIdx.x; i < nvec;
         i += (int64_t)gridDim.x * kSampleGatherThreads)
      ((u32x4*)dst)[i] = __builtin_nontemporal_load((const u32x4*)src + i);
  } else {
    for (int64_t i = (int64_t)blockIdx.x * kSampleGatherThreads + threadIdx.x; i < rb;
         i += (int64_t)gridDim.x * kSampleGatherThreads)
      dst[i] = src[i];
  }
}

__global__ __launch_bounds__(kMaxBatch) void prioritized_update_kernel(PrioUpdateParams p) {
  __shared__ int64_t s_leaf[kMaxBatch];
  __shared__ double s_red[kMaxBatch / 64];
  prio_update_body(p, s_leaf, s_red);
}

__global__ __launch_bounds__(kMaxBatch) void prioritized_add_kernel(
    double* node, int64_t cap, int64_t N, int64_t t, int n, double priority_h,
    const double* priority_d, double exponent, uint32_t* status) {
  __shared__ int64_t s_leaf[kMaxBatch];
  const int i = threadIdx.x;
  const bool active = i < n;
  const double p = priority_d ? *priority_d : priority_h;
  const double v = leaf_from_priority_f64(p, exponent);
  const int64_t leaf = tree_index_of_id(t + (active ? i : 0), N);
  if (active) s_leaf[i] = leaf;
  if (!finite_nonneg(v)) {  // uniform across the block
    if (i == 0) raise(status, DZ_ST_BAD_VALUE);
    return;
  }
  __syncthreads();
  set_leaves_and_ancestors(node, cap, leaf, v, active, s_leaf, n);
}

// One transition: rows of every field (blockIdx.y < num_fields) and, in the
// extra block row, the tree insert (leaf + its 20 ancestors, one thread: program
// order makes every parent exactly fl(left + right) of its final children).
struct InsertArgs { dz_insert_field_t f[DZ_MAX_FIELDS]; int num_fields; };
__global__ __launch_bounds__(256) void replay_insert_kernel(
    InsertArgs a, int64_t slot, double* node, int64_t cap, int64_t N, int64_t t,
    double priority_h, const double* priority_d, double exponent, uint32_t* status) {
  if ((int)blockIdx.y == a.num_fields) {
    if (blockIdx.x != 0 || threadIdx.x != 0 || !node) return;
    const double p = priority_d ? *priority_d : priority_h;
    const double v = leaf_from_priority_f64(p, exponent);
    if (!finite_nonneg(v)) { raise(status, DZ_ST_BAD_VALUE); return; }
    int64_t i = cap + tree_index_of_id(t, N);
    node[i] = v;
    for (i >>= 1; i >= 1; i >>= 1) node[i] = node[2 * i] + node[2 * i + 1];
    return;
  }
  const dz_insert_field_t fd = a.f[blockIdx.y];
  const int64_t rb = fd.row_bytes;
  char* dst = (char*)fd.dst + slot * rb;
  if (!fd.src_row) {
    if (blockIdx.x == 0 && (int64_t)threadIdx.x < rb)
      dst[threadIdx.x] = (char)((fd.imm >> (8 * threadIdx.x)) & 0xff);
    return;
  }
  const char* src = (const char*)fd.src_row;
  const bool vec_ok = ((rb & 15) == 0) && ((((uintptr_t)src) & 15) == 0) &&
                      ((((uintptr_t)fd.dst) & 15) == 0);
  if (vec_ok) {
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    const int64_t nvec = rb >> 4;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nvec;
         i += (int64_t)gridDim.x * 256)
      ((u32x4*)dst)[i] = ((const u32x4*)src)[i];
  } else {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < rb;
         i += (int64_t)gridDim.x * 256)
      dst[i] = src[i];
  }
}

inline int round_up_64(int n) { return (n + 63) / 64 * 64; }

}  // namespace

extern "C" int dz_sumtree_set(double* node, int64_t cap_pow2, int64_t size,
                              const int64_t* idx, const double* val, int n,
                              uint32_t* status, dz_stream_t stream) {
  DZ_REQUIRE(node && idx && val && dz_is_pow2(cap_pow2) && size >= 0 &&
             size <= cap_pow2);
  DZ_REQUIRE(n >= 0 && n <= kMaxBatch);
  if (n == 0) return DZ_OK;
  hipLaunchKernelGGL(sumtree_set_kernel, dim3(1), dim3(round_up_64(n)), 0,
                     dz_s(stream), node, cap_pow2, size, idx, val, n, status);
  DZ_LAUNCH_CHECK();
  return DZ_OK;
}

extern "C" int dz_sumtree_get(const double* node, int64_t cap_pow2, int64_t size,
                              const int64_t* idx, int n, double* out,
                              uint32_t* status, dz_stream_t stream) {
  DZ_REQUIRE(node && idx && out && dz_is_pow2(cap_pow2) && n >= 0);
  if (n == 0) return DZ_OK;
  hipLaunchKernelGGL(sumtree_get_kernel, dim3((n + 255) / 256), dim3(256), 0,
                     dz_s(stream), node, cap_pow2, size, idx, n, out, status);
  DZ_LAUNCH_CHECK();
  return DZ_OK;
}

extern "C" int dz_sumtree_rebuild(double* node, int64_t cap_pow2, int64_t size,
                                  dz_stream_t stream) {
  DZ_REQUIRE(node && dz_is_pow2(cap_pow2) && size >= 0 && size <= cap_pow2);
  const int64_t tail = cap_pow2 - size;
  const int64_t zt = tail > 0 ? tail : 1;
  hipLaunchKernelGGL(sumtree_zero_tail_kernel, dim3((unsigned)((zt + 255) / 256)),
                     dim3(256), 0, dz_s(stream), node, cap_pow2, size);
  DZ_LAUNCH_CHECK();
  for (int64_t first = cap_pow2 >> 1; first >= 1; first >>= 1) {
    hipLaunchKernelGGL(sumtree_level_kernel,
                       dim3((unsigned)((first + 255) / 256)), dim3(256), 0,
                       dz_s(stream), node, first);
    DZ_LAUNCH_CHECK();
  }
  return DZ_OK;
}

extern "C" int dz_sumtree_query(const double* node, int64_t cap_pow2,
                                const double* targets, int n, int64_t* out,
                                uint32_t* status, dz_stream_t stream) {
  DZ_REQUIRE(node && targets && out && dz_is_pow2(cap_pow2) && n >= 0);
  if (n == 0) return DZ_OK;
  hipLaunchKernelGGL(sumtree_query_kernel, dim3((n + 63) / 64), dim3(64), 0,
                     dz_s(stream), node, cap_pow2, targets, n, out, status);
  DZ_LAUNCH_CHECK();
  return DZ_OK;
}

extern "C" int dz_prioritized_sample(const dz_prio_sample_args_t* args, int batch,
                                     int64_t* ids_out, int64_t* tree_idx_out,
                                     double* probs_out, double* weights_out,
                                     float* weights32_out, uint32_t* status,
                                     dz_stream_t stream) {
  DZ_REQUIRE(args && ids_out && batch > 0 && batch <= kMaxBatch);
  DZ_REQUIRE(args->node && dz_is_pow2(args->cap_pow2) && args->capacity > 0 &&
             args->capacity <= args->cap_pow2);
  DZ_REQUIRE(args->size > 0 && args->size <= args->capacity &&
             args->t >= args->size);
  DZ_REQUIRE(args->pos && args->u_target && args->u_mix);
  dz_prof_pair(0, 0, dz_s(stream));
  hipLaunchKernelGGL(prioritized_sample_kernel<0>, dim3(1),
                     dim3(round_up_64(batch)), 0, dz_s(stream), *args, HostDraws{}, batch,
                     ids_out, tree_idx_out, probs_out, weights_out,
                     weights32_out, status);
  DZ_LAUNCH_CHECK();
  dz_prof_pair(0, 1, dz_s(stream));
  return DZ_OK;
}

extern "C" int dz_prioritized_sample_host_draws(
    const dz_prio_sample_args_t* args, int batch, const int64_t* pos_h,
    const double* u_target_h, const double* u_mix_h, int64_t* ids_out,
    int64_t* tree_idx_out, double* probs_out, double* weights_out, float* weights32_out,
    uint32_t* status, dz_stream_t stream) {
  DZ_REQUIRE(args && ids_out && batch > 0 && batch <= kMaxHostDraws);
  DZ_REQUIRE(args->node && dz_is_pow2(args->cap_pow2) && args->capacity > 0 &&
             args->capacity <= args->cap_pow2);
  DZ_REQUIRE(args->size > 0 && args->size <= args->capacity &&
             args->t >= args->size);
  DZ_REQUIRE(pos_h && u_target_h && u_mix_h);
  HostDraws hd;
  for (int i = 0; i < kMaxHostDraws; ++i) {
    const int j = i < batch ? i : 0;
    hd.pos[i] = pos_h[j]; hd.u_target[i] = u_target_h[j]; hd.u_mix[i] = u_mix_h[j];
  }
  dz_prof_pair(0, 0, dz_s(stream));
  hipLaunchKernelGGL(prioritized_sample_kernel<1>, dim3(1), dim3(round_up_64(batch)), 0,
                     dz_s(stream), *args, hd, batch, ids_out, tree_idx_out, probs_out,
                     weights_out, weights32_out, status);
  DZ_LAUNCH_CHECK();
  dz_prof_pair(0, 1, dz_s(stream));
  return DZ_OK;
}

extern "C" int dz_prioritized_sample_gather(
    const dz_prio_sample_args_t* args, int batch, const int64_t* pos_h,
    const double* u_target_h, const double* u_mix_h, const dz_field_t* fields,
    int num_fields, int64_t* ids_out, double* probs_out, double* weights_out,
    float* weights32_out, uint32_t* status, dz_stream_t stream) {
  DZ_REQUIRE(args && ids_out && batch > 0 && batch <= kMaxHostDraws);
  DZ_REQUIRE(args->node && dz_is_pow2(args->cap_pow2) && args->capacity > 0 &&
             args->capacity <= args->cap_pow2);
  DZ_REQUIRE(args->size > 0 && args->size <= args->capacity && args->t >= args->size);
  DZ_REQUIRE(pos_h && u_target_h && u_mix_h && fields && num_fields > 0 &&
             num_fields <= DZ_MAX_FIELDS);
  HostDraws hd;
  for (int i = 0; i < kMaxHostDraws; ++i) {
    const int j = i < batch ? i : 0;
    hd.pos[i] = pos_h[j]; hd.u_target[i] = u_target_h[j]; hd.u_mix[i] = u_mix_h[j];
  }
  SampleGatherFields gf;
  gf.num_fields = num_fields;
  int64_t max_rb = 0;
  for (int i = 0; i < num_fields; ++i) {
    DZ_REQUIRE(fields[i].src && fields[i].dst && fields[i].row_bytes > 0);
    gf.f[i] = fields[i];
    if (fields[i].row_bytes > max_rb) max_rb = fields[i].row_bytes;
  }
  int64_t chunks = ((max_rb >> 4) + kSampleGatherThreads - 1) / kSampleGatherThreads;
  if (chunks < 1) chunks = 1;
  if (chunks > 64) chunks = 64;
  dz_prof_pair(0, 0, dz_s(stream));
  hipLaunchKernelGGL(prioritized_sample_gather_kernel,
                     dim3((unsigned)chunks, (unsigned)batch + 1, (unsigned)num_fields),
                     dim3(kSampleGatherThreads), 0, dz_s(stream), *args, hd, batch, gf, ids_out, probs_out, weights_out,
                     weights32_out, status);
  DZ_LAUNCH_CHECK();
  dz_prof_pair(0, 1, dz_s(stream));
  return DZ_OK;
}

extern "C" int dz_prioritized_update(double* node, int64_t cap_pow2,
                                     int64_t capacity, int64_t size, int64_t t,
                                     const int64_t* ids, const void* priorities,
                                     int prio_is_f32, double exponent, int n,
                                     double* max_seen, uint32_t* status,
                                     dz_stream_t stream) {
  DZ_REQUIRE(node && ids && priorities && dz_is_pow2(cap_pow2));
  DZ_REQUIRE(capacity > 0 && capacity <= cap_pow2 && size >= 0 &&
             size <= capacity && t >= size);
  DZ_REQUIRE(n >= 0 && n <= kMaxBatch && exponent >= 0.0);
  if (n == 0) return DZ_OK;
  dz_prof_pair(2, 0, dz_s(stream));
  const PrioUpdateParams q = {node, cap_pow2, capacity, size, t, ids, priorities, prio_is_f32,
                              exponent, n, max_seen, status, 1};
  hipLaunchKernelGGL(prioritized_update_kernel, dim3(1), dim3(round_up_64(n)), 0,
                     dz_s(stream), q);
  DZ_LAUNCH_CHECK();
  dz_prof_pair(2, 1, dz_s(stream));
  return DZ_OK;
}

extern "C" int dz_prioritized_add(double* node, int64_t cap_pow2,
                                  int64_t capacity, int64_t t, int n,
                                  double priority_h, const double* priority_d,
                                  double exponent, uint32_t* status,
                                  dz_stream_t stream) {
  DZ_REQUIRE(node && dz_is_pow2(cap_pow2) && capacity > 0 &&
             capacity <= cap_pow2 && t >= 0);
  DZ_REQUIRE(n >= 0 && n <= kMaxBatch && n <= capacity && exponent >= 0.0);
  if (n == 0) return DZ_OK;
  hipLaunchKernelGGL(prioritized_add_kernel, dim3(1), dim3(round_up_64(n)), 0,
                     dz_s(stream), node, cap_pow2, capacity, t, n, priority_h,
                     priority_d, exponent, status);
  DZ_LAUNCH_CHECK();
  return DZ_OK;
}

extern "C" int dz_replay_insert(const dz_insert_field_t* fields, int num_fields, int64_t t,
                                int64_t capacity, double* node, int64_t cap_pow2,
                                double priority_h, const double* priority_d,
                                double exponent, uint32_t* status, dz_stream_t stream) {
  DZ_REQUIRE(fields && num_fields > 0 && num_fields <= DZ_MAX_FIELDS && capacity > 0 &&
             t >= 0);
  if (node) DZ_REQUIRE(dz_is_pow2(cap_pow2) && capacity <= cap_pow2 && exponent >= 0.0 && status);
  InsertArgs a;
  a.num_fields = num_fields;
  int64_t max_rb = 0;
  for (int i = 0; i < num_fields; ++i) {
    DZ_REQUIRE(fields[i].dst && fields[i].row_bytes > 0);
    DZ_REQUIRE(fields[i].src_row || fields[i].row_bytes <= 8);
    a.f[i] = fields[i];
    if (fields[i].src_row && fields[i].row_bytes > max_rb) max_rb = fields[i].row_bytes;
  }
  int64_t chunks = ((max_rb >> 4) + 255) / 256;
  if (chunks < 1) chunks = 1;
  if (chunks > 16) chunks = 16;
  hipLaunchKernelGGL(replay_insert_kernel, dim3((unsigned)chunks, (unsigned)num_fields + 1),
                     dim3(256), 0, dz_s(stream), a, t % capacity, node, cap_pow2, capacity, t,
                     priority_h, priority_d, exponent, status);
  DZ_LAUNCH_CHECK();
  return DZ_OK;
}
