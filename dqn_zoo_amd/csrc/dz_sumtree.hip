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

// One launch per level: node[i] = node[2i] + node[2i+1] for i in [first, 2*first).
__global__ void sumtree_level_kernel(double* node, int64_t first) {
  const int64_t i = first + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < 2 * first) node[i] = node[2 * i] + node[2 * i + 1];
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

// Sample AND gather in one launch (batch <= 64, draws in the kernel arguments): the
// blocks of sample_gather_block (dz_sumtree_dev.h), 512 threads each -- the sampler
// block walks 32 elements x 16 lanes in one pass.
constexpr int kSampleGatherThreads = 512;
__global__ __launch_bounds__(kSampleGatherThreads) void prioritized_sample_gather_kernel(
    SampleGatherParams q) {
  sample_gather_block<kSampleGatherThreads>(q, blockIdx.x);
}

__global__ __launch_bounds__(kMaxBatch) void prioritized_update_kernel(PrioUpdateParams p) {
  __shared__ int64_t s_leaf[kMaxBatch];
  __shared__ double s_red[kMaxBatch / 64];
  prio_update_body(p, s_leaf, s_red);
}
// batches <= kWbMaxBatch (255): the two-round-trip walk (launched with 256 threads)
__global__ __launch_bounds__(256) void prioritized_update_fast_kernel(PrioUpdateParams p) {
  __shared__ WbScratch wb;
  PrioUpdateSideFast::run(p, 0, &wb, (int)sizeof(wb));
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
  SampleGatherParams q;
  q.a = *args; q.hd = hd;
  for (int i = 0; i < num_fields; ++i)
    DZ_REQUIRE(fields[i].src && fields[i].dst && fields[i].row_bytes > 0);
  const unsigned blocks = sample_gather_plan(q, fields, num_fields, batch, kSampleGatherThreads);
  q.ids_out = ids_out; q.probs_out = probs_out; q.weights_out = weights_out;
  q.weights32_out = weights32_out; q.status = status;
  dz_prof_pair(0, 0, dz_s(stream));
  hipLaunchKernelGGL(prioritized_sample_gather_kernel, dim3(blocks), dim3(kSampleGatherThreads),
                     0, dz_s(stream), q);
  DZ_LAUNCH_CHECK();
  dz_prof_pair(0, 1, dz_s(stream));
  return DZ_OK;
}

extern "C" int dz_sample_gather_desc(const dz_next_sample_t* d, dz_stream_t stream) {
  DZ_REQUIRE(d);
  return dz_prioritized_sample_gather(&d->args, d->n, d->pos_h, d->u_target_h, d->u_mix_h, d->fields,
                                      d->num_fields, d->ids_out, d->probs_out, d->weights_out,
                                      d->weights32_out, d->status, stream);
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
  if (n <= kWbMaxBatch && cap_pow2 <= ((int64_t)1 << 31))
    hipLaunchKernelGGL(prioritized_update_fast_kernel, dim3(1), dim3(256), 0, dz_s(stream), q);
  else
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
                                double exponent, uint32_t* status, dz_stream_t stream);
extern "C" int dz_replay_insert_v(const dz_replay_insert_args_t* a, dz_stream_t stream) {
  DZ_REQUIRE(a);
  return dz_replay_insert(a->fields, a->num_fields, a->t, a->capacity, a->node, a->cap_pow2,
                          a->priority_h, a->priority_d, a->exponent, a->status, stream);
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
