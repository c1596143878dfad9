/*
 * dqnzoo_hip.h -- C ABI of libdqnzoo_hip.so, the MI355X (gfx950) implementation
 * of dqn_zoo's replay-sampling + Q-loss/update hot path.
 *
 * The reference (google-deepmind/dqn_zoo) has no FFI layer: its boundary is a
 * Python object protocol.  This header is the boundary a
 * maintainer would bind instead (ctypes stub in INTEGRATION.md).  Every entry
 * point cites the reference code it replaces as `ref: file:line`.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes, no C++ types, no exceptions.
 *   - Every `*_d` / unqualified buffer pointer is a DEVICE pointer (HBM) owned by
 *     the caller (in this repo: PyTorch-ROCm tensors, passed as data_ptr()).
 *   - `stream` is a hipStream_t passed as void*; work is only ENQUEUED, nothing
 *     here synchronises unless the comment says so.
 *   - Return value: 0 (DZ_OK) or a negative DZ_ERR_* code.  Data-dependent
 *     errors that the reference raises from inside a loop (e.g. ValueError for
 *     an out-of-range query target) are reported through a caller-provided
 *     device `status` word (DZ_ST_* bits, sticky, OR-ed in) because the host
 *     cannot know them without a sync.
 *   - Not thread-safe; one process per GPU.
 */
#ifndef DQNZOO_HIP_H_
#define DQNZOO_HIP_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DZ_OK 0
#define DZ_ERR_INVALID_ARG (-1)
#define DZ_ERR_HIP (-2)
#define DZ_ERR_UNSUPPORTED (-3)

/* bits of the device status word */
#define DZ_ST_BAD_VALUE 1u      /* ref: replay.py:281-282 ValueError('value must be finite and positive.') */
#define DZ_ST_BAD_TARGET 2u     /* ref: replay.py:408-409 ValueError('Require 0 <= target < total sum.') */
#define DZ_ST_BAD_INDEX 4u      /* ref: replay.py:274-275 IndexError('index out of range ...') */
#define DZ_ST_ZERO_ROOT 8u      /* pipelined sampling met root()==0 (ref takes a different RNG path, replay.py:556-557) */
#define DZ_ST_NONFINITE_WEIGHT 16u /* ref: replay.py:241-242 ValueError('Weights are not finite') */

typedef void* dz_stream_t;

/* Library identification; also what the loader's "does it export every
 * symbol" test keys on. */
const char* dz_version(void);
/* hipGetLastError() of the most recent failing call (as int), for messages. */
int dz_last_hip_error(void);
/* Name of the gfx target the device code was built for ("gfx950"). */
const char* dz_built_arch(void);

/* ------------------------------------------------------------------------- *
 *  Replay storage: HBM-resident field arrays, one row per transition slot.
 *  slot(id) = id mod capacity (FIFO ring; the oldest id is the one evicted,
 *  ref: replay.py:143-145 `popitem(last=False)`).
 * ------------------------------------------------------------------------- */

#define DZ_MAX_FIELDS 8
typedef struct {
  const void* src;     /* field array base, [capacity][row_bytes]            */
  void* dst;           /* batch output, [batch][row_bytes]                   */
  int64_t row_bytes;   /* bytes per row (28224 for an 84x84x4 uint8 state)   */
} dz_field_t;

/* Coalesced transition gather: dst[b] = src[ids[b] mod capacity] for every
 * field, one launch.  Replaces the decode + per-field np.stack of
 * ref: replay.py:152-163 (TransitionReplay.get/sample) and 701-723.        */
int dz_replay_gather(const dz_field_t* fields, int num_fields,
                     const int64_t* ids, int batch, int64_t capacity,
                     dz_stream_t stream);

/* Position -> id map of the reference's swap-remove id list under its only
 * usage pattern (one add at a time, evict oldest): closed form verified against
 * the reference.  ids_out[b] = _ids[pos[b]].
 * ref: replay.py:52-82 (UniformDistribution.add/remove/sample).
 * `t` = number of items ever added, `size` = items currently stored.         */
int dz_uniform_pos_to_id(const int64_t* pos, int batch, int64_t t, int64_t size,
                         int64_t capacity, int64_t* ids_out, dz_stream_t stream);

/* ------------------------------------------------------------------------- *
 *  Sum tree: float64 implicit heap, node[1] = root, leaves at
 *  [cap_pow2, 2*cap_pow2).  ref: replay.py:246-426 (class SumTree).
 * ------------------------------------------------------------------------- */

/* node[cap_pow2 + idx[i]] = val[i] (duplicates: LAST wins, as NumPy fancy
 * assignment) then every ancestor of every idx is recomputed as left+right.
 * Nothing is written and DZ_ST_BAD_VALUE is raised if any val is negative or
 * non-finite.  ref: replay.py:278-290 (SumTree.set).                        */
int dz_sumtree_set(double* node, int64_t cap_pow2, int64_t size,
                   const int64_t* idx, const double* val, int n,
                   uint32_t* status, dz_stream_t stream);

/* out[i] = node[cap_pow2 + idx[i]].  ref: replay.py:271-276 (SumTree.get).   */
int dz_sumtree_get(const double* node, int64_t cap_pow2, int64_t size,
                   const int64_t* idx, int n, double* out, uint32_t* status,
                   dz_stream_t stream);

/* Recomputes every internal node bottom-up from the leaves and zeroes leaves
 * [size, cap_pow2) and node[0].  ref: replay.py:395-404 (SumTree._set_values). */
int dz_sumtree_rebuild(double* node, int64_t cap_pow2, int64_t size,
                       dz_stream_t stream);

/* out[i] = smallest leaf index whose cumulative sum exceeds targets[i]
 * (left-if-target<left_sum descent).  DZ_ST_BAD_TARGET unless
 * 0 <= target < root.  ref: replay.py:299-313, 406-426 (SumTree.query).      */
int dz_sumtree_query(const double* node, int64_t cap_pow2,
                     const double* targets, int n, int64_t* out,
                     uint32_t* status, dz_stream_t stream);

/* ------------------------------------------------------------------------- *
 *  Prioritized distribution in its fixed-capacity form (the only one
 *  reachable through PrioritizedTransitionReplay, ref: replay.py:678-684).
 *  tree index of id i = capacity-1-(i mod capacity)  (free stack is popped
 *  from the end, ref: replay.py:457,499).
 * ------------------------------------------------------------------------- */

typedef struct {
  double* node;            /* sum-tree storage, 2*cap_pow2 doubles            */
  int64_t cap_pow2;
  int64_t capacity;        /* replay capacity N                               */
  int64_t size;            /* items currently stored                          */
  int64_t t;               /* items ever added (id of the next item)          */
  /* host-drawn randomness, uploaded by the caller in the reference's draw
   * order (ref: replay.py:551-566): randint(size), uniform(), uniform()     */
  const int64_t* pos;      /* [batch] positions into the active list          */
  const double* u_target;  /* [batch] raw uniforms; target = u * root()       */
  const double* u_mix;     /* [batch] raw uniforms; uniform branch iff < usp  */
  double usp;              /* uniform_sample_probability                      */
  double one_minus_usp;    /* (1.0 - usp), computed by the host in float64    */
  double usp_times_up;     /* usp * (1.0/size), computed by the host          */
  double uniform_prob;     /* 1.0/size                                        */
  double beta;             /* importance_sampling_exponent(t)                 */
  int normalize;           /* normalize_weights                               */
  int compute_weights;     /* 0: probs only (host computes exact weights)     */
  int assume_nonzero_root; /* 1: pipelined mode, raise DZ_ST_ZERO_ROOT if 0   */
} dz_prio_sample_args_t;

/* Fused: position->tree index, target scaling, 20-level descent, uniform mix,
 * probabilities, importance weights (device pow; see DESIGN.md for the exact
 * mode), ids.  Outputs (any may be NULL except ids_out):
 *   ids_out int64[batch], tree_idx_out int64[batch], probs_out f64[batch],
 *   weights_out f64[batch], weights32_out f32[batch].
 * ref: replay.py:547-583 (PrioritizedDistribution.sample) +
 *      replay.py:211-243 (importance_sampling_weights).                     */
int dz_prioritized_sample(const dz_prio_sample_args_t* args, int batch,
                          int64_t* ids_out, int64_t* tree_idx_out,
                          double* probs_out, double* weights_out,
                          float* weights32_out, uint32_t* status,
                          dz_stream_t stream);

/* leaf(id) = power_zero_safe(priority, exponent) for each id, then SumTree.set.
 * `prio_is_f32`: priorities are float32 and -- as NumPy does for an f32 array
 * raised to a Python-float exponent -- the power is evaluated in float32.
 * exponent 0.5 / 1.0 / 0.0 are exact (sqrt / identity / 1); other exponents
 * use the device pow (not bit-identical to NumPy's SIMD pow: DESIGN.md).
 * If `max_seen` is non-NULL: *max_seen = max(*max_seen, max_i priority_i)
 * (ref: rainbow/agent.py:196-197).  Unknown ids raise DZ_ST_BAD_INDEX
 * (ref: replay.py:541-543).
 * ref: replay.py:536-545 (update_priorities), 203-208 (_power).             */
int dz_prioritized_update(double* node, int64_t cap_pow2, int64_t capacity,
                          int64_t size, int64_t t, const int64_t* ids,
                          const void* priorities, int prio_is_f32,
                          double exponent, int n, double* max_seen,
                          uint32_t* status, dz_stream_t stream);

/* Tree side of `add(item, priority)` for ids t, t+1, ..., t+n-1 (evicting
 * the oldest when full re-uses the same tree index, so the net effect is one
 * leaf set per add).  priority = *priority_d if non-NULL else priority_h.
 * ref: replay.py:690-699, 475-534.                                          */
int dz_prioritized_add(double* node, int64_t cap_pow2, int64_t capacity,
                       int64_t t, int n, double priority_h,
                       const double* priority_d, double exponent,
                       uint32_t* status, dz_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* DQNZOO_HIP_H_ */
