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
#define DZ_ST_CHAIN_TIMEOUT 32u /* a multi-role learner launch gave up on one of its in-launch seams (no reference
                                 * counterpart; dz_rainbow_args_t::separate_launches); the step's losses are NaN */

typedef void* dz_stream_t;

/* Library identification; also what the loader's "does it export every
 * symbol" test keys on. */
const char* dz_version(void);
/* hipGetLastError() of the most recent failing call (as int), for messages. */
int dz_last_hip_error(void);
/* Name of the gfx target the device code was built for ("gfx950"). */
const char* dz_built_arch(void);
/* sizeof() of the ABI structs as the library was compiled, so that a binding
 * can verify its mirror: 0 dz_field_t, 1 dz_prio_sample_args_t,
 * 2 dz_rainbow_layout_t, 3 dz_rainbow_args_t, 4 dz_dense_layout_t, 9 dz_next_sample_t,
 * 5 dz_dense_args_t, 6 dz_iqn_layout_t, 7 dz_iqn_args_t, 8 dz_insert_field_t,
 * 10 dz_replay_insert_args_t, 11 dz_rainbow_act_args_t.
 * -1 for an unknown id.                                                      */
int dz_struct_size(int which);

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

/* Uniform sample in ONE launch (batch <= 64): `pos_host[b]` are the host's
 * `randint(size, size=batch)` draws (read before the call returns; they travel in
 * the kernel arguments), ids_out[b] = _ids[pos[b]] by the closed form of
 * dz_uniform_pos_to_id (may be NULL), dst[b] = src[ids[b] mod capacity].
 * ref: replay.py:76-82 + 152-163 (TransitionReplay.sample).                  */
int dz_replay_sample_uniform(const dz_field_t* fields, int num_fields,
                             const int64_t* pos_host, int batch, int64_t t,
                             int64_t size, int64_t capacity, int64_t* ids_out,
                             dz_stream_t stream);

/* One transition into the store, and (node != NULL) its sum-tree leaf, in ONE
 * launch: field i's row `t mod capacity` is written from `src_row` (a DEVICE
 * pointer to row_bytes bytes, e.g. the observation the actor just uploaded) or,
 * when src_row is NULL, from the little-endian bytes of `imm` (row_bytes <= 8:
 * the action / reward / discount scalars travel as kernel arguments).  The tree
 * part is dz_prioritized_add for n = 1.  Replaces the per-field writes of
 * ref: replay.py:141-150 (TransitionReplay.add) and 690-699.                 */
typedef struct {
  void* dst;            /* field array base, [capacity][row_bytes]            */
  const void* src_row;  /* device pointer, or NULL to use imm                 */
  int64_t row_bytes;
  uint64_t imm;
} dz_insert_field_t;

int dz_replay_insert(const dz_insert_field_t* fields, int num_fields, int64_t t,
                     int64_t capacity, double* node, int64_t cap_pow2,
                     double priority_h, const double* priority_d, double exponent,
                     uint32_t* status, dz_stream_t stream);

/* dz_replay_insert with its arguments in a struct the caller keeps and patches (`t`, the
 * fields' src_row / imm): for bindings that pay per marshalled argument (ctypes: ~0.15 us
 * each -- the agents' loop inserts once per frame).  Same launch, same checks.
 * ref: replay.py:141-150 (TransitionReplay.add), 690-699 (PrioritizedTransitionReplay.add).  */
typedef struct {
  const dz_insert_field_t* fields;
  int32_t num_fields;
  int32_t reserved;
  int64_t t;
  int64_t capacity;
  double* node;
  int64_t cap_pow2;
  double priority_h;
  const double* priority_d;
  double exponent;
  uint32_t* status;
} dz_replay_insert_args_t;
int dz_replay_insert_v(const dz_replay_insert_args_t* args, dz_stream_t stream);

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

/* The same for batch <= 64 with the three RNG draw arrays given as HOST
 * pointers (args->pos/u_target/u_mix are ignored): the draws travel in the
 * kernel arguments, so the per-step H2D copy of the draws and its blit launch
 * disappear.  The arrays are read before the call returns.                   */
int dz_prioritized_sample_host_draws(
    const dz_prio_sample_args_t* args, int batch, const int64_t* pos_host,
    const double* u_target_host, const double* u_mix_host, int64_t* ids_out,
    int64_t* tree_idx_out, double* probs_out, double* weights_out,
    float* weights32_out, uint32_t* status, dz_stream_t stream);

/* dz_prioritized_sample_host_draws AND dz_replay_gather in ONE launch: the
 * gather blocks re-derive their element's tree index with the sampler's own
 * arithmetic instead of waiting for ids[] from a previous launch.
 * ref: replay.py:706-723 (PrioritizedTransitionReplay.sample).                */
int dz_prioritized_sample_gather(
    const dz_prio_sample_args_t* args, int batch, const int64_t* pos_host,
    const double* u_target_host, const double* u_mix_host, const dz_field_t* fields,
    int num_fields, int64_t* ids_out, double* probs_out, double* weights_out,
    float* weights32_out, uint32_t* status, dz_stream_t stream);

/* dz_prioritized_sample_gather with its arguments in the descriptor a learner step can also
 * carry (dz_next_sample_t, declared with dz_rainbow_args_t below): a binding keeps one per
 * output slot and patches `args` per call.
 * ref: replay.py:706-723 (PrioritizedTransitionReplay.sample).                           */
struct dz_next_sample;
int dz_sample_gather_desc(const struct dz_next_sample* desc, dz_stream_t stream);

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


/* ------------------------------------------------------------------------- *
 *  Rainbow learner step: 3 network applies + categorical double-Q loss +
 *  backward + clip_by_global_norm + Adam, all enqueued on one stream.
 *  ref: rainbow/agent.py:85-121 (loss_fn, update), networks.py:224-261
 *  (rainbow_atari_network), rainbow/run_atari.py:229-235 (optimizer chain).
 * ------------------------------------------------------------------------- */

/* Offsets (in floats) of every tensor inside the flat parameter buffer, of
 * every noise vector inside one apply's noise block, and of every intermediate
 * inside the workspace.  Single source of truth for Python and for tests that
 * inspect intermediates.  All weight matrices are row-major [in][out] (the
 * reference's own layout, networks_test.py:44,53).                         */
typedef struct {
  int32_t num_actions, num_atoms, batch, groups;
  int32_t adv2_ld, val2_ld; /* leading dimensions (floats) of the fc2 matrices */
  int32_t fc1_ld, pad0_;    /* leading dimension of the fused fc1 matrices     */
  int64_t param_count;      /* floats in a parameter buffer (16-byte padded)  */
  int64_t param_count_ref;  /* the reference's count (6 868 485 for A=6)      */
  /* parameters */
  int64_t conv_w[3], conv_b[3];         /* [kh*kw*cin][cout], [cout]          */
  int64_t fc1_mu_w, fc1_mu_b;           /* [3136][fc1_ld], cols [adv1 512 | val1 512]; [1024] */
  int64_t fc1_sig_w, fc1_sig_b;
  int64_t adv2_mu_w, adv2_sig_w;        /* [512][adv2_ld], first A*K columns  */
  int64_t val2_mu_w, val2_sig_w;        /* [512][val2_ld], first K columns    */
  int64_t fc2_sig_b;                    /* [A*K + K] = [adv2 | val2]          */
  /* one apply's noise block */
  int64_t noise_stride;
  int64_t n_adv1_in, n_val1_in, n_fc1_out, n_adv2_in, n_val2_in, n_fc2_out;
  /* workspace (floats) */
  int64_t ws_count;
  int64_t ws_act1, ws_act2, ws_feat, ws_fc1_part, ws_h1, ws_fc2_part, ws_fc2_out;
  int64_t ws_dout2, ws_dh1, ws_dfeat_part, ws_dfeat, ws_dact2, ws_dact1;
  int64_t ws_wgrad_part, ws_norm_part, ws_scalars, ws_q_sel, ws_target_probs;
  int64_t ws_colsum_part;
  int64_t ws_act_seams;     /* counters and flags of the one-launch decision (dz_rainbow_act, batch 1): zero
                             * when the workspace is created, owned by that kernel afterwards          */
} dz_rainbow_layout_t;

int dz_rainbow_layout(int num_actions, int num_atoms, int batch,
                      dz_rainbow_layout_t* out);

/* ws_scalars[]: */
#define DZ_SC_GNORM 0     /* global gradient norm before clipping             */
#define DZ_SC_LOSS 1      /* mean(losses * weights)                           */
#define DZ_SC_BC1 2       /* 1 - b1^count                                     */
#define DZ_SC_BC2 3       /* 1 - b2^count                                     */
#define DZ_SC_CLIP 4      /* 1 if gnorm < max_norm else 0                     */
#define DZ_SC_CHAIN_FAIL 6 /* (uint32 bits) sticky, non-zero once a multi-role launch of a step timed out */

/* The NEXT step's replay sample, carried by a learner step (dz_rainbow_args_t::
 * next_sample): exactly the arguments of dz_prioritized_sample_gather.  The draws
 * are HOST arrays (n each), copied into the kernel arguments at enqueue time.     */
typedef struct dz_next_sample {
  dz_prio_sample_args_t args;
  const int64_t* pos_h;
  const double* u_target_h;
  const double* u_mix_h;
  const dz_field_t* fields;
  int32_t num_fields, n;
  int64_t* ids_out;
  double* probs_out;
  double* weights_out;
  float* weights32_out;
  uint32_t* status;
} dz_next_sample_t;

typedef struct {
  int32_t num_actions, num_atoms, batch;
  /* parameter-shaped buffers (dz_rainbow_layout.param_count floats each) */
  float* online;
  const float* target;
  float* grad;
  float* adam_m;
  float* adam_v;
  int32_t* adam_count;       /* device int32, incremented by the step         */
  /* the sampled batch, as PrioritizedTransitionReplay hands it over           */
  const uint8_t* s_tm1;      /* [B][84][84][4]                                */
  const uint8_t* s_t;
  const int64_t* a_tm1;      /* int64 -> int32 like the jit boundary          */
  const double* r_t;         /* float64 -> float32 like the jit boundary      */
  const double* discount_t;
  const float* weights;      /* importance weights (float32)                  */
  const float* support;      /* [K] atoms, passed as data                     */
  /* noise: 3 applies x noise_stride, order online(s_tm1), online(s_t),
   * target(s_t) (rainbow/agent.py:87-96)                                      */
  const float* noise;
  float* ws;                 /* workspace, ws_count floats                    */
  float* losses;             /* [B] per-sample cross-entropy (aux output)     */
  float* priorities;         /* [B] clip(|loss|,0,100) (rainbow/agent.py:194) */
  /* optimizer: optax.chain(clip_by_global_norm(max_norm), adam(lr, eps))     */
  float lr, b1, b2, eps, max_norm;
  /* if non-zero, the forward phase first regenerates the 3 noise blocks on the
   * device from (noise_seed, *adam_count) -- fresh noise per step without any
   * per-step host argument, so that the step can be replayed from a hipGraph  */
  int32_t resample_noise;
  uint64_t noise_seed;
  /* Optional priority write-back inside the backward launch (prio_node != NULL
   * and the call includes DZ_PHASE_BACKWARD): the step then performs
   * dz_prioritized_update(prio_node, ..., prio_ids, priorities [float32],
   * prio_exponent, batch, prio_max_seen, prio_status) itself, as one extra block
   * of a weight-gradient launch, instead of a separate single-workgroup kernel
   * after the step (ref: rainbow/agent.py:194-198).  The ids must be the ones the
   * batch was sampled with (they are not re-validated against the live range).  */
  double* prio_node;
  int64_t prio_cap_pow2, prio_capacity;
  const int64_t* prio_ids;
  double prio_exponent;
  double* prio_max_seen;
  uint32_t* prio_status;
  /* 0 (default): `grad` is not a complete gradient vector after the call.
   *   - A call that runs the loss, the backward pass AND the optimiser (batch <= 32)
   *     writes NEITHER of fc1's two weight-gradient blocks (2 x 12.85 MB): the optimiser
   *     forms each entry from the layer's input and output gradient at the moment it
   *     updates the weight, and the blocks' share of the global norm comes from Gram
   *     matrices (same mathematics, float32 rounding order of the 32-term batch sums
   *     apart from the stored form).
   *   - A call that runs backward + optimiser without the loss (a step split over several
   *     calls), or a batch > 32, stores fc1's mu-weight gradient and derives the sigma
   *     block in the optimiser (bit-identical to storing it).
   * 1: every gradient block is materialised in `grad` (inspection, tests, A/B).       */
  int32_t keep_all_grads;
  /* 0 (default): in a call that runs the whole step (as above, batch <= 32) the four launches
   * between fc1's forward stream and fc1's input gradient -- fc1 epilogue, noisy fc2, loss, fc2
   * backward -- are workgroup ROLES of ONE launch that hand h1, the fc2 slabs and dlogits to each
   * other through in-launch seams (csrc/dz_head_chain.h; bit-identical results).  Liveness as for
   * dz_rainbow_act's one-launch form: progress needs only the in-order workgroup dispatch of CDNA
   * hardware; every in-launch wait is bounded, and a timeout makes the step's losses NaN, sets
   * ws_scalars[DZ_SC_CHAIN_FAIL] and raises DZ_ST_CHAIN_TIMEOUT in prio_status (if given).
   * 1: one launch per stage, as in every other shape of the call (A/B, tests).               */
  int32_t separate_launches;
  /* Optional (needs DZ_PHASE_BACKWARD | DZ_PHASE_OPTIMIZER in the call): the sample +
   * gather of the NEXT step rides in this step's optimiser launch as extra blocks, and
   * the priority write-back (prio_*) moves into an EARLIER backward launch, so that
   * sample(k+1) sees write-back(k) as in the sequential order (rainbow/agent.py:
   * 181-198).  Only valid when nothing is added to the replay between this step and
   * the next sample (a learner over a static replay); the caller then uses the
   * buffers of `next_sample` as the next batch instead of sampling.  Eager launches
   * only (the draws are by-value kernel arguments: not graph-replayable).          */
  const dz_next_sample_t* next_sample;
} dz_rainbow_args_t;

#define DZ_PHASE_FORWARD 1   /* the applies + loss (+ dlogits) = NETS | LOSS   */
#define DZ_PHASE_BACKWARD 2  /* gradients into args->grad                     */
#define DZ_PHASE_OPTIMIZER 4 /* global norm + clip + Adam into args->online   */
#define DZ_PHASE_ALL 7
/* the two halves of DZ_PHASE_FORWARD, for callers that enqueue them separately  */
#define DZ_PHASE_FWD_NETS 8  /* the network applies up to the fc2 partial slabs */
#define DZ_PHASE_FWD_LOSS 16 /* loss kernel (folds the slabs): losses, priorities, dlogits */

int dz_rainbow_learn(const dz_rainbow_args_t* args, int phases, dz_stream_t stream);

/* One network apply (inference): q_values_out[b][a] for `batch` uint8 states
 * with ONE noise block, plus optionally the greedy action (first maximum) and
 * max_a q per state.  Uses the same kernels and workspace layout as the learner
 * (the workspace must hold dz_rainbow_layout(.., batch).ws_count floats and must
 * not be the one a concurrently enqueued learn step uses).
 * ref: rainbow/agent.py:125-131 (select_action), networks.py:224-261.       */
int dz_rainbow_apply(int num_actions, int num_atoms, int batch, const float* params,
                     const uint8_t* states, const float* noise, const float* support,
                     float* ws, float* q_values_out, int32_t* greedy_out,
                     float* vmax_out, dz_stream_t stream);

/* The actor's apply: dz_rainbow_apply with the noise block (noise_stride floats)
 * first redrawn on the device from (noise_seed, noise_counter) -- inside the
 * conv1 launch -- and the fc2 split-K fold inside the q-value kernel.
 * step_counter (nullable device int32): if given, the noise stream position is
 * noise_counter + *step_counter * noise_stride and the apply's last launch does
 * ++*step_counter, so that the SAME argument list draws fresh noise every time:
 * the call can be captured once (dz_graph_capture_begin/end) and replayed per
 * decision.  greedy_out / vmax_out may point into pinned, device-mapped host
 * memory (the action is then on the host when the stream reaches that point).
 * batch == 1 with a step_counter (the agent's decision): the whole apply is ONE launch
 * (csrc/dz_act_one.h); `ws` must have been all-zero when it was created and its
 * ws_act_seams region must be touched by nothing else (the kernel keeps a generation
 * word and two alternating sets of intermediates there).  With greedy_out and vmax_out
 * adjacent and 8-byte aligned the pair is written with ONE 8-byte store: a host that set
 * the action word to -1 before the call may poll it with plain loads.
 * Liveness of the one-launch form: its workgroup roles (torso -> fc1 -> tail) wait for each
 * other inside the launch, with dependencies pointing from lower to higher block ids only.
 * That is deadlock-free under the dispatch order CDNA hardware implements (linear ids dealt
 * round-robin to the XCDs, each XCD starting its share in ascending order) WITHOUT any
 * co-residency requirement -- other streams may fill the chip -- but HIP does not promise
 * that order, so every in-kernel wait is bounded: a seam that stays empty for
 * dz_act_debug_spin_limit() polling rounds sets the sticky word at ws_act_seams + 5 * 64 and
 * the decision comes back as greedy = DZ_ACT_FAILED with NaN value and NaN q-values --
 * never as an action -- for this and every later call until the caller zeroes the
 * ws_act_seams region (ws_count - ws_act_seams floats) again.
 * ref: rainbow/agent.py:125-131, 171-179 (select_action with a fresh key).    */
#define DZ_ACT_FAILED (-2)
int dz_rainbow_act(int num_actions, int num_atoms, int batch, const float* params,
                   const uint8_t* states, float* noise, uint64_t noise_seed,
                   uint64_t noise_counter, int32_t* step_counter,
                   const float* support, float* ws,
                   float* q_values_out, int32_t* greedy_out, float* vmax_out,
                   dz_stream_t stream);

/* dz_rainbow_act with its arguments in a struct the caller fills once per (observation slot,
 * result slot): the agent's per-frame decision is then a two-argument call.
 * ref: rainbow/agent.py:125-131, 171-179 (select_action with a fresh key).                 */
typedef struct {
  int32_t num_actions, num_atoms, batch, reserved;
  const float* params;
  const uint8_t* states;
  float* noise;
  uint64_t noise_seed, noise_counter;
  int32_t* step_counter;
  const float* support;
  float* ws;
  float* q_values_out;
  int32_t* greedy_out;
  float* vmax_out;
} dz_rainbow_act_args_t;
int dz_rainbow_act_v(const dz_rainbow_act_args_t* args, dz_stream_t stream);

/* Test hook of the one-launch decision kernels (dz_rainbow_act batch 1, dz_dense_act):
 * sets the number of polling rounds a workgroup spends on an empty seam before it gives up
 * (default 200000, >= 5 ms) and returns the previous value; limit < 0 only queries.  0 makes
 * every consumer give up at its first look, i.e. forces the failure path described above. */
int dz_act_debug_spin_limit(int limit);

/* hipGraph form of dz_rainbow_learn: captures the launches of one call (same
 * args, same phases; every pointer in `args` is baked in) on `stream` and
 * returns an executable graph; dz_graph_launch replays it with one API call
 * (host cost ~10 us instead of ~35 kernel launches).  Requires
 * args->resample_noise or externally supplied noise; the event profiler must
 * be off.  dz_graph_destroy releases it.                                    */
int dz_rainbow_graph_capture(const dz_rainbow_args_t* args, int phases,
                             dz_stream_t stream, void** graph_exec_out);
int dz_graph_launch(void* graph_exec, dz_stream_t stream);
int dz_graph_destroy(void* graph_exec);

/* Generic capture for the other learners (dz_dense_learn, dz_iqn_learn) and
 * any fixed sequence of this library's launches: everything enqueued on
 * `stream` (non-default) between begin and end becomes ONE executable graph.
 * dz_graph_capture_end must be called even if a call in between failed (it
 * then discards the partial graph and returns that the capture was aborted).
 * The reference's counterpart is jax.jit tracing `update` once and replaying
 * the compiled executable (dqn/agent.py:109-117).                           */
int dz_graph_capture_begin(dz_stream_t stream);
int dz_graph_capture_end(dz_stream_t stream, int discard, void** graph_exec_out);

/* Fills n noise blocks with f(x)=sign(x)sqrt|x|, x ~ truncated normal on
 * [-2,2] (ref: networks.py:142-144), from a counter-based generator keyed by
 * (seed, counter).  Distribution-equivalent to the reference, not bit-equal
 * to JAX's threefry stream.                                     */
int dz_noise_fill(float* noise, int64_t count, uint64_t seed, uint64_t counter,
                  dz_stream_t stream);

/* ------------------------------------------------------------------------- *
 *  Dense-head agents: DQN, double-Q, prioritized (double-Q + importance
 *  weights), C51 and QR-DQN.  Network = dqn_torso + linear(512) + ReLU +
 *  linear(num_outputs) (ref: networks.py:181-221, 295-363); the learner step is
 *  the agent's jitted `update` (ref: dqn/agent.py:85-117, double_q/agent.py:
 *  85-120, prioritized/agent.py:86-122, c51/agent.py:87-116, qrdqn/agent.py:
 *  88-119) with the optimizer of its run_atari.py.
 * ------------------------------------------------------------------------- */
#define DZ_LOSS_Q 0            /* rlax.q_learning, 2 applies                      */
#define DZ_LOSS_DOUBLE_Q 1     /* rlax.double_q_learning, 3 applies (+ weights)   */
#define DZ_LOSS_CATEGORICAL 2  /* rlax.categorical_q_learning (C51), 2 applies    */
#define DZ_LOSS_QUANTILE 3     /* rlax.quantile_q_learning (QR-DQN), 2 applies    */
#define DZ_OPT_RMSPROP 0       /* optax.rmsprop(lr, decay, eps, centered=True)    */
#define DZ_OPT_ADAM 1          /* optax.chain(clip_by_global_norm, adam)          */

typedef struct {
  int32_t num_outputs, shared_bias, batch, groups;
  int32_t fc1_ld, fc2_ld;
  int64_t param_count, param_count_ref;
  int64_t conv_w[3], conv_b[3];
  int64_t fc1_w, fc1_b;          /* [3136][fc1_ld] (512 used), [512]            */
  int64_t fc2_w, fc2_b;          /* [512][fc2_ld], [num_outputs] or [1] shared   */
  int64_t ws_count;
  int64_t ws_act1, ws_act2, ws_feat, ws_fc1_part, ws_h1, ws_fc2_part, ws_out;
  int64_t ws_dout, ws_dh1, ws_dfeat_part, ws_dfeat, ws_dact2, ws_dact1;
  int64_t ws_wgrad_part, ws_norm_part, ws_scalars, ws_zeros;
  int64_t ws_act_seams;          /* one-launch decision (dz_dense_act): zero at creation, owned by it */
} dz_dense_layout_t;

/* groups: 2 (Q / categorical / quantile) or 3 (double-Q).                     */
int dz_dense_layout(int num_outputs, int shared_bias, int batch, int groups,
                    dz_dense_layout_t* out);

typedef struct {
  int32_t loss, optimizer;
  int32_t num_actions, num_outputs, batch, shared_bias;
  int32_t num_atoms;         /* categorical: K; quantile: N; else 0             */
  float* online;
  const float* target;
  float* grad;
  float* opt_m;              /* rmsprop mu / adam m                             */
  float* opt_v;              /* rmsprop nu / adam v                             */
  int32_t* opt_count;        /* adam step count (device int32)                  */
  const uint8_t* s_tm1;
  const uint8_t* s_t;
  const int64_t* a_tm1;
  const double* r_t;
  const double* discount_t;
  const float* weights;      /* NULL = unweighted                               */
  const float* aux;          /* categorical: support[K]; quantile: tau[N]       */
  float* ws;
  float* losses;             /* [B]: td errors (Q, double-Q) or per-sample loss */
  float* priorities;         /* [B] |td| (prioritized/agent.py:202) or NULL     */
  float lr, decay_or_b1, b2, eps, max_norm;
  float grad_error_bound;    /* Q / double-Q: clip of the td gradient           */
  float huber;               /* quantile: kappa                                 */
  /* optional priority write-back inside the backward launch, as in
   * dz_rainbow_args_t (prioritized/agent.py:202-206): needs `priorities`.      */
  double* prio_node;
  int64_t prio_cap_pow2, prio_capacity;
  const int64_t* prio_ids;
  double prio_exponent;
  double* prio_max_seen;
  uint32_t* prio_status;
  /* Optional, as dz_rainbow_args_t::next_sample (full step only): the NEXT step's replay
   * sample + gather rides in this step's optimiser launch.  next_sample->args.node == NULL
   * selects the UNIFORM replay (TransitionReplay: positions -> ids -> rows; no tree,
   * `u_*_h`, `probs_out`, `weights*_out` unused).                                      */
  const dz_next_sample_t* next_sample;
  /* 0 (default): a call that runs the backward pass AND the RMSProp optimiser at batch <= 32
   * (narrow Q heads) does not write fc1's weight-gradient block (6.4 MB) into `grad`: the
   * optimiser forms each entry from the layer's input and output gradient when it updates
   * the weight (as dz_rainbow_args_t::keep_all_grads).  1: every block is materialised.    */
  int32_t keep_all_grads;
  int32_t pad2_;
} dz_dense_args_t;

int dz_dense_learn(const dz_dense_args_t* args, int phases, dz_stream_t stream);

/* One apply of a dense-head network: raw head outputs [batch][num_outputs]
 * into out (may be NULL) and, for Q heads (num_outputs == num_actions), the
 * q-values, greedy action and max.  ref: dqn/agent.py:121-131.               */
int dz_dense_apply(int num_actions, int num_outputs, int shared_bias, int batch,
                   const float* params, const uint8_t* states, float* ws,
                   float* out, float* q_values_out, int32_t* greedy_out,
                   float* vmax_out, dz_stream_t stream);

/* The actor's decision for ONE observation as ONE launch, any dense head (DQN / double-Q /
 * prioritized: num_outputs = A, dqn/agent.py:121-131; C51: 51 A, c51/agent.py:118-126;
 * QR-DQN: 201 A, qrdqn/agent.py:121-129).
 * `ws`: a dz_dense_layout(num_outputs, shared_bias, 1, 1) workspace, zero when created
 * and used by nothing else in between (its ws_act_seams region belongs to this call).
 * `pairs_out` (8-byte aligned; pinned device-mapped host memory or device memory):
 * num_outputs 8-byte words {float value, float 1.0f}, each written with ONE store -- a
 * host that zeroed the words before the call may poll them with plain loads instead of
 * waiting for the stream, and then forms q-values from the head outputs as the
 * reference's network does (softmax expectation / quantile mean / identity).
 * Liveness and failure: as dz_rainbow_act's one-launch form; a decision whose seams timed
 * out (or that found the sticky word set) delivers {NaN, DZ_ACT_FAILED_MARKER} words.  */
#define DZ_ACT_FAILED_MARKER 2.0f
int dz_dense_act(int num_outputs, int shared_bias, const float* params,
                 const uint8_t* state, float* ws, void* pairs_out, dz_stream_t stream);

/* ------------------------------------------------------------------------- *
 *  IQN learner step (ref: iqn/agent.py:176-232 loss_fn/update,
 *  networks.py:264-292 iqn_atari_network).  Three applies, each on
 *  batch x samples rows: online(s_tm1, tau_tm1) [gradient],
 *  target(s_t, tau_sel) [greedy-action selector], target(s_t, tau_t) [targets];
 *  the two target applies share one torso pass.  The tau draws are INPUTS
 *  (device arrays [batch][samples]); dz_uniform_fill produces them on the device.
 *  Parameter vector: conv1..3 (w,b), tau-embedding linear [latent][3136] + b,
 *  fc1 [3136][512] + b, fc2 [512][fc2_ld] + b  (haiku creation order).
 * ------------------------------------------------------------------------- */
typedef struct {
  int32_t num_actions, latent_dim, batch;
  int32_t samples[3];        /* tau_samples_s_tm1, _policy, _s_t                */
  int32_t emb_ld, fc1_ld, fc2_ld;
  int32_t pad_;
  int64_t conv_w[3], conv_b[3];
  int64_t emb_w, emb_b, fc1_w, fc1_b, fc2_w, fc2_b;
  int64_t param_count, param_count_ref;
  int64_t ws_count;
  int64_t ws_act1, ws_act2, ws_feat, ws_cos, ws_hin;
  int64_t ws_temb;           /* empty since round 5 (the embedding activation is not stored) */
  int64_t ws_h1, ws_out;
  int64_t ws_dout, ws_dh1, ws_dhin, ws_dfeat, ws_dact2, ws_dact1;
  int64_t ws_wgrad_part, ws_fc2w_part, ws_embw_part, ws_bias_part;
  int64_t ws_norm_part, ws_scalars, ws_zeros;
  int64_t ws_act_seams;      /* one-launch decision (dz_iqn_act): zero at creation, owned by it; the last region */
} dz_iqn_layout_t;

int dz_iqn_layout(int num_actions, int latent_dim, int batch, int samples_tm1,
                  int samples_sel, int samples_t, dz_iqn_layout_t* out);

typedef struct {
  int32_t num_actions, latent_dim, batch;
  int32_t samples[3];
  float* online;
  const float* target;
  float* grad;
  float* opt_m;
  float* opt_v;
  int32_t* opt_count;
  const uint8_t* s_tm1;
  const uint8_t* s_t;
  const int64_t* a_tm1;
  const double* r_t;
  const double* discount_t;
  const float* tau_tm1;      /* [batch][samples[0]]                             */
  const float* tau_sel;      /* [batch][samples[1]]                             */
  const float* tau_t;        /* [batch][samples[2]]                             */
  float* ws;
  float* losses;             /* [batch] per-sample quantile-regression loss     */
  float lr, b1, b2, eps;
  float max_norm;            /* <= 0: no clipping (iqn/run_atari.py:213-215);
                                a one-call step then computes no global norm:
                                ws_scalars[DZ_SC_GNORM] reads 0                  */
  float huber;               /* kappa                                           */
} dz_iqn_args_t;

int dz_iqn_learn(const dz_iqn_args_t* args, int phases, dz_stream_t stream);

/* One IQN apply on batch x samples rows: q_dist [batch][samples][num_actions]
 * (may be NULL), q_values = mean over samples, greedy action (first maximum)
 * and its value.  ref: iqn/agent.py:72-83 (actor), 234-247 (select_action).   */
int dz_iqn_apply(int num_actions, int latent_dim, int batch, int samples,
                 const float* params, const uint8_t* states, const float* taus,
                 float* ws, float* q_dist_out, float* q_values_out,
                 int32_t* greedy_out, float* vmax_out, dz_stream_t stream);

/* The IQN actor's decision for ONE observation as ONE launch (csrc/dz_iqn_act.h; ref:
 * iqn/agent.py:234-247 select_action): samples (<= 64: the reference default) fresh tau draws -- tau_j is the value
 * dz_uniform_fill(seed = tau_seed, counter = tau_counter + j) produces, drawn inside the kernel and
 * written to taus_out [samples] if given --, the network on those taus, q = mean over the taus.
 * num_actions <= 32, latent_dim <= 64.  `ws`: a dz_iqn_layout(num_actions, latent_dim, 1, samples,
 * 1, 1) workspace, zero when created and used by nothing else in between (its ws_act_seams region
 * belongs to this call).  `pairs_out` (8-byte aligned; pinned device-mapped host memory or device
 * memory): num_actions words {float q, float 1.0f}, each written with ONE store: a host that
 * zeroed them before the call may poll them.  Liveness and failure as dz_dense_act
 * ({NaN, DZ_ACT_FAILED_MARKER} words; zero the ws_act_seams region before the next decision).   */
int dz_iqn_act(int num_actions, int latent_dim, int samples, const float* params,
               const uint8_t* state, uint64_t tau_seed, uint64_t tau_counter, float* taus_out,
               float* ws, void* pairs_out, dz_stream_t stream);

/* out[i] = U[0,1) from the counter-based generator keyed by (seed, counter +
 * *step * n + i); `step` may be NULL.  The tau draws of iqn/agent.py:47-51.     */
int dz_uniform_fill(float* out, int64_t n, uint64_t seed, uint64_t counter,
                    const int32_t* step, dz_stream_t stream);

/* Optional per-kernel timing: when enabled, dz_rainbow_learn records a HIP
 * event on the launch stream before its first kernel and after every kernel.
 * dz_prof_read (call after synchronising the stream) returns the number of
 * marks of the LAST learn call and writes, per mark, the elapsed milliseconds
 * since the previous event and a NUL-terminated name of at most 31 chars
 * (names_out: max_marks * 32 bytes).  Used by bench.py for `roofline`.
 * dz_prof_enable(2): the marks take the calling thread's monotonic clock instead of
 * recording events -- dz_prof_read then returns what the ENQUEUE of each launch cost
 * the host (no synchronisation needed; tools/window_events.py).               */
int dz_prof_enable(int on);
int dz_prof_read(int max_marks, float* ms_out, char* names_out);
/* With timing enabled, dz_prioritized_sample (0), dz_replay_gather (1) and
 * dz_prioritized_update (2) record an event pair around their launch; this
 * returns the last elapsed milliseconds of each (-1 if never run) in ms_out[3].
 * Call after synchronising the stream.                                       */
int dz_prof_read_replay(float* ms_out);

/* ------------------------------------------------------------------------- *
 *  Atari observation preprocessing (next row f4)
 *  ref: dqn_zoo/processors.py:367-371 (rgb2y), 374-387 (resize: PIL BILINEAR),
 *       486-505 (observation branch of atari(): pool, grayscale, resize, stack)
 * ------------------------------------------------------------------------- */

/* One preprocessed observation in ONE launch:
 *   frame = resize(rgb2y(max(frames[0..n_frames))))  -> ring[slot]
 *   obs[y][x][j] = j-th oldest of the `count` newest ring frames, zeros after
 * frames[f]: device uint8 [height][width][channels] (host array of n_frames <= 4
 *   device pointers; n_frames = 0 pools nothing: a black frame); channels 3 = RGB (grayscaled with the reference's float64
 *   weights, un-fused, truncated, when grayscale != 0; with grayscale == 0 the three
 *   bands are kept and resampled independently, as PIL does for mode "RGB" --
 *   atari(grayscaling=False), processors.py:429,495: ring is then
 *   [stack][out_h][out_w][3] and obs [out_h][out_w][3][stack]) or 1 = already gray.
 * xbounds/xcoeffs, ybounds/ycoeffs: Pillow's 8-bit resample tables for each axis,
 *   device int32 [out][2] = (first input index, taps) and [out][ksize] 22-bit
 *   fixed-point weights (ksize <= 64); the horizontal pass runs first with a
 *   uint8 intermediate, exactly as PIL.Image.resize does.
 * ring: device uint8 [stack][out_h][out_w] frame stack (stack <= 8); the new
 *   frame is written to `slot`, `count` = frames held including it.
 * obs: device uint8 [out_h][out_w][stack] (np.stack(..., axis=-1) order).     */
int dz_atari_observation(const uint8_t* const* frames, int n_frames, int height, int width,
                         int channels, int grayscale, const int32_t* xbounds, const int32_t* xcoeffs,
                         int xksize, const int32_t* ybounds, const int32_t* ycoeffs,
                         int yksize, int out_h, int out_w, uint8_t* ring, int stack,
                         int slot, int count, uint8_t* obs, dz_stream_t stream);

/* STRUCTURAL LIMITS (each returns DZ_ERR_INVALID_ARG when exceeded):
 *  - categorical heads (Rainbow, C51): num_atoms <= 64 (one wavefront lane per
 *    atom in the loss kernel), num_actions <= 256; quantile heads: <= 256
 *    quantiles per action, num_actions <= 256;
 *  - the priority write-back carried inside a learner launch (`prio_*` fields of
 *    dz_rainbow_args_t / dz_dense_args_t): batch <= 256 (one workgroup walks all
 *    leaves); larger batches call dz_prioritized_update as its own launch;
 *  - the one-launch sample entry points (dz_replay_sample_uniform,
 *    dz_prioritized_sample_gather): batch <= 64 (the RNG draws travel in the
 *    kernel arguments); larger batches use dz_prioritized_sample + dz_replay_gather;
 *  - dz_rainbow_layout: batch <= 1024.
 * There are no run-time tuning knobs: launch shapes are compile-time constants
 * (csrc/dz_rainbow.hip, csrc/dz_qnet_kernels.h), chosen by measurement.       */

/* dst = src for a parameter buffer (target network sync,
 * ref: rainbow/agent.py:157-158).                                           */
int dz_param_copy(float* dst, const float* src, int64_t count, dz_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* DQNZOO_HIP_H_ */
