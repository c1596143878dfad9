"""ctypes binding of libdqnzoo_hip.so (the C ABI declared in include/dqnzoo_hip.h).

The product path has NO CPU fallback: if the shared library is missing or does
not export a symbol, importing/using it raises immediately (task rule: "the
product path must fail loudly when the HIP extension is missing").
"""

import ctypes
import os

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_PKG, 'libdqnzoo_hip.so')

DZ_OK = 0
DZ_ERR_INVALID_ARG = -1
DZ_ERR_HIP = -2
DZ_ERR_UNSUPPORTED = -3

ACT_FAILED = -2           # DZ_ACT_FAILED: the 'action' of a one-launch decision whose seams timed out
ACT_FAILED_MARKER = 2.0   # DZ_ACT_FAILED_MARKER (dz_dense_act)

ST_BAD_VALUE = 1
ST_BAD_TARGET = 2
ST_BAD_INDEX = 4
ST_ZERO_ROOT = 8
ST_NONFINITE_WEIGHT = 16
ST_CHAIN_TIMEOUT = 32

MAX_FIELDS = 8

c_vp = ctypes.c_void_p
c_i64 = ctypes.c_int64
c_int = ctypes.c_int
c_f64 = ctypes.c_double
c_f32 = ctypes.c_float


class FieldDesc(ctypes.Structure):
  _fields_ = [('src', c_vp), ('dst', c_vp), ('row_bytes', c_i64)]


class InsertField(ctypes.Structure):
  """dz_insert_field_t."""
  _fields_ = [('dst', c_vp), ('src_row', c_vp), ('row_bytes', c_i64),
              ('imm', ctypes.c_uint64)]


class PrioSampleArgs(ctypes.Structure):
  _fields_ = [
      ('node', c_vp), ('cap_pow2', c_i64), ('capacity', c_i64),
      ('size', c_i64), ('t', c_i64), ('pos', c_vp), ('u_target', c_vp),
      ('u_mix', c_vp), ('usp', c_f64), ('one_minus_usp', c_f64),
      ('usp_times_up', c_f64), ('uniform_prob', c_f64), ('beta', c_f64),
      ('normalize', c_int), ('compute_weights', c_int),
      ('assume_nonzero_root', c_int),
  ]


c_i32 = ctypes.c_int32


class RainbowLayout(ctypes.Structure):
  _fields_ = (
      [('num_actions', c_i32), ('num_atoms', c_i32), ('batch', c_i32),
       ('groups', c_i32), ('adv2_ld', c_i32), ('val2_ld', c_i32), ('fc1_ld', c_i32), ('pad0_', c_i32),
       ('param_count', c_i64), ('param_count_ref', c_i64),
       ('conv_w', c_i64 * 3), ('conv_b', c_i64 * 3)] +
      [(n, c_i64) for n in (
          'fc1_mu_w', 'fc1_mu_b', 'fc1_sig_w', 'fc1_sig_b', 'adv2_mu_w',
          'adv2_sig_w', 'val2_mu_w', 'val2_sig_w', 'fc2_sig_b', 'noise_stride',
          'n_adv1_in', 'n_val1_in', 'n_fc1_out', 'n_adv2_in', 'n_val2_in',
          'n_fc2_out', 'ws_count', 'ws_act1', 'ws_act2', 'ws_feat',
          'ws_fc1_part', 'ws_h1', 'ws_fc2_part', 'ws_fc2_out', 'ws_dout2',
          'ws_dh1', 'ws_dfeat_part', 'ws_dfeat', 'ws_dact2', 'ws_dact1',
          'ws_wgrad_part', 'ws_norm_part', 'ws_scalars', 'ws_q_sel',
          'ws_target_probs', 'ws_colsum_part', 'ws_act_seams')])


class RainbowArgs(ctypes.Structure):
  _fields_ = [
      ('num_actions', c_i32), ('num_atoms', c_i32), ('batch', c_i32),
      ('online', c_vp), ('target', c_vp), ('grad', c_vp), ('adam_m', c_vp),
      ('adam_v', c_vp), ('adam_count', c_vp), ('s_tm1', c_vp), ('s_t', c_vp),
      ('a_tm1', c_vp), ('r_t', c_vp), ('discount_t', c_vp), ('weights', c_vp),
      ('support', c_vp), ('noise', c_vp), ('ws', c_vp), ('losses', c_vp),
      ('priorities', c_vp), ('lr', c_f32), ('b1', c_f32), ('b2', c_f32),
      ('eps', c_f32), ('max_norm', c_f32), ('resample_noise', c_i32),
      ('noise_seed', ctypes.c_uint64),
      ('prio_node', c_vp), ('prio_cap_pow2', c_i64), ('prio_capacity', c_i64),
      ('prio_ids', c_vp), ('prio_exponent', c_f64), ('prio_max_seen', c_vp),
      ('prio_status', c_vp), ('keep_all_grads', c_i32), ('separate_launches', c_i32),
      ('next_sample', c_vp),
  ]


class NextSample(ctypes.Structure):
  """dz_next_sample_t: the arguments of dz_prioritized_sample_gather, carried by a
  learner step (RainbowArgs.next_sample)."""
  _fields_ = [
      ('args', PrioSampleArgs), ('pos_h', c_vp), ('u_target_h', c_vp), ('u_mix_h', c_vp),
      ('fields', c_vp), ('num_fields', c_i32), ('n', c_i32), ('ids_out', c_vp),
      ('probs_out', c_vp), ('weights_out', c_vp), ('weights32_out', c_vp),
      ('status', c_vp),
  ]


class DenseLayout(ctypes.Structure):
  _fields_ = (
      [('num_outputs', c_i32), ('shared_bias', c_i32), ('batch', c_i32),
       ('groups', c_i32), ('fc1_ld', c_i32), ('fc2_ld', c_i32),
       ('param_count', c_i64), ('param_count_ref', c_i64),
       ('conv_w', c_i64 * 3), ('conv_b', c_i64 * 3)] +
      [(n, c_i64) for n in (
          'fc1_w', 'fc1_b', 'fc2_w', 'fc2_b', 'ws_count', 'ws_act1', 'ws_act2',
          'ws_feat', 'ws_fc1_part', 'ws_h1', 'ws_fc2_part', 'ws_out', 'ws_dout',
          'ws_dh1', 'ws_dfeat_part', 'ws_dfeat', 'ws_dact2', 'ws_dact1',
          'ws_wgrad_part', 'ws_norm_part', 'ws_scalars', 'ws_zeros', 'ws_act_seams')])


class DenseArgs(ctypes.Structure):
  _fields_ = [
      ('loss', c_i32), ('optimizer', c_i32), ('num_actions', c_i32),
      ('num_outputs', c_i32), ('batch', c_i32), ('shared_bias', c_i32),
      ('num_atoms', c_i32), ('online', c_vp), ('target', c_vp), ('grad', c_vp),
      ('opt_m', c_vp), ('opt_v', c_vp), ('opt_count', c_vp), ('s_tm1', c_vp),
      ('s_t', c_vp), ('a_tm1', c_vp), ('r_t', c_vp), ('discount_t', c_vp),
      ('weights', c_vp), ('aux', c_vp), ('ws', c_vp), ('losses', c_vp),
      ('priorities', c_vp), ('lr', c_f32), ('decay_or_b1', c_f32), ('b2', c_f32),
      ('eps', c_f32), ('max_norm', c_f32), ('grad_error_bound', c_f32),
      ('huber', c_f32),
      ('prio_node', c_vp), ('prio_cap_pow2', c_i64), ('prio_capacity', c_i64),
      ('prio_ids', c_vp), ('prio_exponent', c_f64), ('prio_max_seen', c_vp),
      ('prio_status', c_vp), ('next_sample', c_vp),
      ('keep_all_grads', c_i32), ('pad2_', c_i32),
  ]


class IqnLayout(ctypes.Structure):
  """dz_iqn_layout_t."""
  _fields_ = [
      ('num_actions', c_i32), ('latent_dim', c_i32), ('batch', c_i32),
      ('samples', c_i32 * 3), ('emb_ld', c_i32), ('fc1_ld', c_i32),
      ('fc2_ld', c_i32), ('pad_', c_i32),
      ('conv_w', c_i64 * 3), ('conv_b', c_i64 * 3),
      ('emb_w', c_i64), ('emb_b', c_i64), ('fc1_w', c_i64), ('fc1_b', c_i64),
      ('fc2_w', c_i64), ('fc2_b', c_i64),
      ('param_count', c_i64), ('param_count_ref', c_i64), ('ws_count', c_i64),
      ('ws_act1', c_i64), ('ws_act2', c_i64), ('ws_feat', c_i64),
      ('ws_cos', c_i64), ('ws_hin', c_i64), ('ws_temb', c_i64),
      ('ws_h1', c_i64), ('ws_out', c_i64), ('ws_dout', c_i64),
      ('ws_dh1', c_i64), ('ws_dhin', c_i64), ('ws_dfeat', c_i64),
      ('ws_dact2', c_i64), ('ws_dact1', c_i64), ('ws_wgrad_part', c_i64),
      ('ws_fc2w_part', c_i64), ('ws_embw_part', c_i64),
      ('ws_bias_part', c_i64), ('ws_norm_part', c_i64), ('ws_scalars', c_i64),
      ('ws_zeros', c_i64), ('ws_act_seams', c_i64),
  ]


class IqnArgs(ctypes.Structure):
  """dz_iqn_args_t."""
  _fields_ = [
      ('num_actions', c_i32), ('latent_dim', c_i32), ('batch', c_i32),
      ('samples', c_i32 * 3),
      ('online', c_vp), ('target', c_vp), ('grad', c_vp), ('opt_m', c_vp),
      ('opt_v', c_vp), ('opt_count', c_vp), ('s_tm1', c_vp), ('s_t', c_vp),
      ('a_tm1', c_vp), ('r_t', c_vp), ('discount_t', c_vp), ('tau_tm1', c_vp),
      ('tau_sel', c_vp), ('tau_t', c_vp), ('ws', c_vp), ('losses', c_vp),
      ('lr', c_f32), ('b1', c_f32), ('b2', c_f32), ('eps', c_f32),
      ('max_norm', c_f32), ('huber', c_f32),
  ]


LOSS_Q, LOSS_DOUBLE_Q, LOSS_CATEGORICAL, LOSS_QUANTILE = 0, 1, 2, 3
OPT_RMSPROP, OPT_ADAM = 0, 1
SC_GNORM, SC_LOSS, SC_BC1, SC_BC2, SC_CLIP = 0, 1, 2, 3, 4
SC_CHAIN_FAIL = 6
PHASE_FORWARD, PHASE_BACKWARD, PHASE_OPTIMIZER, PHASE_ALL = 1, 2, 4, 7
PHASE_FWD_NETS, PHASE_FWD_LOSS = 8, 16   # the two halves of PHASE_FORWARD

class ReplayInsertArgs(ctypes.Structure):
  """dz_replay_insert_args_t."""
  _fields_ = [('fields', ctypes.POINTER(InsertField)), ('num_fields', c_i32), ('reserved', c_i32),
              ('t', c_i64), ('capacity', c_i64), ('node', c_vp), ('cap_pow2', c_i64),
              ('priority_h', c_f64), ('priority_d', c_vp), ('exponent', c_f64), ('status', c_vp)]


class RainbowActArgs(ctypes.Structure):
  """dz_rainbow_act_args_t."""
  _fields_ = [('num_actions', c_i32), ('num_atoms', c_i32), ('batch', c_i32), ('reserved', c_i32),
              ('params', c_vp), ('states', c_vp), ('noise', c_vp),
              ('noise_seed', ctypes.c_uint64), ('noise_counter', ctypes.c_uint64),
              ('step_counter', c_vp), ('support', c_vp), ('ws', c_vp), ('q_values_out', c_vp),
              ('greedy_out', c_vp), ('vmax_out', c_vp)]


STRUCT_IDS = {0: FieldDesc, 1: PrioSampleArgs, 2: RainbowLayout, 3: RainbowArgs,
              4: DenseLayout, 5: DenseArgs, 6: IqnLayout, 7: IqnArgs,
              8: InsertField, 9: NextSample, 10: ReplayInsertArgs, 11: RainbowActArgs}

# name -> (restype, argtypes).  tests/test_abi.py checks this table against
# the prototypes in include/dqnzoo_hip.h and against the built library.
SIGNATURES = {
    'dz_version': (ctypes.c_char_p, []),
    'dz_last_hip_error': (c_int, []),
    'dz_built_arch': (ctypes.c_char_p, []),
    'dz_struct_size': (c_int, [c_int]),
    'dz_rainbow_layout': (c_int, [c_int, c_int, c_int,
                                  ctypes.POINTER(RainbowLayout)]),
    'dz_rainbow_learn': (c_int, [ctypes.POINTER(RainbowArgs), c_int, c_vp]),
    'dz_rainbow_apply': (c_int, [c_int, c_int, c_int, c_vp, c_vp, c_vp, c_vp,
                                 c_vp, c_vp, c_vp, c_vp, c_vp]),
    'dz_rainbow_act': (c_int, [c_int, c_int, c_int, c_vp, c_vp, c_vp, ctypes.c_uint64,
                               ctypes.c_uint64, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
    'dz_rainbow_graph_capture': (c_int, [ctypes.POINTER(RainbowArgs), c_int, c_vp,
                                         ctypes.POINTER(c_vp)]),
    'dz_atari_observation': (c_int, [c_vp, c_int, c_int, c_int, c_int, c_int, c_vp, c_vp, c_int,
                                     c_vp, c_vp, c_int, c_int, c_int, c_vp, c_int, c_int,
                                     c_int, c_vp, c_vp]),
    'dz_graph_capture_begin': (c_int, [c_vp]),
    'dz_graph_capture_end': (c_int, [c_vp, c_int, ctypes.POINTER(c_vp)]),
    'dz_graph_launch': (c_int, [c_vp, c_vp]),
    'dz_graph_destroy': (c_int, [c_vp]),
    'dz_dense_layout': (c_int, [c_int, c_int, c_int, c_int,
                                ctypes.POINTER(DenseLayout)]),
    'dz_dense_learn': (c_int, [ctypes.POINTER(DenseArgs), c_int, c_vp]),
    'dz_dense_apply': (c_int, [c_int, c_int, c_int, c_int, c_vp, c_vp, c_vp, c_vp,
                               c_vp, c_vp, c_vp, c_vp]),
    'dz_dense_act': (c_int, [c_int, c_int, c_vp, c_vp, c_vp, c_vp, c_vp]),
    'dz_act_debug_spin_limit': (c_int, [c_int]),
    'dz_iqn_layout': (c_int, [c_int, c_int, c_int, c_int, c_int, c_int,
                              ctypes.POINTER(IqnLayout)]),
    'dz_iqn_learn': (c_int, [ctypes.POINTER(IqnArgs), c_int, c_vp]),
    'dz_iqn_apply': (c_int, [c_int, c_int, c_int, c_int, c_vp, c_vp, c_vp, c_vp,
                             c_vp, c_vp, c_vp, c_vp, c_vp]),
    'dz_iqn_act': (c_int, [c_int, c_int, c_int, c_vp, c_vp, ctypes.c_uint64, ctypes.c_uint64, c_vp,
                           c_vp, c_vp, c_vp]),
    'dz_uniform_fill': (c_int, [c_vp, c_i64, ctypes.c_uint64, ctypes.c_uint64,
                                c_vp, c_vp]),
    'dz_noise_fill': (c_int, [c_vp, c_i64, ctypes.c_uint64, ctypes.c_uint64,
                              c_vp]),
    'dz_param_copy': (c_int, [c_vp, c_vp, c_i64, c_vp]),
    'dz_prof_enable': (c_int, [c_int]),
    'dz_prof_read': (c_int, [c_int, c_vp, c_vp]),
    'dz_prof_read_replay': (c_int, [c_vp]),
    'dz_replay_gather': (c_int, [ctypes.POINTER(FieldDesc), c_int, c_vp, c_int,
                                 c_i64, c_vp]),
    'dz_replay_sample_uniform': (c_int, [ctypes.POINTER(FieldDesc), c_int, c_vp, c_int,
                                         c_i64, c_i64, c_i64, c_vp, c_vp]),
    'dz_replay_insert': (c_int, [ctypes.POINTER(InsertField), c_int, c_i64, c_i64,
                                 c_vp, c_i64, c_f64, c_vp, c_f64, c_vp, c_vp]),
    'dz_replay_insert_v': (c_int, [ctypes.POINTER(ReplayInsertArgs), c_vp]),
    'dz_sample_gather_desc': (c_int, [ctypes.POINTER(NextSample), c_vp]),
    'dz_rainbow_act_v': (c_int, [ctypes.POINTER(RainbowActArgs), c_vp]),
    'dz_uniform_pos_to_id': (c_int, [c_vp, c_int, c_i64, c_i64, c_i64, c_vp,
                                     c_vp]),
    'dz_sumtree_set': (c_int, [c_vp, c_i64, c_i64, c_vp, c_vp, c_int, c_vp,
                               c_vp]),
    'dz_sumtree_get': (c_int, [c_vp, c_i64, c_i64, c_vp, c_int, c_vp, c_vp,
                               c_vp]),
    'dz_sumtree_rebuild': (c_int, [c_vp, c_i64, c_i64, c_vp]),
    'dz_sumtree_query': (c_int, [c_vp, c_i64, c_vp, c_int, c_vp, c_vp, c_vp]),
    'dz_prioritized_sample': (c_int, [ctypes.POINTER(PrioSampleArgs), c_int,
                                      c_vp, c_vp, c_vp, c_vp, c_vp, c_vp,
                                      c_vp]),
    'dz_prioritized_sample_host_draws': (c_int, [ctypes.POINTER(PrioSampleArgs), c_int,
                                                 c_vp, c_vp, c_vp, c_vp, c_vp, c_vp,
                                                 c_vp, c_vp, c_vp, c_vp]),
    'dz_prioritized_sample_gather': (c_int, [ctypes.POINTER(PrioSampleArgs), c_int, c_vp, c_vp,
                                             c_vp, ctypes.POINTER(FieldDesc), c_int, c_vp,
                                             c_vp, c_vp, c_vp, c_vp, c_vp]),
    'dz_prioritized_update': (c_int, [c_vp, c_i64, c_i64, c_i64, c_i64, c_vp,
                                      c_vp, c_int, c_f64, c_int, c_vp, c_vp,
                                      c_vp]),
    'dz_prioritized_add': (c_int, [c_vp, c_i64, c_i64, c_i64, c_int, c_f64,
                                   c_vp, c_f64, c_vp, c_vp]),
}


class HipLibraryError(RuntimeError):
  pass


_lib = None


def load():
  """Loads the library once; raises HipLibraryError if it is unusable."""
  global _lib
  if _lib is not None:
    return _lib
  if not os.path.exists(LIB_PATH):
    raise HipLibraryError(
        'libdqnzoo_hip.so is not built (%s). Run `python -m dqn_zoo_amd.build` '
        '(hipcc, --offload-arch=gfx950). There is no CPU fallback.' % LIB_PATH)
  # PyTorch-ROCm bundles its own libamdhip64.so; it must be mapped BEFORE this
  # library so that both resolve to ONE HIP runtime instance (streams and
  # device pointers are shared between them).
  import torch  # noqa: F401  pylint: disable=unused-import,import-outside-toplevel
  try:
    lib = ctypes.CDLL(LIB_PATH)
  except OSError as e:
    raise HipLibraryError('cannot load %s: %s' % (LIB_PATH, e)) from e
  for name, (res, args) in SIGNATURES.items():
    try:
      fn = getattr(lib, name)
    except AttributeError as e:
      raise HipLibraryError('%s does not export %s' % (LIB_PATH, name)) from e
    fn.restype = res
    fn.argtypes = args
  for which, cls in STRUCT_IDS.items():
    if lib.dz_struct_size(which) != ctypes.sizeof(cls):
      raise HipLibraryError(
          'ABI mismatch: struct %s is %d bytes in the library, %d in _lib.py' %
          (cls.__name__, lib.dz_struct_size(which), ctypes.sizeof(cls)))
  _check_single_hip_runtime()
  _lib = lib
  return lib


def _check_single_hip_runtime():
  try:
    with open('/proc/self/maps') as f:
      paths = {line.split()[-1] for line in f if 'libamdhip64' in line}
  except OSError:
    return
  real = {os.path.realpath(p) for p in paths}
  if len(real) > 1:
    raise HipLibraryError(
        'two HIP runtimes are mapped in this process (%s): import torch before '
        'loading libdqnzoo_hip.so' % sorted(real))


def check(code, what):
  """Maps a DZ_ERR_* return code to a Python exception."""
  if code == DZ_OK:
    return
  if code == DZ_ERR_INVALID_ARG:
    raise ValueError('%s: invalid argument' % what)
  if code == DZ_ERR_HIP:
    raise HipLibraryError('%s: HIP error %d' % (what, load().dz_last_hip_error()))
  if code == DZ_ERR_UNSUPPORTED:
    raise NotImplementedError('%s: unsupported configuration' % what)
  raise HipLibraryError('%s: unknown error code %d' % (what, code))


def capture_graph(stream, enqueue):
  """Runs `enqueue()` (calls into this library on `stream`) under hipGraph
  capture and returns the executable graph handle."""
  lib = load()
  if not stream:
    raise RuntimeError('hipGraph capture needs a non-default stream: run under '
                       '`torch.cuda.stream(torch.cuda.Stream())`')
  # (refused -- ValueError -- while the event profiler is on: its marks are
  # event records between the kernels and cannot live inside a graph)
  check(lib.dz_graph_capture_begin(stream), 'dz_graph_capture_begin')
  h = ctypes.c_void_p()
  try:
    enqueue()
  except BaseException:
    lib.dz_graph_capture_end(stream, 1, ctypes.byref(h))
    raise
  check(lib.dz_graph_capture_end(stream, 0, ctypes.byref(h)), 'dz_graph_capture_end')
  return h


def ptr(t):
  """Device pointer of a torch tensor (or 0 for None)."""
  if t is None:
    return None
  return t.data_ptr()


def current_stream_ptr():
  import torch
  return torch.cuda.current_stream().cuda_stream


# torch.cuda.current_stream() builds a Stream object through three Python layers
# (~4 us; the agent loop asked for it 3-4 times per frame).  The raw handle of torch's
# CURRENT stream comes from one C call; Stream objects (needed by Event.record) are
# cached per raw handle.
_stream_objects = {}


def _device_index(device):
  import torch
  idx = getattr(device, 'index', device)
  return torch.cuda.current_device() if idx is None else idx


def stream_ptr(device=None):
  """hipStream_t of torch's current stream on `device`, as an integer."""
  import torch
  raw = getattr(torch._C, '_cuda_getCurrentRawStream', None)
  if raw is None:
    return torch.cuda.current_stream(device).cuda_stream
  return raw(_device_index(device))


def current_stream(device=None):
  """torch's current stream on `device` as a (cached) torch.cuda.Stream object."""
  import torch
  idx = _device_index(device)
  key = (idx, stream_ptr(idx))
  s = _stream_objects.get(key)
  if s is None:
    s = torch.cuda.current_stream(idx)
    _stream_objects[key] = s
  return s
