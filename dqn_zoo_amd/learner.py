"""On-device Rainbow learner: the reference's jitted `update`
(ref: rainbow/agent.py:85-123) as one C-ABI call that enqueues every kernel of
the step (3 network applies, categorical double-Q loss, backward,
clip_by_global_norm + Adam) on the current HIP stream.

All state (online/target parameters, Adam moments, step count, workspace,
noise) lives in PyTorch-ROCm tensors owned by this object and is handed to the
library as raw device pointers; nothing is copied to the host unless asked.
"""

import ctypes
import time
import typing

import numpy as np
import torch

from dqn_zoo_amd import _lib
from dqn_zoo_amd import networks


class ActDecisionError(RuntimeError):
  """A one-launch decision (dz_rainbow_act batch 1 / dz_dense_act) gave up on one of its
  in-kernel seams (include/dqnzoo_hip.h: DZ_ACT_FAILED).  The workspace has been re-armed when
  this is raised: the next decision starts from a clean seam area."""


def read_packed_action(greedy: torch.Tensor, vmax: torch.Tensor):
  """Row 0 of (greedy int32 [B], vmax float32 [B]) that are views of ONE
  [2, B] int32 buffer: one 8*B-byte device->host copy instead of two syncs."""
  base = greedy._base if greedy._base is not None else None  # pylint: disable=protected-access
  if base is None or base.dim() != 2 or vmax.data_ptr() != base[1].data_ptr():
    a, v = int(greedy[0].item()), float(vmax[0].item())
  else:
    h = base.cpu()
    a, v = int(h[0, 0]), float(h[1].view(torch.float32)[0])
  if a == _lib.ACT_FAILED:
    raise ActDecisionError('the one-launch decision gave up on a seam (DZ_ACT_FAILED): zero the '
                           'ws_act_seams region of the acting workspace before the next decision')
  return a, v


def check_batch(b, s_tm1, a_tm1, r_t, discount_t, s_t, weights=None):
  """Validates a device batch (survives `python -O`, unlike assert): the
  kernels reinterpret raw pointers, so a wrong dtype would be silent garbage."""
  for name, x, dt, shape in (
      ('s_tm1', s_tm1, torch.uint8, (b, 84, 84, 4)),
      ('s_t', s_t, torch.uint8, (b, 84, 84, 4)),
      ('a_tm1', a_tm1, torch.int64, (b,)), ('r_t', r_t, torch.float64, (b,)),
      ('discount_t', discount_t, torch.float64, (b,)),
      ('weights', weights, torch.float32, (b,))):
    if x is None and name == 'weights':
      continue
    if not isinstance(x, torch.Tensor) or x.dtype != dt:
      raise TypeError('%s must be a %s device tensor, got %s' % (
          name, dt, getattr(x, 'dtype', type(x))))
    if tuple(x.shape) != shape or not x.is_contiguous():
      raise ValueError('%s must be contiguous with shape %s, got %s' % (
          name, shape, tuple(x.shape)))


class AdamConfig(typing.NamedTuple):
  """optax.chain(clip_by_global_norm(max_norm), adam(lr, eps=eps))
  (ref: rainbow/run_atari.py:77-81, 229-235).  max_norm <= 0 disables the clip."""
  learning_rate: float = 0.00025 / 4
  eps: float = 0.005 / 32
  b1: float = 0.9
  b2: float = 0.999
  max_global_grad_norm: float = 10.0


class RainbowLearner:

  MAX_AUTO_GRAPHS = 32   # distinct call signatures the automatic graph mode will capture

  def __init__(self, network: networks.RainbowNetwork, optimizer: AdamConfig,
               batch_size: int, seed: int = 1, device=None, params=None):
    self._lib = _lib.load()
    if not torch.cuda.is_available():
      raise _lib.HipLibraryError('RainbowLearner needs an AMD GPU; no CPU fallback')
    self.device = torch.device('cuda', torch.cuda.current_device()) \
        if device is None else torch.device(device)
    self.network = network
    self.opt = optimizer
    self.batch_size = int(batch_size)
    self.layout = network.layout(self.batch_size)
    L = self.layout
    if params is None:
      params = network.init(np.random.RandomState(seed))
    f32 = dict(dtype=torch.float32, device=self.device)
    self.online = torch.from_numpy(L.pack(params)).to(self.device)
    self.target = self.online.clone()  # rainbow/agent.py:72
    self.grad = torch.zeros(L.param_count, **f32)
    self.adam_m = torch.zeros(L.param_count, **f32)
    self.adam_v = torch.zeros(L.param_count, **f32)
    self.adam_count = torch.zeros(1, dtype=torch.int32, device=self.device)
    self.ws = torch.zeros(L.ws_count, **f32)
    self.noise = torch.zeros(3 * L.noise_stride, **f32)
    self.losses = torch.zeros(self.batch_size, **f32)
    self.priorities = torch.zeros(self.batch_size, **f32)
    self.support = torch.from_numpy(network.support).to(self.device)
    self._noise_seed = int(seed) & 0xFFFFFFFFFFFFFFFF
    self._noise_counter = 0
    self._args = None
    self._graphs = {}        # (input pointers, phases, noise flag) -> hipGraphExec
    # replay each distinct call signature from a hipGraph: True / False, or None =
    # automatically whenever the current stream allows capture (any non-default stream)
    self.use_graphs = None
    # False: a one-call step (loss + backward + optimiser, batch <= 32) stores NEITHER of
    # fc1's weight-gradient blocks -- the optimiser forms their entries from the layer's
    # input and output gradient (csrc/dz_fc1_onfly.h) -- and a split step stores only the
    # mu block; True: `grad` holds every block (inspection, tests, A/B)
    self.keep_all_grads = False
    # True: one launch per stage of the head chain (fc1 epilogue, fc2, loss, fc2 backward) instead
    # of the default multi-role launch of the one-call step (csrc/dz_head_chain.h): A/B and the
    # bit-identity tests
    self.separate_launches = False
    # inference (acting) side: own workspace + one noise block, so that an
    # apply never aliases the buffers of an enqueued learner step.
    self._act_batch = 0
    self._act_ws = None
    self._act_noise = torch.zeros(L.noise_stride, **f32)
    self.act_graphs = True   # replay the acting apply from a hipGraph (apply_async)
    self._act_graphs = {}    # (state buffer, result slot, parameters) -> (graph, q, greedy, vmax)
    self._act_fast = {}      # apply_async's steady state: -> (pre-bound enqueue, slot reader)

  # -- state ------------------------------------------------------------------
  def get_params(self, which='online') -> dict:
    t = self.online if which == 'online' else self.target
    return self.layout.unpack(t.cpu().numpy())

  def set_params(self, params: dict, which='online') -> None:
    t = self.online if which == 'online' else self.target
    t.copy_(torch.from_numpy(self.layout.pack(params)))

  def sync_target(self) -> None:
    """target <- online (ref: rainbow/agent.py:157-158)."""
    _lib.check(self._lib.dz_param_copy(
        self.target.data_ptr(), self.online.data_ptr(), self.layout.param_count,
        _lib.stream_ptr(self.device)), 'dz_param_copy')

  def set_noise(self, noises: typing.Sequence[dict]) -> None:
    """Explicit noise for the 3 applies (parity runs)."""
    blocks = [self.layout.pack_noise(n) for n in noises]
    self.noise.copy_(torch.from_numpy(np.concatenate(blocks)))

  def set_noise_state(self, seed: int, counter: int, act_step: int = 0) -> None:
    """Restores the noise stream positions (agent `set_state`).  The cached
    argument block and every captured hipGraph bake the old seed in: drop them."""
    self._noise_seed = int(seed) & 0xFFFFFFFFFFFFFFFF
    self._noise_counter = int(counter)
    self._args = None
    self.drop_graphs()
    if getattr(self, '_act_step', None) is None:
      self._act_step = torch.zeros(1, dtype=torch.int32, device=self.device)
    self._act_step.fill_(int(act_step))

  def act_step(self) -> int:
    """Number of acting applies drawn so far (the actor's noise stream position)."""
    return 0 if getattr(self, '_act_step', None) is None else int(self._act_step.item())

  MAX_ACT_GRAPHS = 64    # acting graphs: ACT_RING result slots x observation slots

  def _drop_act_graphs(self) -> None:
    graphs, self._act_graphs = getattr(self, '_act_graphs', {}), {}
    self._act_fast = {}
    for g in graphs.values():
      if g[0] is not None:
        self._lib.dz_graph_destroy(g[0])

  def drop_graphs(self) -> None:
    graphs, self._graphs = self._graphs, {}
    for g in graphs.values():
      self._lib.dz_graph_destroy(g)
    self._drop_act_graphs()

  def resample_noise(self) -> None:
    """Fresh factorised noise for the 3 applies, generated on the device."""
    n = self.noise.numel()
    _lib.check(self._lib.dz_noise_fill(
        self.noise.data_ptr(), n, self._noise_seed, self._noise_counter,
        _lib.stream_ptr(self.device)), 'dz_noise_fill')
    self._noise_counter += n

  def apply(self, states: torch.Tensor, which: str = 'online', noise=None,
            resample_noise: bool = True, packed_out=None):
    """One network apply on uint8 states [B,84,84,4] (device tensor).
    Returns device tensors (q_values [B,A] f32, greedy action [B] i32,
    max_a q [B] f32).  ref: rainbow/agent.py:125-131 (select_action)."""
    stream = _lib.stream_ptr(self.device)
    params = self.online if which == 'online' else self.target
    if (packed_out is not None and noise is None and resample_noise and self.act_graphs and
        stream):
      # steady state of the agent loop: same observation slot, same result slot, same
      # parameters -> replay the captured launches, nothing to allocate
      # (the key carries shape and dtype: an address the allocator recycled for a
      # different tensor must not replay a graph captured for the old one)
      g = self._act_graphs.get((states.data_ptr(), packed_out.data_ptr(), params.data_ptr(),
                                states.shape, states.dtype))
      if g is not None:
        if g[0] is None:
          g[4](stream)   # one launch: enqueued directly (a graph of one node costs more to launch)
        else:
          _lib.check(self._lib.dz_graph_launch(g[0], stream), 'dz_graph_launch')
        return g[1], g[2], g[3]
    assert states.dtype == torch.uint8 and states.is_contiguous()
    b = int(states.shape[0])
    assert tuple(states.shape[1:]) == (84, 84, 4)
    if self._act_batch != b:
      self._drop_act_graphs()  # captured against the old workspace
      self._act_ws = torch.zeros(self.network.layout(b).ws_count,
                                 dtype=torch.float32, device=self.device)
      self._act_batch = b
    a = self.network.num_actions
    q = torch.empty((b, a), dtype=torch.float32, device=self.device)
    # (greedy action, max q) packed in one 8-byte buffer per row so that the
    # actor's device->host read is ONE copy (read_action)
    # (`packed_out`: a pinned host int32 [2, b] tensor the kernel writes through its
    # device mapping instead, see apply_async)
    packed = torch.empty((2, b), dtype=torch.int32, device=self.device) \
        if packed_out is None else packed_out
    greedy, vmax = packed[0], packed[1].view(torch.float32)
    if noise is None and resample_noise:
      # fresh noise drawn inside the apply's own launches (dz_rainbow_act); the
      # stream position is a DEVICE counter the apply advances itself, so the
      # argument list is constant and the 8 launches replay from a hipGraph
      if getattr(self, '_act_step', None) is None:
        self._act_step = torch.zeros(1, dtype=torch.int32, device=self.device)
      enqueue = lambda: _lib.check(self._lib.dz_rainbow_act(
          a, self.network.num_atoms, b, params.data_ptr(), states.data_ptr(),
          self._act_noise.data_ptr(), self._noise_seed ^ 0xA5A5A5A5, 0,
          self._act_step.data_ptr(), self.support.data_ptr(), self._act_ws.data_ptr(),
          q.data_ptr(), greedy.data_ptr(), vmax.data_ptr(), stream), 'dz_rainbow_act')
      if (self.act_graphs and stream and packed_out is not None and
          len(self._act_graphs) < self.MAX_ACT_GRAPHS):
        # (a caller that hands over a fresh observation tensor per decision --
        # atari(device_observations=True) -- fills the cache with single-use graphs:
        # beyond the cap the apply launches eagerly, as the learn step does)
        key = (states.data_ptr(), packed_out.data_ptr(), params.data_ptr(),
               states.shape, states.dtype)
        q = self._act_q = torch.empty((b, a), dtype=torch.float32, device=self.device)
        if b == 1 and self.act_direct:
          # (the arguments in a struct filled once: the per-frame call marshals two)
          fn, chk = self._lib.dz_rainbow_act_v, _lib.check
          av = _lib.RainbowActArgs()
          (av.num_actions, av.num_atoms, av.batch, av.params, av.states, av.noise, av.noise_seed,
           av.noise_counter, av.step_counter, av.support, av.ws, av.q_values_out, av.greedy_out,
           av.vmax_out) = (
               a, self.network.num_atoms, b, params.data_ptr(), states.data_ptr(),
               self._act_noise.data_ptr(), self._noise_seed ^ 0xA5A5A5A5, 0,
               self._act_step.data_ptr(), self.support.data_ptr(), self._act_ws.data_ptr(),
               q.data_ptr(), greedy.data_ptr(), vmax.data_ptr())
          ref = ctypes.byref(av)

          def direct(st, fn=fn, ref=ref, av=av):   # (`av` kept alive by the closure)
            rc = fn(ref, st)
            if rc:
              chk(rc, 'dz_rainbow_act')
          self._act_graphs[key] = (None, q, greedy, vmax, direct)   # the same cache, no graph
          direct(stream)
          return q, greedy, vmax
        g = self._act_graphs[key] = (_lib.capture_graph(stream, enqueue), q, greedy, vmax)
        _lib.check(self._lib.dz_graph_launch(g[0], stream), 'dz_graph_launch')
        return q, greedy, vmax
      enqueue()
      return q, greedy, vmax
    if noise is not None:
      self._act_noise.copy_(torch.from_numpy(self.layout.pack_noise(noise)))
    _lib.check(self._lib.dz_rainbow_apply(
        a, self.network.num_atoms, b, params.data_ptr(), states.data_ptr(),
        self._act_noise.data_ptr(), self.support.data_ptr(),
        self._act_ws.data_ptr(), q.data_ptr(), greedy.data_ptr(),
        vmax.data_ptr(), stream), 'dz_rainbow_apply')
    return q, greedy, vmax

  ACT_RING = 8   # acting results in flight (pinned host words)
  ACT_POLL_SECONDS = 2.0   # polled reads of the slot this long before falling back to a stream sync
  last_act_fail = 0        # the sticky word found by the last reset
  # (properties: toggling either drops the cached pre-bound enqueues and readers built for the
  # other setting)
  _poll_action_slot = True
  _act_direct = True   # the one-launch decision (batch 1) is enqueued directly, not from a hipGraph

  @property
  def poll_action_slot(self):
    return self._poll_action_slot

  @poll_action_slot.setter
  def poll_action_slot(self, v):
    if bool(v) != self._poll_action_slot:
      self._poll_action_slot = bool(v)
      self._act_fast = {}

  @property
  def act_direct(self):
    return self._act_direct

  @act_direct.setter
  def act_direct(self, v):
    if bool(v) != self._act_direct:
      self._act_direct = bool(v)
      self._drop_act_graphs()

  def act_seam_words(self):
    """(generation, tickets, sticky failure) of the one-launch decision's seam area (synchronises)."""
    if self._act_ws is None:
      return (0, 0, 0)
    off = int(self.network.layout(self._act_batch).c.ws_act_seams)
    w = self._act_ws[off:off + 64 * 8:64].view(torch.int32).tolist()
    return (w[3], w[4], w[5])

  def _reset_act_seams(self) -> None:
    """After a failed decision: waits for the device, records the sticky word and returns the
    seam area (generation, tickets, sticky word, both sets of intermediates) to all-zero bits --
    the state of a fresh workspace, from which the next decision is correct."""
    torch.cuda.synchronize(self.device)
    if self._act_ws is None:
      return
    lay = self.network.layout(self._act_batch).c
    off = int(lay.ws_act_seams)
    self.last_act_fail = int(self._act_ws[off + 5 * 64:off + 5 * 64 + 1].view(torch.int32).item())
    self._act_ws[off:int(lay.ws_count)].zero_()
    torch.cuda.synchronize(self.device)

  def apply_async(self, states: torch.Tensor):
    """Acting apply whose (greedy action, max q) pair is written by the kernel
    straight into pinned, device-mapped HOST memory: no device->host copy is
    enqueued and nothing synchronises here.  Returns `read() -> (action, value)`
    which waits for THESE launches only (an event recorded right behind them), so
    a learner step enqueued afterwards does not delay the action."""
    sp = _lib.stream_ptr(self.device)
    if self._act_fast:
      # steady state of an agent loop: the same observation slot, result slot and parameters as
      # some earlier frame -> mark the slot, the pre-bound two-argument enqueue, the slot's reader
      k = self._act_pos % self.ACT_RING
      fast = self._act_fast.get((states.data_ptr(), k, self.online.data_ptr(), states.numel(), sp))
      if fast is not None:
        self._act_pos += 1
        fast[2][0, 0] = -1
        fast[0](sp)
        return fast[1]
    b = int(states.shape[0])
    if getattr(self, '_act_host', None) is None or self._act_host.shape[2] != b:
      self._act_host = torch.empty((self.ACT_RING, 2, b), dtype=torch.int32).pin_memory()
      self._act_host_np = self._act_host.numpy()   # the same pinned words, for the polled read
      self._act_events = [torch.cuda.Event() for _ in range(self.ACT_RING)]
      self._act_pos = 0
      self._act_fast = {}   # (its entries point into the previous result slots)
    k = self._act_pos % self.ACT_RING
    self._act_pos += 1
    slot = self._act_host[k]
    lay = self.layout.c
    # (the kernels that end in dz_q_from_row store the pair as one 8-byte word: dz_rainbow_act at
    # batch <= 8 with a fc2 row of <= 1024 floats and <= 64 atoms -- every Atari action set)
    if (b == 1 and self.poll_action_slot and int(lay.adv2_ld) + int(lay.val2_ld) <= 1024 and
        self.network.num_atoms <= 64):
      # One observation: the decision kernel's last store is the (action, value) pair as ONE
      # 8-byte word into this pinned slot.  The host marks the slot (action -1) before the
      # enqueue and reads it with plain loads until the pair is there: no event to record, no
      # completion signal to wait for (the wake-up after `Event.synchronize` cost more than the
      # kernel's last phase).
      words = self._act_host_np[k]
      words[0, 0] = -1
      fk = (states.data_ptr(), k, self.online.data_ptr(), states.numel(), sp)
      self.apply(states, packed_out=slot)
      device = self.device
      vals = words[1].view(np.float32)

      enq_stream = _lib.current_stream(device)   # the stream THIS decision was enqueued on
      owner = self

      def read_polled():
        # plain loads of the pinned word; bounded by time (the GIL is held while spinning)
        deadline = None
        while True:
          for _ in range(2000):
            a = words[0, 0]
            if a != -1:
              break
          if a != -1:
            break
          now = time.monotonic()
          if deadline is None:
            deadline = now + owner.ACT_POLL_SECONDS
          elif now > deadline:
            enq_stream.synchronize()   # stuck or very slow: the stream the kernel is on decides
            a = words[0, 0]
            break
          else:
            time.sleep(0)   # a decision this late (> 2 000 looks) is behind other work: let other threads run
        if a < 0:
          # DZ_ACT_FAILED (a seam of the decision kernel timed out: sticky word set) or the
          # slot was never written: never hand the caller an action
          owner._reset_act_seams()   # pylint: disable=protected-access
          raise ActDecisionError(
              'dz_rainbow_act: the one-launch decision did not complete (slot word %d, sticky '
              'failure word %d); the acting workspace was re-armed' % (int(a), owner.last_act_fail))
        return int(a), float(vals[0])

      g = self._act_graphs.get((states.data_ptr(), slot.data_ptr(), self.online.data_ptr(),
                                states.shape, states.dtype))
      if g is not None and g[0] is None and states.dtype == torch.uint8:
        self._act_fast[fk] = (g[4], read_polled, words)
      return read_polled
    self.apply(states, packed_out=slot)
    ev = self._act_events[k]
    ev.record(_lib.current_stream(self.device))
    owner = self

    def read():
      ev.synchronize()
      a = int(slot[0, 0])
      if a == _lib.ACT_FAILED:   # (batch 1 is the one-launch decision whatever reads the slot)
        owner._reset_act_seams()   # pylint: disable=protected-access
        raise ActDecisionError('dz_rainbow_act: a seam of the one-launch decision timed out (sticky '
                               'failure word %d); the acting workspace was re-armed' % owner.last_act_fail)
      return a, float(slot[1].view(torch.float32)[0])

    return read

  @staticmethod
  def read_action(greedy: torch.Tensor, vmax: torch.Tensor):
    """(int action, float value) of row 0 with a single device->host copy;
    `greedy`/`vmax` are the tensors `apply` returned (views of one buffer)."""
    return read_packed_action(greedy, vmax)

  def get_opt_state(self) -> dict:
    return dict(count=int(self.adam_count.item()),
                mu=self.layout.unpack(self.adam_m.cpu().numpy()),
                nu=self.layout.unpack(self.adam_v.cpu().numpy()))

  def set_opt_state(self, state: dict) -> None:
    self.adam_count.fill_(int(state['count']))
    self.adam_m.copy_(torch.from_numpy(self.layout.pack(state['mu'])))
    self.adam_v.copy_(torch.from_numpy(self.layout.pack(state['nu'])))

  def scalars(self) -> dict:
    """Synchronises and returns the step's scalars (gnorm, loss, ...)."""
    off = int(self.layout.c.ws_scalars)
    sc = self.ws[off:off + 8].cpu().numpy()
    return dict(gnorm=float(sc[_lib.SC_GNORM]), loss=float(sc[_lib.SC_LOSS]),
                bc1=float(sc[_lib.SC_BC1]), bc2=float(sc[_lib.SC_BC2]),
                unclipped=bool(sc[_lib.SC_CLIP] != 0),
                chain_failed=bool(sc[_lib.SC_CHAIN_FAIL:_lib.SC_CHAIN_FAIL + 1].view(np.uint32)[0]))

  def check_status(self, fallback: bool = True) -> None:
    """Synchronises; raises `replay.ChainTimeoutError` if a multi-role launch of an earlier step
    gave up on one of its in-launch seams (ws_scalars[DZ_SC_CHAIN_FAIL], sticky: cleared here).
    Such a step is VOID -- its losses are NaN and its finalize / optimiser / priority write-back
    launches changed nothing (they read the word) -- and every step enqueued behind it was void
    too until this call clears the word.  `fallback`: later steps use the four-launch form of the
    head (`separate_launches`), whose liveness needs no dispatch-order assumption."""
    off = int(self.layout.c.ws_scalars) + _lib.SC_CHAIN_FAIL
    if int(self.ws[off:off + 1].view(torch.int32).item()) != 0:
      self.ws[off:off + 1].zero_()
      if fallback and not self.separate_launches:
        self.separate_launches = True
        self.drop_graphs()   # (captured steps bake the launch form in)
      from dqn_zoo_amd import replay as _replay
      raise _replay.ChainTimeoutError(
          'a multi-role learner launch timed out on an in-launch seam (DZ_SC_CHAIN_FAIL): the '
          'step was skipped (NaN losses, no parameter / moment / priority change)'
          + ('; falling back to separate launches' if fallback else ''))

  def ws_view(self, name: str, count: int) -> torch.Tensor:
    off = int(getattr(self.layout.c, 'ws_' + name))
    return self.ws[off:off + count]

  # -- the step -----------------------------------------------------------------
  def step(self, s_tm1, a_tm1, r_t, discount_t, s_t, weights,
           phases: int = _lib.PHASE_ALL, resample_noise: bool = True,
           priority_sink=None, next_sample=None) -> None:
    """Enqueues one learner step.  Inputs are device tensors exactly as
    `PrioritizedTransitionReplay.sample_device` returns them: uint8 states
    [B,84,84,4], int64 actions, float64 rewards/discounts, float32 weights.
    Results: `self.losses`, `self.priorities` (device), updated parameters.

    `priority_sink` (from `PrioritizedTransitionReplay.priority_sink(ids)`):
    the step writes the new priorities into that replay's sum tree itself,
    inside its backward launches -- the caller then must NOT call
    `update_priorities` for this batch.  Needs PHASE_BACKWARD in `phases`.

    `next_sample` (descriptor from `PrioritizedTransitionReplay.prepare_next_sample`,
    with `priority_sink` and a full step): the optimiser launch carries the NEXT
    step's sample + gather as extra blocks -- after this step's write-back, so the
    order of replay operations is the sequential one.  Launches eagerly."""
    b = self.batch_size
    stream = _lib.stream_ptr(self.device)
    graphs = bool(stream) if self.use_graphs is None else self.use_graphs
    if next_sample is not None:
      graphs = False   # the draws are by-value kernel arguments
      if phases & (_lib.PHASE_BACKWARD | _lib.PHASE_OPTIMIZER) != (
          _lib.PHASE_BACKWARD | _lib.PHASE_OPTIMIZER):
        raise ValueError('next_sample needs the backward and optimiser phases in this call')
    nets = bool(phases & (_lib.PHASE_FORWARD | _lib.PHASE_FWD_NETS))
    if graphs and self._args is not None and weights is not None:
      # fast path: a call signature that was validated and captured before (the
      # replay hands out the same ring slots over and over) is one graph launch
      sink = priority_sink if priority_sink is not None else (None,) * 4
      key = (s_tm1.data_ptr(), s_t.data_ptr(), a_tm1.data_ptr(), r_t.data_ptr(),
             discount_t.data_ptr(), weights.data_ptr(), phases,
             int(bool(resample_noise) and nets),
             sink[0], sink[3], int(self.keep_all_grads), int(self.separate_launches))
      g = self._graphs.get(key)
      if g is not None:
        _lib.check(self._lib.dz_graph_launch(g, stream), 'dz_graph_launch')
        return
    check_batch(b, s_tm1, a_tm1, r_t, discount_t, s_t, weights)
    if weights is None:
      raise TypeError('weights must be a float32 device tensor')
    a = self._args
    if a is None:  # everything that never changes is filled once
      a = self._args = _lib.RainbowArgs()
      a.num_actions = self.network.num_actions
      a.num_atoms = self.network.num_atoms
      a.batch = b
      a.online = self.online.data_ptr()
      a.target = self.target.data_ptr()
      a.grad = self.grad.data_ptr()
      a.adam_m = self.adam_m.data_ptr()
      a.adam_v = self.adam_v.data_ptr()
      a.adam_count = self.adam_count.data_ptr()
      a.support = self.support.data_ptr()
      a.noise = self.noise.data_ptr()
      a.ws = self.ws.data_ptr()
      a.losses = self.losses.data_ptr()
      a.priorities = self.priorities.data_ptr()
      a.lr = self.opt.learning_rate
      a.b1 = self.opt.b1
      a.b2 = self.opt.b2
      a.eps = self.opt.eps
      a.max_norm = self.opt.max_global_grad_norm
      a.noise_seed = self._noise_seed
    a.s_tm1 = s_tm1.data_ptr()
    a.s_t = s_t.data_ptr()
    a.a_tm1 = a_tm1.data_ptr()
    a.r_t = r_t.data_ptr()
    a.discount_t = discount_t.data_ptr()
    a.weights = weights.data_ptr()
    # fresh noise for the 3 applies is generated by the step itself from
    # (seed, Adam step count): no per-step host argument, graph-replayable.
    a.resample_noise = int(bool(resample_noise) and nets)
    a.keep_all_grads = int(self.keep_all_grads)
    a.separate_launches = int(self.separate_launches)
    a.next_sample = None if next_sample is None else ctypes.addressof(next_sample)
    if priority_sink is not None:
      if not phases & _lib.PHASE_BACKWARD:
        raise ValueError('priority_sink needs the backward phase in this call')
      (a.prio_node, a.prio_cap_pow2, a.prio_capacity, a.prio_ids, a.prio_exponent,
       a.prio_max_seen, a.prio_status) = priority_sink
    else:
      a.prio_node = None
      a.prio_ids = None
    stream = _lib.stream_ptr(self.device)
    if graphs:
      if not stream:
        raise RuntimeError(
            'hipGraph capture needs a non-default stream: run the learner under '
            '`torch.cuda.stream(torch.cuda.Stream())` (bench.py does)')
      key = (a.s_tm1, a.s_t, a.a_tm1, a.r_t, a.discount_t, a.weights, phases,
             a.resample_noise, a.prio_node, a.prio_ids, a.keep_all_grads, a.separate_launches)
      g = self._graphs.get(key)
      if g is None:
        if self.use_graphs is None and len(self._graphs) >= self.MAX_AUTO_GRAPHS:
          # automatic mode met a caller that passes fresh buffers every step: a
          # graph per call would only grow the cache -- launch eagerly from now on
          self.drop_graphs()
          self.use_graphs = False
          _lib.check(self._lib.dz_rainbow_learn(ctypes.byref(a), phases, stream),
                     'dz_rainbow_learn')
          return
        h = ctypes.c_void_p()
        _lib.check(self._lib.dz_rainbow_graph_capture(
            ctypes.byref(a), phases, stream, ctypes.byref(h)),
                   'dz_rainbow_graph_capture')
        g = self._graphs[key] = h
      _lib.check(self._lib.dz_graph_launch(g, stream), 'dz_graph_launch')
      return
    _lib.check(self._lib.dz_rainbow_learn(ctypes.byref(a), phases, stream),
               'dz_rainbow_learn')

  def __del__(self):
    try:
      self.drop_graphs()
    except Exception:  # pylint: disable=broad-except
      pass


# --------------------------------------------------------------------------- #
#  Dense-head learners (DQN, double-Q, prioritized, C51, QR-DQN)
# --------------------------------------------------------------------------- #
class RmsPropConfig(typing.NamedTuple):
  """optax.rmsprop(learning_rate, decay, eps, centered=True)
  (ref: dqn/run_atari.py:78-83, 205-210)."""
  learning_rate: float = 0.00025
  decay: float = 0.95
  eps: float = 0.01 / 32 ** 2


class DenseLearner:
  """The jitted `update` of the dense-head agents as one C-ABI call.

  loss: 'q' (dqn/agent.py:85-117), 'double_q' (double_q/agent.py:85-120 and,
  with importance weights, prioritized/agent.py:86-122), 'categorical'
  (c51/agent.py:87-116), 'quantile' (qrdqn/agent.py:88-119).
  optimizer: RmsPropConfig or AdamConfig."""

  LOSSES = {'q': _lib.LOSS_Q, 'double_q': _lib.LOSS_DOUBLE_Q,
            'categorical': _lib.LOSS_CATEGORICAL, 'quantile': _lib.LOSS_QUANTILE}

  def __init__(self, network: networks.DenseNetwork, loss: str, optimizer,
               batch_size: int, grad_error_bound: float = 1.0 / 32,
               huber_param: float = 1.0, seed: int = 1, device=None, params=None):
    self._lib = _lib.load()
    if not torch.cuda.is_available():
      raise _lib.HipLibraryError('DenseLearner needs an AMD GPU; no CPU fallback')
    self.device = torch.device('cuda', torch.cuda.current_device()) \
        if device is None else torch.device(device)
    self.network = network
    self.loss = loss
    self.loss_id = self.LOSSES[loss]
    self.opt = optimizer
    self.batch_size = int(batch_size)
    self.groups = 3 if loss == 'double_q' else 2
    self.layout = network.layout(self.batch_size, self.groups)
    L = self.layout
    if params is None:
      params = network.init(np.random.RandomState(seed))
    f32 = dict(dtype=torch.float32, device=self.device)
    self.online = torch.from_numpy(L.pack(params)).to(self.device)
    self.target = self.online.clone()
    self.grad = torch.zeros(L.param_count, **f32)
    self.opt_m = torch.zeros(L.param_count, **f32)
    self.opt_v = torch.zeros(L.param_count, **f32)
    self.opt_count = torch.zeros(1, dtype=torch.int32, device=self.device)
    self.ws = torch.zeros(L.ws_count, **f32)
    self.losses = torch.zeros(self.batch_size, **f32)      # td errors or losses
    self.priorities = torch.zeros(self.batch_size, **f32)  # |td|
    self.ones = torch.ones(self.batch_size, **f32)
    aux = network.support if network.kind == 'c51' else network.quantiles
    self.aux = None if aux is None else torch.from_numpy(aux).to(self.device)
    self.grad_error_bound = float(grad_error_bound)
    self.huber_param = float(huber_param)
    self._act_ws = None
    self._act_batch = 0
    self._graphs = {}        # (input pointers, phases, sink) -> hipGraphExec
    self.use_graphs = None   # True / False / None = whenever the stream allows capture
    # False (default): a full RMSProp step of a narrow-head learner never stores fc1's
    # weight gradient (dz_dense_args_t::keep_all_grads); True materialises every block
    self.keep_all_grads = False

  def get_params(self, which='online') -> dict:
    t = self.online if which == 'online' else self.target
    return self.layout.unpack(t.cpu().numpy())

  def set_params(self, params: dict, which='online') -> None:
    t = self.online if which == 'online' else self.target
    t.copy_(torch.from_numpy(self.layout.pack(params)))

  def sync_target(self) -> None:
    _lib.check(self._lib.dz_param_copy(
        self.target.data_ptr(), self.online.data_ptr(), self.layout.param_count,
        _lib.stream_ptr(self.device)), 'dz_param_copy')

  def get_opt_state(self) -> dict:
    return dict(count=int(self.opt_count.item()),
                mu=self.layout.unpack(self.opt_m.cpu().numpy()),
                nu=self.layout.unpack(self.opt_v.cpu().numpy()))

  def set_opt_state(self, state: dict) -> None:
    self.opt_count.fill_(int(state['count']))
    self.opt_m.copy_(torch.from_numpy(self.layout.pack(state['mu'])))
    self.opt_v.copy_(torch.from_numpy(self.layout.pack(state['nu'])))

  def ws_view(self, name: str, count: int) -> torch.Tensor:
    off = int(getattr(self.layout.c, 'ws_' + name))
    return self.ws[off:off + count]

  def step(self, s_tm1, a_tm1, r_t, discount_t, s_t, weights=None,
           phases: int = _lib.PHASE_ALL, priority_sink=None, next_sample=None) -> None:
    """`priority_sink`: see RainbowLearner.step (the |td| priorities go into the
    replay's sum tree inside the backward launches).  `next_sample`: descriptor from
    the replay's `prepare_next_sample` -- the optimiser launch carries the NEXT step's
    sample + gather (full step, eager launches; RainbowLearner.step)."""
    b = self.batch_size
    check_batch(b, s_tm1, a_tm1, r_t, discount_t, s_t, weights)
    a = _lib.DenseArgs()
    net = self.network
    a.loss = self.loss_id
    adam = isinstance(self.opt, AdamConfig)
    a.optimizer = _lib.OPT_ADAM if adam else _lib.OPT_RMSPROP
    a.num_actions = net.num_actions
    a.num_outputs = net.num_outputs
    a.batch = b
    a.shared_bias = int(net.shared_bias)
    a.num_atoms = net.num_atoms
    a.online = self.online.data_ptr()
    a.target = self.target.data_ptr()
    a.grad = self.grad.data_ptr()
    a.opt_m = self.opt_m.data_ptr()
    a.opt_v = self.opt_v.data_ptr()
    a.opt_count = self.opt_count.data_ptr()
    a.s_tm1 = s_tm1.data_ptr()
    a.s_t = s_t.data_ptr()
    a.a_tm1 = a_tm1.data_ptr()
    a.r_t = r_t.data_ptr()
    a.discount_t = discount_t.data_ptr()
    if weights is None and self.loss == 'categorical':
      weights = self.ones   # the categorical head kernel always takes weights
    if weights is not None:
      a.weights = weights.data_ptr()
    a.aux = None if self.aux is None else self.aux.data_ptr()
    a.ws = self.ws.data_ptr()
    a.losses = self.losses.data_ptr()
    a.priorities = self.priorities.data_ptr()
    if adam:
      a.lr, a.decay_or_b1, a.b2 = self.opt.learning_rate, self.opt.b1, self.opt.b2
      a.eps, a.max_norm = self.opt.eps, self.opt.max_global_grad_norm
    else:
      a.lr, a.decay_or_b1, a.b2 = self.opt.learning_rate, self.opt.decay, 0.0
      a.eps, a.max_norm = self.opt.eps, 0.0
    a.grad_error_bound = self.grad_error_bound
    a.huber = self.huber_param
    a.keep_all_grads = int(self.keep_all_grads)
    if priority_sink is not None:
      if not phases & _lib.PHASE_BACKWARD:
        raise ValueError('priority_sink needs the backward phase in this call')
      (a.prio_node, a.prio_cap_pow2, a.prio_capacity, a.prio_ids, a.prio_exponent,
       a.prio_max_seen, a.prio_status) = priority_sink
    if next_sample is not None:
      if phases & (_lib.PHASE_BACKWARD | _lib.PHASE_OPTIMIZER) != (
          _lib.PHASE_BACKWARD | _lib.PHASE_OPTIMIZER):
        raise ValueError('next_sample needs the backward and optimiser phases in this call')
      a.next_sample = ctypes.addressof(next_sample)
    stream = _lib.stream_ptr(self.device)
    enqueue = lambda: _lib.check(self._lib.dz_dense_learn(
        ctypes.byref(a), phases, stream), 'dz_dense_learn')
    if next_sample is not None or not (
        bool(stream) if self.use_graphs is None else self.use_graphs):
      enqueue()   # (next_sample: its draws are by-value kernel arguments -- no graph)
      return
    # every argument is a pointer or a constant of this object: the launches of
    # one call signature are captured once and replayed (as jax.jit does)
    key = (a.s_tm1, a.s_t, a.a_tm1, a.r_t, a.discount_t, a.weights, phases,
           a.prio_node, a.prio_ids, a.keep_all_grads)
    g = self._graphs.get(key)
    if g is None:
      if self.use_graphs is None and len(self._graphs) >= RainbowLearner.MAX_AUTO_GRAPHS:
        self.drop_graphs()       # fresh buffers every step: stop capturing (see RainbowLearner)
        self.use_graphs = False
        enqueue()
        return
      g = self._graphs[key] = _lib.capture_graph(stream, enqueue)
    _lib.check(self._lib.dz_graph_launch(g, stream), 'dz_graph_launch')

  def drop_graphs(self) -> None:
    graphs, self._graphs = self._graphs, {}
    for g in graphs.values():
      self._lib.dz_graph_destroy(g)
    heads, self._head_graphs = getattr(self, '_head_graphs', {}), {}
    for g in heads.values():
      self._lib.dz_graph_destroy(g)

  def __del__(self):
    try:
      self.drop_graphs()
    except Exception:  # pylint: disable=broad-except
      pass

  ACT_RING = 8

  def _realloc_act_ws(self, b: int) -> None:
    """New acting workspace for batch b.  The captured head graphs bake the OLD
    workspace's address in: destroy them first (replaying one would run the whole
    acting forward in memory the allocator may have handed to another tensor)."""
    heads, self._head_graphs = getattr(self, '_head_graphs', {}), {}
    for g in heads.values():
      self._lib.dz_graph_destroy(g)
    self._act_ws = torch.zeros(self.network.layout(b, 1).ws_count, dtype=torch.float32,
                               device=self.device)
    self._act_batch = b

  act_one_launch = True   # the decision for one state is ONE launch, its slot is polled
  last_act_fail = 0

  def _reset_act_seams(self) -> None:
    """As RainbowLearner._reset_act_seams."""
    torch.cuda.synchronize(self.device)
    if self._act_ws is None:
      return
    lay = self.network.layout(self._act_batch, 1).c
    off = int(lay.ws_act_seams)
    self.last_act_fail = int(self._act_ws[off + 5 * 64:off + 5 * 64 + 1].view(torch.int32).item())
    self._act_ws[off:int(lay.ws_count)].zero_()
    torch.cuda.synchronize(self.device)

  def _q_async(self, states: torch.Tensor):
    """`head_async` as ONE launch (dz_dense_act): every head output (A q-values, or the
    51 A / 201 A distribution outputs of C51 / QR-DQN) lands in the pinned slot as an 8-byte
    {value, marker} word; the host clears the
    words before the enqueue and `read()` polls the markers with plain loads -- no graph,
    no event, no completion signal.  Falls back to a stream sync if the markers do not
    show up."""
    a = self.network.num_outputs
    if getattr(self, '_q_host', None) is None:
      self._q_host = torch.zeros((self.ACT_RING, a, 2), dtype=torch.float32).pin_memory()
      self._q_host_np = self._q_host.numpy()
      self._q_pos = 0
      self._q_calls = {}
    if self._act_batch != 1:
      self._realloc_act_ws(1)
    k = self._q_pos % self.ACT_RING
    self._q_pos += 1
    words = self._q_host_np[k]
    words[:] = 0.0
    key = (states.data_ptr(), k, self._act_ws.data_ptr(), self.online.data_ptr())
    hit = self._q_calls.get(key)
    sp = _lib.stream_ptr(self.device)
    if hit is not None and hit[2] == sp:
      # steady state of an agent loop: the pre-bound enqueue and the slot's reader (whose
      # time-out fallback synchronises the stream it was enqueued on: the same one)
      hit[0](sp)
      return hit[1]
    if len(self._q_calls) > 4 * self.ACT_RING * 64:
      self._q_calls.clear()
    fn, chk = self._lib.dz_dense_act, _lib.check
    args = (a, int(self.network.shared_bias), self.online.data_ptr(), states.data_ptr(),
            self._act_ws.data_ptr(), self._q_host[k].data_ptr())
    call = lambda st: chk(fn(*args, st), 'dz_dense_act')
    call(sp)
    marks, vals = words[:, 1], words[:, 0]
    enq_stream = _lib.current_stream(self.device)   # the stream THIS decision was enqueued on
    owner = self

    n_out = float(len(marks))

    def read():
      # (one reduction per look: every marker 1.0 <=> sum == n; a FAILED marker is 2.0)
      deadline = None
      while marks.min() == 0.0:
        now = time.monotonic()
        if deadline is None:
          deadline = now + RainbowLearner.ACT_POLL_SECONDS
        elif now > deadline:
          enq_stream.synchronize()   # stuck or very slow: the stream the kernel is on decides
          break
        elif now > deadline - RainbowLearner.ACT_POLL_SECONDS + 2e-4:
          time.sleep(0)   # 200 us late: yield the GIL between looks
      if marks.sum() != n_out:
        # a seam of the decision kernel timed out (DZ_ACT_FAILED_MARKER) or outputs never arrived
        owner._reset_act_seams()   # pylint: disable=protected-access
        raise ActDecisionError(
            'dz_dense_act: the one-launch decision did not complete (sticky failure word %d); '
            'the acting workspace was re-armed' % owner.last_act_fail)
      return vals.copy()

    self._q_calls[key] = (call, read, sp)
    return read

  def head_async(self, states: torch.Tensor):
    """Acting apply for ONE state: the head outputs go straight to a pinned host
    slot (the copy-out of dz_dense_apply targets it), nothing synchronises here, and
    the launches replay from a hipGraph per (state buffer, slot) pair when the stream
    allows capture.  Returns `read() -> float32 [num_outputs]` (host array), which
    waits for these launches only."""
    if int(states.shape[0]) != 1:
      raise ValueError('head_async takes one state')
    net = self.network
    if self.act_one_launch:
      return self._q_async(states)
    if getattr(self, '_head_host', None) is None:
      self._head_host = torch.empty((self.ACT_RING, net.num_outputs),
                                    dtype=torch.float32).pin_memory()
      self._head_events = [torch.cuda.Event() for _ in range(self.ACT_RING)]
      self._head_graphs = getattr(self, '_head_graphs', {})
      self._head_pos = 0
    if self._act_batch != 1:
      self._realloc_act_ws(1)
    k = self._head_pos % self.ACT_RING
    self._head_pos += 1
    slot, ev = self._head_host[k], self._head_events[k]
    stream = _lib.stream_ptr(self.device)
    # Q heads: the head kernel writes the q-values through the slot's device mapping
    # itself (no copy node); distributional heads copy their outputs out
    is_q = net.num_outputs == net.num_actions
    enqueue = lambda: _lib.check(self._lib.dz_dense_apply(
        net.num_actions, net.num_outputs, int(net.shared_bias), 1, self.online.data_ptr(),
        states.data_ptr(), self._act_ws.data_ptr(), None if is_q else slot.data_ptr(),
        slot.data_ptr() if is_q else None, None, None, stream), 'dz_dense_apply')
    if stream:
      key = (states.data_ptr(), k)
      g = self._head_graphs.get(key)
      if g is None:
        g = self._head_graphs[key] = _lib.capture_graph(stream, enqueue)
      _lib.check(self._lib.dz_graph_launch(g, stream), 'dz_graph_launch')
    else:
      enqueue()
    ev.record(_lib.current_stream(self.device))

    def read():
      ev.synchronize()
      return slot.numpy().copy()

    return read

  def apply(self, states: torch.Tensor, which: str = 'online'):
    """Head outputs [B, num_outputs] for uint8 states; for Q heads also
    (q_values, greedy, max).  ref: dqn/agent.py:121-131."""
    assert states.dtype == torch.uint8 and states.is_contiguous()
    b = int(states.shape[0])
    net = self.network
    if self._act_batch != b:
      self._realloc_act_ws(b)
    out = torch.empty((b, net.num_outputs), dtype=torch.float32, device=self.device)
    is_q = net.num_outputs == net.num_actions
    q = torch.empty((b, net.num_actions), dtype=torch.float32, device=self.device) \
        if is_q else None
    greedy = torch.empty(b, dtype=torch.int32, device=self.device) if is_q else None
    vmax = torch.empty(b, dtype=torch.float32, device=self.device) if is_q else None
    params = self.online if which == 'online' else self.target
    _lib.check(self._lib.dz_dense_apply(
        net.num_actions, net.num_outputs, int(net.shared_bias), b,
        params.data_ptr(), states.data_ptr(), self._act_ws.data_ptr(),
        out.data_ptr(), None if q is None else q.data_ptr(),
        None if greedy is None else greedy.data_ptr(),
        None if vmax is None else vmax.data_ptr(),
        _lib.stream_ptr(self.device)), 'dz_dense_apply')
    return out, q, greedy, vmax


# --------------------------------------------------------------------------- #
#  IQN learner
# --------------------------------------------------------------------------- #
class IqnLearner:
  """The jitted `update` of the IQN agent (ref: iqn/agent.py:176-232) as one
  C-ABI call.  The three tau draws of a step (`_sample_tau`, iqn/agent.py:47-51)
  are made on the device by a counter-based generator unless given."""

  def __init__(self, network: networks.IqnNetwork, optimizer: AdamConfig,
               batch_size: int, tau_samples=(64, 64, 64), huber_param: float = 1.0,
               seed: int = 1, device=None, params=None):
    self._lib = _lib.load()
    if not torch.cuda.is_available():
      raise _lib.HipLibraryError('IqnLearner needs an AMD GPU; no CPU fallback')
    self.device = torch.device('cuda', torch.cuda.current_device()) \
        if device is None else torch.device(device)
    self.network = network
    self.opt = optimizer
    self.batch_size = int(batch_size)
    self.tau_samples = tuple(int(x) for x in tau_samples)  # (s_tm1, policy, s_t)
    self.layout = network.layout(self.batch_size, self.tau_samples)
    L = self.layout
    if params is None:
      params = network.init(np.random.RandomState(seed))
    f32 = dict(dtype=torch.float32, device=self.device)
    self.online = torch.from_numpy(L.pack(params)).to(self.device)
    self.target = self.online.clone()
    self.grad = torch.zeros(L.param_count, **f32)
    self.opt_m = torch.zeros(L.param_count, **f32)
    self.opt_v = torch.zeros(L.param_count, **f32)
    self.opt_count = torch.zeros(1, dtype=torch.int32, device=self.device)
    self.ws = torch.zeros(L.ws_count, **f32)
    self.losses = torch.zeros(self.batch_size, **f32)
    b = self.batch_size
    self.taus = torch.zeros(b * sum(self.tau_samples), **f32)
    n0, n1, n2 = self.tau_samples
    self.tau_tm1 = self.taus[:b * n0].view(b, n0)
    self.tau_sel = self.taus[b * n0:b * (n0 + n1)].view(b, n1)
    self.tau_t = self.taus[b * (n0 + n1):].view(b, n2)
    self.huber_param = float(huber_param)
    self._seed = int(seed) & 0xFFFFFFFFFFFFFFFF
    self._act_ws = {}
    self._graphs = {}
    self.use_graphs = None   # True / False / None = whenever the stream allows capture

  drop_graphs = DenseLearner.drop_graphs
  __del__ = DenseLearner.__del__
  get_params = DenseLearner.get_params
  set_params = DenseLearner.set_params
  sync_target = DenseLearner.sync_target
  get_opt_state = DenseLearner.get_opt_state
  set_opt_state = DenseLearner.set_opt_state
  ws_view = DenseLearner.ws_view

  ACT_RING = 8
  ONE_LAUNCH_MAX_TAUS = 64      # dz_iqn_act: samples <= 64 (the reference's default), num_actions <= 32, latent_dim <= 64
  last_act_fail = 0

  def can_decide_in_one_launch(self, samples: int) -> bool:
    return (samples <= self.ONE_LAUNCH_MAX_TAUS and self.network.num_actions <= 32 and
            self.network.latent_dim <= 64 and self.network.latent_dim % 8 == 0)

  def q_async(self, states: torch.Tensor, samples: int, tau_seed: int, tau_counter: int,
              taus_out=None):
    """The actor's decision for ONE state as ONE launch (dz_iqn_act, csrc/dz_iqn_act.h):
    `samples` tau draws at stream position `tau_counter` of seed `tau_seed` (the draws
    `dz_uniform_fill` would make there; written to `taus_out` [samples] if given), q = mean over
    the taus.  Every q-value lands in a pinned slot as an 8-byte {value, marker} word; returns
    `read() -> float32 [num_actions]`, which polls the markers with plain loads.
    ref: iqn/agent.py:234-247."""
    a = self.network.num_actions
    if getattr(self, '_q_host', None) is None:
      self._q_host = torch.zeros((self.ACT_RING, a, 2), dtype=torch.float32).pin_memory()
      self._q_host_np = self._q_host.numpy()
      self._q_pos = 0
    key = (1, int(samples))
    if key not in self._act_ws:
      self._act_ws[key] = torch.zeros(self.network.layout(1, (int(samples), 1, 1)).ws_count,
                                      dtype=torch.float32, device=self.device)
    ws = self._act_ws[key]
    k = self._q_pos % self.ACT_RING
    self._q_pos += 1
    words = self._q_host_np[k]
    words[:] = 0.0
    _lib.check(self._lib.dz_iqn_act(
        a, self.network.latent_dim, int(samples), self.online.data_ptr(), states.data_ptr(),
        int(tau_seed) & 0xFFFFFFFFFFFFFFFF, int(tau_counter), 
        None if taus_out is None else taus_out.data_ptr(), ws.data_ptr(),
        self._q_host[k].data_ptr(), _lib.stream_ptr(self.device)), 'dz_iqn_act')
    marks, vals = words[:, 1], words[:, 0]
    enq_stream = _lib.current_stream(self.device)
    owner, samples = self, int(samples)

    def read():
      deadline = None
      while not marks.all():
        now = time.monotonic()
        if deadline is None:
          deadline = now + RainbowLearner.ACT_POLL_SECONDS
        elif now > deadline:
          enq_stream.synchronize()
          break
        elif now > deadline - RainbowLearner.ACT_POLL_SECONDS + 2e-4:
          time.sleep(0)   # 200 us late: yield the GIL between looks
      if not marks.all() or (marks == _lib.ACT_FAILED_MARKER).any():
        owner._reset_act_seams(samples)   # pylint: disable=protected-access
        raise ActDecisionError(
            'dz_iqn_act: the one-launch decision did not complete (sticky failure word %d); '
            'the acting workspace was re-armed' % owner.last_act_fail)
      return vals.copy()

    return read

  def _reset_act_seams(self, samples: int) -> None:
    """As RainbowLearner._reset_act_seams."""
    torch.cuda.synchronize(self.device)
    ws = self._act_ws.get((1, int(samples)))
    if ws is None:
      return
    lay = self.network.layout(1, (int(samples), 1, 1)).c
    off = int(lay.ws_act_seams)
    self.last_act_fail = int(ws[off + 5 * 64:off + 5 * 64 + 1].view(torch.int32).item())
    ws[off:int(lay.ws_count)].zero_()
    torch.cuda.synchronize(self.device)

  def sample_taus(self) -> None:
    """Fresh U[0,1) draws for the three tau sets; the stream position is the
    optimiser step count, read on the device (no host sync)."""
    _lib.check(self._lib.dz_uniform_fill(
        self.taus.data_ptr(), self.taus.numel(), self._seed, 0,
        self.opt_count.data_ptr(),
        _lib.stream_ptr(self.device)), 'dz_uniform_fill')

  def step(self, s_tm1, a_tm1, r_t, discount_t, s_t, taus=None,
           phases: int = _lib.PHASE_ALL) -> None:
    """taus: optional (tau_tm1 [B,N0], tau_sel [B,N1], tau_t [B,N2]) float32
    device tensors; default: drawn on the device."""
    b = self.batch_size
    check_batch(b, s_tm1, a_tm1, r_t, discount_t, s_t)
    if taus is not None:
      for dst, src in zip((self.tau_tm1, self.tau_sel, self.tau_t), taus):
        dst.copy_(src)
    a = _lib.IqnArgs()
    net = self.network
    a.num_actions, a.latent_dim, a.batch = net.num_actions, net.latent_dim, b
    for i in range(3):
      a.samples[i] = self.tau_samples[i]
    a.online = self.online.data_ptr()
    a.target = self.target.data_ptr()
    a.grad = self.grad.data_ptr()
    a.opt_m = self.opt_m.data_ptr()
    a.opt_v = self.opt_v.data_ptr()
    a.opt_count = self.opt_count.data_ptr()
    a.s_tm1 = s_tm1.data_ptr()
    a.s_t = s_t.data_ptr()
    a.a_tm1 = a_tm1.data_ptr()
    a.r_t = r_t.data_ptr()
    a.discount_t = discount_t.data_ptr()
    a.tau_tm1 = self.tau_tm1.data_ptr()
    a.tau_sel = self.tau_sel.data_ptr()
    a.tau_t = self.tau_t.data_ptr()
    a.ws = self.ws.data_ptr()
    a.losses = self.losses.data_ptr()
    a.lr, a.b1, a.b2 = self.opt.learning_rate, self.opt.b1, self.opt.b2
    a.eps, a.max_norm = self.opt.eps, self.opt.max_global_grad_norm
    a.huber = self.huber_param
    stream = _lib.stream_ptr(self.device)
    def enqueue():
      if taus is None:
        self.sample_taus()   # device draw keyed by the optimiser step count: graph-safe
      _lib.check(self._lib.dz_iqn_learn(ctypes.byref(a), phases, stream), 'dz_iqn_learn')

    graphs = bool(stream) if self.use_graphs is None else self.use_graphs
    if not graphs or taus is not None:   # caller-supplied taus are copied in per call
      enqueue()
      return
    key = (a.s_tm1, a.s_t, a.a_tm1, a.r_t, a.discount_t, phases)
    g = self._graphs.get(key)
    if g is None:
      if self.use_graphs is None and len(self._graphs) >= RainbowLearner.MAX_AUTO_GRAPHS:
        self.drop_graphs()
        self.use_graphs = False
        enqueue()
        return
      g = self._graphs[key] = _lib.capture_graph(stream, enqueue)
    _lib.check(self._lib.dz_graph_launch(g, stream), 'dz_graph_launch')

  def apply(self, states: torch.Tensor, taus: torch.Tensor, which: str = 'online'):
    """(q_dist [B,N,A], q_values [B,A], greedy [B], max [B]) for uint8 states
    and taus [B,N].  ref: iqn/agent.py:234-247."""
    return iqn_apply(self._lib, self.network, self._act_ws,
                     self.online if which == 'online' else self.target, states,
                     taus, self.device)


def iqn_apply(lib, net, ws_cache, params, states, taus, device):
  assert states.dtype == torch.uint8 and states.is_contiguous()
  assert taus.dtype == torch.float32 and taus.is_contiguous()
  b, n = int(taus.shape[0]), int(taus.shape[1])
  assert int(states.shape[0]) == b
  key = (b, n)
  if key not in ws_cache:
    ws_cache[key] = torch.zeros(net.layout(b, (n, 1, 1)).ws_count,
                                dtype=torch.float32, device=device)
  f32 = dict(dtype=torch.float32, device=device)
  q_dist = torch.empty((b, n, net.num_actions), **f32)
  q = torch.empty((b, net.num_actions), **f32)
  greedy = torch.empty(b, dtype=torch.int32, device=device)
  vmax = torch.empty(b, **f32)
  _lib.check(lib.dz_iqn_apply(
      net.num_actions, net.latent_dim, b, n, params.data_ptr(), states.data_ptr(),
      taus.data_ptr(), ws_cache[key].data_ptr(), q_dist.data_ptr(), q.data_ptr(),
      greedy.data_ptr(), vmax.data_ptr(),
      _lib.stream_ptr(device)), 'dz_iqn_apply')
  return q_dist, q, greedy, vmax


# --------------------------------------------------------------------------- #
#  Inference-only network (evaluation actor)
# --------------------------------------------------------------------------- #
class InferenceNet:
  """Parameters + workspace for network applies only (no optimiser state):
  what `parts.EpsilonGreedyActor` needs (ref: parts.py:342-411)."""

  def __init__(self, network, seed: int = 1, device=None):
    self._lib = _lib.load()
    if not torch.cuda.is_available():
      raise _lib.HipLibraryError('InferenceNet needs an AMD GPU; no CPU fallback')
    self.device = torch.device('cuda', torch.cuda.current_device()) \
        if device is None else torch.device(device)
    self.network = network
    self.is_rainbow = isinstance(network, networks.RainbowNetwork)
    self.is_iqn = isinstance(network, networks.IqnNetwork)
    if self.is_iqn:
      self.layout = network.layout(1, (1, 1, 1))  # parameters only; ws per apply
    else:
      self.layout = network.layout(1) if self.is_rainbow else network.layout(1, 1)
    self._iqn_ws = {}
    self.params = torch.zeros(self.layout.param_count, dtype=torch.float32,
                              device=self.device)
    self.ws = torch.zeros(self.layout.ws_count, dtype=torch.float32,
                          device=self.device)
    self._has_params = False
    if self.is_rainbow:
      self.noise = torch.zeros(self.layout.noise_stride, dtype=torch.float32,
                               device=self.device)
      self.support = torch.from_numpy(network.support).to(self.device)
    self._seed = int(seed) & 0xFFFFFFFFFFFFFFFF
    self._counter = 0

  def set_params(self, params) -> None:
    """params: dict of Haiku-shaped arrays (what `agent.online_params` returns)
    or a flat device tensor with the learner's layout."""
    if isinstance(params, torch.Tensor):
      self.params.copy_(params)
    else:
      self.params.copy_(torch.from_numpy(self.layout.pack(params)))
    self._has_params = True

  def iqn_q_values(self, obs_u8: torch.Tensor, tau_samples: int) -> np.ndarray:
    """Host Q-values [A] of the IQN net for one state with `tau_samples` fresh
    tau draws (ref: iqn/agent.py:72-83)."""
    if not self._has_params:
      raise RuntimeError('network_params have not been set')
    stream = _lib.stream_ptr(self.device)
    taus = torch.empty((1, tau_samples), dtype=torch.float32, device=self.device)
    _lib.check(self._lib.dz_uniform_fill(taus.data_ptr(), tau_samples, self._seed,
                                         self._counter, None, stream),
               'dz_uniform_fill')
    self._counter += tau_samples
    _, q, _, _ = iqn_apply(self._lib, self.network, self._iqn_ws, self.params,
                           obs_u8, taus, self.device)
    return q[0].cpu().numpy()

  def q_values(self, obs_u8: torch.Tensor) -> np.ndarray:
    """Host Q-values [A] for ONE uint8 state tensor [1,84,84,4] on the device."""
    if not self._has_params:
      raise RuntimeError('network_params have not been set')
    net = self.network
    stream = _lib.stream_ptr(self.device)
    if self.is_rainbow:
      n = self.noise.numel()
      _lib.check(self._lib.dz_noise_fill(self.noise.data_ptr(), n, self._seed,
                                         self._counter, stream), 'dz_noise_fill')
      self._counter += n
      q = torch.empty((1, net.num_actions), dtype=torch.float32, device=self.device)
      _lib.check(self._lib.dz_rainbow_apply(
          net.num_actions, net.num_atoms, 1, self.params.data_ptr(),
          obs_u8.data_ptr(), self.noise.data_ptr(), self.support.data_ptr(),
          self.ws.data_ptr(), q.data_ptr(), None, None, stream),
                 'dz_rainbow_apply')
      return q[0].cpu().numpy()
    out = torch.empty((1, net.num_outputs), dtype=torch.float32, device=self.device)
    _lib.check(self._lib.dz_dense_apply(
        net.num_actions, net.num_outputs, int(net.shared_bias), 1,
        self.params.data_ptr(), obs_u8.data_ptr(), self.ws.data_ptr(),
        out.data_ptr(), None, None, None, stream), 'dz_dense_apply')
    h = out[0].cpu().numpy()
    if net.kind == 'c51':
      lg = h.reshape(net.num_actions, net.num_atoms).astype(np.float64)
      lg -= lg.max(axis=1, keepdims=True)
      p = np.exp(lg)
      p /= p.sum(axis=1, keepdims=True)
      return (p * net.support[None, :]).sum(axis=1)
    if net.kind == 'qr':
      return h.reshape(net.num_atoms, net.num_actions).mean(axis=0)
    return h
