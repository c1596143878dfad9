"""The Rainbow learner loop, software-pipelined across steps on two HIP streams.

One step of the reference's learner is  sample -> update -> priority write-back
(rainbow/agent.py:181-198).  Inside `update`, the target network's apply
(rainbow/agent.py:91-96) depends on the target parameters and on the sampled
batch -- never on the optimiser step that runs just before it; and sample(k+1)
depends on write-back(k), which needs only the per-sample losses of step k.  So
everything step k+1 needs from the replay and from the target network can run
UNDER step k's backward pass and optimiser:

    main : online(s_tm1), online(s_t) [k] -> loss(k) -> backward(k) -> clip/Adam(k) | online [k+1] ...
    side :                                   write-back(k) -> sample+gather(k+1) -> target(s_t) [k+1]
                                          ^ E_loss             E_sample ^            E_target ^

Three event hops per step, none of them on the critical path: the side chain is
~70 us of latency-bound launches against ~150 us of main-stream work.  The events
are device-scope (no system fence: `dz_event_create(1, ..)`).  Same operations in
the same data order as the sequential step: ids, losses, parameters and tree are
bit-identical (tests/test_pipeline_gpu.py).

This is the loop for a learner that consumes a replay nobody adds to between two
of its steps (the benchmark's metric; offline training).  An agent that inserts
transitions between learner steps must sample AFTER its inserts
(rainbow/agent.py:141-151) and uses the sequential step (`Rainbow._learn`).
"""

import ctypes

import torch

from dqn_zoo_amd import _lib


class _Events:
  """A small ring of device-scope events (re-recorded round-robin)."""

  def __init__(self, lib, n, device_scope):
    self._lib = lib
    self._ev = []
    for _ in range(n):
      h = ctypes.c_void_p()
      _lib.check(lib.dz_event_create(int(device_scope), ctypes.byref(h)),
                 'dz_event_create')
      self._ev.append(h)

  def __getitem__(self, k):
    return self._ev[k % len(self._ev)]

  def destroy(self):
    ev, self._ev = self._ev, []
    for h in ev:
      self._lib.dz_event_destroy(h)


class PipelinedRainbowLoop:
  """`step()` = one learner step (sample -> update -> write-back), pipelined.

  replay: PrioritizedTransitionReplay; learner: RainbowLearner; the CURRENT torch
  stream at construction is the main stream (non-default if hipGraphs are wanted).
  """

  RING = 4

  def __init__(self, replay, learner, batch_size: int, device_scope_events: bool = True,
               prefetch_target: bool = True):
    self._lib = _lib.load()
    self.replay, self.learner, self.batch = replay, learner, int(batch_size)
    self.device = learner.device
    self.main = torch.cuda.current_stream(self.device)
    self.side = torch.cuda.Stream(self.device)
    self._main_ptr = self.main.cuda_stream
    self._side_ptr = self.side.cuda_stream
    mk = lambda: _Events(self._lib, self.RING, device_scope_events)
    self._e_sample, self._e_target, self._e_loss, self._e_misc = mk(), mk(), mk(), mk()
    # False: only the replay operations (write-back, sample, gather) run ahead on the
    # side stream; the main stream keeps the three-apply step
    self.prefetch_target = bool(prefetch_target)
    self._k = 0
    self._next = None
    self._target_stale = False

  # -- plumbing -----------------------------------------------------------------
  def _record(self, ev, stream_ptr):
    _lib.check(self._lib.dz_event_record(ev, stream_ptr), 'dz_event_record')

  def _wait(self, stream_ptr, ev):
    _lib.check(self._lib.dz_stream_wait_event(stream_ptr, ev), 'dz_stream_wait_event')

  def _prefetch(self, k, first):
    """On the side stream: sample+gather for step k, then target(s_t) of it."""
    ln = self.learner
    torch.cuda.set_stream(self.side)
    try:
      s = self.replay.sample_device(self.batch)
      self._record(self._e_sample[k], self._side_ptr)
      if self.prefetch_target:
        ln.target_forward(s.transitions.s_t, step_from=ln.adam_count if first else None)
        self._record(self._e_target[k], self._side_ptr)
    finally:
      torch.cuda.set_stream(self.main)
    return s

  def _prime(self):
    # everything enqueued on main so far (fills, parameter writes) precedes the side chain
    self._record(self._e_misc[0], self._main_ptr)
    self._wait(self._side_ptr, self._e_misc[0])
    self._next = self._prefetch(self._k, first=True)

  # -- the step -------------------------------------------------------------------
  def step(self):
    ln, rep = self.learner, self.replay
    if self._next is None:
      self._prime()
    k, s = self._k, self._next
    t = s.transitions
    batch = (t.s_tm1, t.a_tm1, t.r_t, t.discount_t, t.s_t, s.weights32)
    # main: the two online applies need the sampled batch ...
    self._wait(self._main_ptr, self._e_sample[k])
    if not self.prefetch_target:
      ln.step(*batch, phases=_lib.PHASE_FORWARD)
      self._record(self._e_loss[k], self._main_ptr)
      ln.step(*batch, phases=_lib.PHASE_BACKWARD | _lib.PHASE_OPTIMIZER)
      self._after_loss(k, s)
      return s
    ln.step(*batch, phases=_lib.PHASE_FWD_NETS, target_pre=True)
    # ... the loss needs target(s_t)
    self._wait(self._main_ptr, self._e_target[k])
    if self._target_stale:
      # target parameters changed after the prefetch: redo the apply in line, with
      # the noise block that run drew (its stream position is already consumed)
      ln.target_forward(t.s_t, resample_noise=False)
      self._target_stale = False
    ln.step(*batch, phases=_lib.PHASE_FWD_LOSS, target_pre=True)
    self._record(self._e_loss[k], self._main_ptr)
    ln.step(*batch, phases=_lib.PHASE_BACKWARD | _lib.PHASE_OPTIMIZER, target_pre=True)
    self._after_loss(k, s)
    return s

  def _after_loss(self, k, s):
    # side, under the backward pass: write-back(k) -> sample(k+1) -> target apply(k+1)
    self._wait(self._side_ptr, self._e_loss[k])
    torch.cuda.set_stream(self.side)
    try:
      self.replay.update_priorities(s.ids, self.learner.priorities)
    finally:
      torch.cuda.set_stream(self.main)
    self._k = k + 1
    self._next = self._prefetch(self._k, first=False)

  def sync_target(self):
    """target <- online between two steps (rainbow/agent.py:157-158)."""
    if self._next is not None and self.prefetch_target:
      # the prefetched target apply may still be reading the old parameters
      self._wait(self._main_ptr, self._e_target[self._k])
      self._target_stale = True
    self.learner.sync_target()

  def drain(self):
    """Joins the side stream into main (the prefetched batch stays pending)."""
    self._record(self._e_misc[1], self._side_ptr)
    self._wait(self._main_ptr, self._e_misc[1])

  def close(self):
    self.drain()
    torch.cuda.synchronize(self.device)
    for e in (self._e_sample, self._e_target, self._e_loss, self._e_misc):
      e.destroy()
