"""Device copies of the observations an agent has just acted on.

`agent.step()` uploads every observation once, for the action-selection
apply.  The transitions the accumulator emits afterwards are built from those
same observation objects (`s_t` is the current one, `s_tm1` the one from n
steps back; ref: replay.py:771-892), so the replay insert can take both states
from HBM instead of uploading 2 x 28 KB again per `add`.
Identity (`is`) of the host arrays is the key: a transition field that is not
one of the cached objects simply takes the host upload path.
"""

import numpy as np
import torch


def canonical_scalars(transition):
  """Scalar fields in the dtypes the reference's `np.stack` of Python scalars
  yields (replay.py:157-163: int -> int64, float -> float64).  An environment or
  processor that emits np.float32 rewards / np.int32 actions would otherwise
  freeze the store's dtype to that of the first item, and the learner kernels
  read int64 / float64."""
  return transition._replace(a_tm1=np.int64(transition.a_tm1),
                             r_t=np.float64(transition.r_t),
                             discount_t=np.float64(transition.discount_t))


def depth_for(accumulator, floor: int = 8) -> int:
  """Slots an ObservationCache needs so that no slot is rewritten while a replay
  insert that reads it can still be queued: the insert enqueued at frame t-1 reads
  the observations of frames t-1 and t-1-n (n-step window), and `step()` waits only
  for the acting launches, so slot reuse must be at least n + 2 frames apart (one
  spare frame on top).  n = 3 (Rainbow) fits the default 8; n = 7 would alias.

  The window is read from the accumulator's public `window_size` (this package's
  accumulators), else from the deque the reference's `NStepTransitionAccumulator`
  keeps (`_transitions`, replay.py:841).  An accumulator that tells neither (a wrapper,
  a custom class) gets UNKNOWN_WINDOW_DEPTH slots -- 1.8 MB of pinned memory instead of
  a silent fall-back to 8 that an n >= 6 window would alias."""
  n = getattr(accumulator, 'window_size', None)
  if n is None:
    for name in ('_transitions', '_window'):
      n = getattr(getattr(accumulator, name, None), 'maxlen', None)
      if n is not None:
        break
  if n is None:
    return max(int(floor), UNKNOWN_WINDOW_DEPTH)
  return max(int(floor), int(n) + 3)


UNKNOWN_WINDOW_DEPTH = 64
_TRANSITION_FIELDS = ('s_tm1', 'a_tm1', 'r_t', 'discount_t', 's_t')


class ObservationCache:

  def __init__(self, device, depth: int = 8, shape=(84, 84, 4)):
    self._depth = depth
    self._pin = torch.empty((depth,) + tuple(shape), dtype=torch.uint8,
                            pin_memory=True)
    self._pin_np = self._pin.numpy()
    self._pin_rows = [self._pin[k] for k in range(depth)]   # (the same tensor objects every frame)
    self._pin_batches = [self._pin[k:k + 1] for k in range(depth)]
    self._host = [None] * depth
    self._ext = [None] * depth   # device tensors handed in as observations
    self._slot_of = {}           # id(observation) -> slot (checked with `is`: ids are recycled)
    self._pos = 0

  def upload(self, observation) -> torch.Tensor:
    """Makes the observation readable by kernels; returns a [1, H, W, C] uint8
    tensor whose data_ptr() is valid on the device (a pinned host slot, or the
    caller's own CUDA tensor).  A slot is reused `depth` calls later; the agent
    waits for the acting launches of every frame, and the replay insert queued
    behind them reads slots at most n + 1 frames old (`depth_for`)."""
    k = self._pos % self._depth
    self._pos += 1
    old = self._host[k]
    if old is not None and self._slot_of.get(id(old)) == k:
      del self._slot_of[id(old)]
    self._slot_of[id(observation)] = k
    if isinstance(observation, torch.Tensor):
      # already in HBM (processors.atari(device_observations=True)): no copy at
      # all; remembered by identity so that the replay insert finds it too
      if observation.dtype != torch.uint8 or not observation.is_cuda:
        raise TypeError('device observations must be uint8 CUDA tensors')
      self._host[k] = observation
      self._ext[k] = observation.contiguous()
      return self._ext[k][None]
    self._ext[k] = None
    self._host[k] = observation
    # ZERO-COPY: the observation is written into a pinned, device-mapped host slot
    # and the kernels (conv1 of the acting apply, the replay insert) read it from
    # there over PCIe -- 28 KB per decision, once or twice -- instead of paying a
    # pinned copy plus an async H2D memcpy launch (~12 us of host time per frame
    # during which the GPU had nothing to do)
    np.copyto(self._pin_np[k], observation, casting='same_kind')
    return self._pin_batches[k]

  def lookup(self, observation):
    k = self._slot_of.get(id(observation))
    if k is None or self._host[k] is not observation:
      return None
    e = self._ext[k]
    return self._pin_rows[k] if e is None else e

  def on_device(self, transition):
    """The transition with `s_tm1` / `s_t` replaced by their device copies
    where the cache holds them."""
    a, b = self.lookup(transition.s_tm1), self.lookup(transition.s_t)
    if transition._fields == _TRANSITION_FIELDS:   # one construction instead of two `_replace`s
      return type(transition)(transition.s_tm1 if a is None else a, np.int64(transition.a_tm1),
                              np.float64(transition.r_t), np.float64(transition.discount_t),
                              transition.s_t if b is None else b)
    return canonical_scalars(transition)._replace(
        s_tm1=transition.s_tm1 if a is None else a,
        s_t=transition.s_t if b is None else b)

  def clear(self) -> None:
    self._host = [None] * self._depth
    self._ext = [None] * self._depth
    self._slot_of = {}
