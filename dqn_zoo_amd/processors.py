"""Atari timestep preprocessing with the pixel path on the GPU
(ref: dqn_zoo/processors.py).

The reference composes ~20 small processors into `atari()` and runs everything,
pixels included, in NumPy/PIL on the host (processors.py:421-508).  Here:

  * the CONTROL part -- life-loss discounts, action repeats, reward / discount /
    step-type aggregation -- is a few dozen scalar operations per frame and stays
    on the host, as one explicit state machine (`AtariPreprocessor`) instead of a
    chain of closures; the reference's composable pieces are provided too, for
    code that builds its own chain;
  * the PIXEL part -- max-pool of the last raw frames, `rgb2y`, the PIL-bilinear
    resize to 84x84 and the 4-frame stack (processors.py:367-387, 486-505) -- is
    ONE HIP launch per emitted observation (`dz_atari_observation`,
    csrc/dz_atari.hip).  The frame stack lives in HBM; with
    `device_observations=True` the stacked observation is handed to the agent as a
    device tensor, so acting and the replay insert read it where it already is.

Bit-exactness: the kernel's arithmetic is pinned, through the CPU oracle, to the
sha256 the reference's own test holds for this path (processors_test.py:472-475):
float64 un-fused `rgb2y`, Pillow's 22-bit fixed-point two-pass resample.  The
coefficient tables are computed here in float64 exactly as Pillow computes them.

There is no CPU fallback for the pixel path: without an AMD GPU the pipeline
raises on first use (control-only processors work anywhere).
"""

import collections
import math
from typing import Any, Callable, Iterable, List, Optional, Sequence, Tuple

import numpy as np

from dqn_zoo_amd import dm_env_shim as dm_env

Processor = Callable[[Any], Optional[Any]]
StepType = dm_env.StepType


def reset(processor: Processor) -> None:
  """Calls `reset()` on a processor if it has one (ref: processors.py:48-51)."""
  fn = getattr(processor, 'reset', None)
  if fn is not None:
    fn()


identity = lambda v: v


class Identity:
  """Processor for already-preprocessed observations (uint8 84x84x4 stacks)."""

  def __call__(self, timestep):
    return timestep

  def reset(self) -> None:
    pass


# --------------------------------------------------------------------------- #
#  Composable helpers (same names and behaviour as the reference's)
# --------------------------------------------------------------------------- #
def trailing_zero_pad(length: int):
  """Pads a list of arrays with zero arrays up to `length` entries."""

  def pad(arrays):
    missing = length - len(arrays)
    if missing <= 0:
      return arrays
    return arrays + [np.zeros_like(arrays[0])] * missing

  return pad


def none_to_zero_pad(values: List[Optional[Any]]) -> List[Any]:
  """`None` entries -> named tuples of zeros shaped like the real entries."""
  real = [v for v in values if v is not None]
  if not real:
    raise ValueError('Must have at least one value which is not None.')
  if len(real) == len(values):
    return values
  zero = type(real[0])(*[np.zeros_like(x) for x in real[0]])
  return [zero if v is None else v for v in values]


def named_tuple_sequence_stack(values: Sequence[Any]) -> Any:
  """[T(1, 2), T(3, 4)] -> T((1, 3), (2, 4))."""
  return type(values[0])(*zip(*values))


class Deque:
  """Bounded deque with initial values; returns itself after each append."""

  def __init__(self, max_length: int, initial_values: Optional[Iterable[Any]] = None):
    self._deque = collections.deque(maxlen=max_length)
    self._initial_values = list(initial_values or [])

  def reset(self) -> None:
    self._deque.clear()
    self._deque.extend(self._initial_values)

  def __call__(self, value: Any):
    self._deque.append(value)
    return self._deque


class FixedPaddedBuffer:
  """`None`-padded buffer of `length` slots that restarts once full: with length 3
  and initial_index 2 the values 0..6 come out as ~~0, 1~~, 12~, 123, 4~~, 45~, 456
  (ref: processors.py:109-148)."""

  def __init__(self, length: int, initial_index: int):
    self._length = length
    self._start = initial_index % length
    self.reset()

  def reset(self) -> None:
    self._pos = self._start
    self._slots = [None] * self._length

  def __call__(self, value: Any) -> Sequence[Any]:
    if self._pos == self._length:
      self._pos, self._slots = 0, [None] * self._length
    self._slots[self._pos] = value
    self._pos += 1
    return self._slots


class ConditionallySubsample:
  """Passes the input through when `condition(input)` holds, else `None`."""

  def __init__(self, condition):
    self._condition = condition

  def reset(self) -> None:
    reset(self._condition)

  def __call__(self, value: Any) -> Optional[Any]:
    return value if self._condition(value) else None


class _RepeatClock:
  """When does a buffer of timesteps go out?  On FIRST, on LAST, and every
  `period` steps after FIRST (ref: processors.py:164-222).  Shared by
  TimestepBufferCondition and AtariPreprocessor."""

  def __init__(self, period: int):
    self._period = period
    self.reset()

  def reset(self) -> None:
    self._since_first = None
    self._ended = False

  def tick(self, step_types: Iterable[Any]) -> bool:
    if self._ended:
      raise RuntimeError('Should have reset.')
    kind = StepType.MID
    for st in step_types:
      if st in (StepType.FIRST, StepType.LAST):
        if kind != StepType.MID:
          raise RuntimeError('Expected at most one FIRST or LAST.')
        kind = st
    if self._since_first is None and kind != StepType.FIRST:
      raise RuntimeError('After reset first timestep should be FIRST.')
    if kind == StepType.FIRST:
      self._since_first = 0
      return True
    if kind == StepType.LAST:
      self._since_first, self._ended = None, True
      return True
    self._since_first += 1
    return self._since_first % self._period == 0


class TimestepBufferCondition:
  """True when an iterable of timesteps (with `None` padding) should be passed on."""

  def __init__(self, period: int):
    self._clock = _RepeatClock(period)

  def reset(self):
    self._clock.reset()

  def __call__(self, timesteps: Iterable[Any]) -> bool:
    return self._clock.tick(t.step_type for t in timesteps if t is not None)


class ApplyToNamedTupleField:
  """Runs processors on one field of a named tuple."""

  def __init__(self, field: str, *processors):
    self._field = field
    self._processors = processors

  def reset(self) -> None:
    for p in self._processors:
      reset(p)

  def __call__(self, value):
    x = getattr(value, self._field)
    for p in self._processors:
      x = p(x)
    return value._replace(**{self._field: x})


class Maybe:
  """`None` in -> `None` out, otherwise the wrapped processor."""

  def __init__(self, processor):
    self._processor = processor

  def reset(self) -> None:
    reset(self._processor)

  def __call__(self, value):
    return None if value is None else self._processor(value)


class Sequential:
  """Chains processors."""

  def __init__(self, *processors):
    self._processors = processors

  def reset(self) -> None:
    for p in self._processors:
      reset(p)

  def __call__(self, value):
    for p in self._processors:
      value = p(value)
    return value


class ZeroDiscountOnLifeLoss:
  """Discount 0 on a MID timestep whose lives count (observation[1]) dropped."""

  def __init__(self):
    self._lives = None

  def reset(self) -> None:
    self._lives = None

  def __call__(self, timestep):
    lives = timestep.observation[1]
    lost = timestep.mid() and (lives < self._lives)
    self._lives = lives
    return timestep._replace(discount=0.0) if lost else timestep


def reduce_step_type(step_types: Sequence[Any], debug: bool = False):
  """Representative step type of a zero-padded buffer: padding (0) reads as FIRST
  and may only precede the FIRST; LAST may only be followed by padding."""
  for i, st in enumerate(step_types):
    if st == 0:
      if debug and not (np.array(step_types) == 0).all():
        raise ValueError('Expected zero padding followed by FIRST.')
      return StepType.FIRST
    if st == StepType.LAST:
      if debug and not (np.array(step_types)[i + 1:] == 0).all():
        raise ValueError('Expected LAST to be followed by zero padding.')
      return StepType.LAST
    if st != StepType.MID:
      raise ValueError('Expected MID if not FIRST or LAST.')
  return StepType.MID


def _only_first_is_none(values, what):
  arr = np.array(values)
  if not (arr[-1] is None and (arr[:-1] == 0).all()):
    raise ValueError('Should only have a None %s for FIRST.' % what)


def aggregate_rewards(rewards: Sequence[Optional[float]], debug: bool = False):
  """Sum of rewards (discount taken as 1); `None` if the buffer holds a FIRST."""
  if None in rewards:
    if debug:
      _only_first_is_none(rewards, 'reward')
    return None
  return sum(rewards)


def aggregate_discounts(discounts: Sequence[Optional[float]], debug: bool = False):
  """Product of discounts (each 0, 1 or `None`); `None` if there is a FIRST."""
  if debug and not np.isin(np.array(discounts), [0.0, 1.0, None]).all():
    raise ValueError('All discounts should be 0 or 1, got: %s.' % np.array(discounts))
  if None in discounts:
    if debug:
      _only_first_is_none(discounts, 'discount')
    return None
  out = 1
  for d in discounts:
    out *= d
  return out


def select_rgb_observation(timestep):
  """(rgb, lives) observation -> rgb."""
  return timestep._replace(observation=timestep.observation[0])


def apply_additional_discount(additional_discount: float):
  return lambda d: None if d is None else additional_discount * d


def clip_reward(bound: float):
  return lambda r: None if r is None else max(min(r, bound), -bound)


def show(prefix: str):
  def show_fn(value):
    print('%s: %s' % (prefix, value))
    return value
  return show_fn


# --------------------------------------------------------------------------- #
#  Pixel path on the device
# --------------------------------------------------------------------------- #
PRECISION_BITS = 32 - 8 - 2


def resample_coeffs(in_size: int, out_size: int):
  """Pillow's coefficient tables for an 8-bit BILINEAR resample of a whole axis
  (libImaging/Resample.c precompute_coeffs + normalize_coeffs_8bpc; what
  `Image.resize(..., BILINEAR)` of processors.py:383-385 runs): float64 triangle
  weights over `support = max(1, in/out)`, normalised, rounded to 22-bit fixed
  point.  Returns (bounds int32 [out, 2] = (first input index, taps),
  coeffs int32 [out, ksize])."""
  scale = float(in_size) / out_size
  fscale = max(scale, 1.0)
  support = 1.0 * fscale
  ksize = int(math.ceil(support)) * 2 + 1
  bounds = np.zeros((out_size, 2), np.int32)
  coeffs = np.zeros((out_size, ksize), np.int32)
  inv = 1.0 / fscale
  one = float(1 << PRECISION_BITS)
  for o in range(out_size):
    center = (o + 0.5) * scale
    lo = max(int(center - support + 0.5), 0)
    hi = min(int(center + support + 0.5), in_size)
    taps = []
    for x in range(lo, hi):
      t = abs((x - center + 0.5) * inv)
      taps.append(1.0 - t if t < 1.0 else 0.0)
    total = 0.0
    for w in taps:
      total += w
    for j, w in enumerate(taps):
      k = w / total if total != 0.0 else w
      coeffs[o, j] = int(-0.5 + k * one) if k < 0 else int(0.5 + k * one)
    bounds[o] = (lo, hi - lo)
  return bounds, coeffs


class ObservationPipeline:
  """max-pool -> grayscale -> resize -> stack, on the GPU
  (the observation branch of atari(), ref: processors.py:486-505).

  `__call__(frames)` takes the raw frames of one action-repeat buffer (host
  uint8 [H, W, 3] arrays, or [H, W] when `grayscaling=False`; `None` = padding),
  pools the last `num_pooled_frames` of them and returns the stacked
  observation [h, w, num_stacked_frames]: a NumPy array (one 28 KB device->host
  copy, synchronising) or, with `device_observations=True`, a uint8 CUDA tensor
  (no synchronisation; a fresh tensor per call, so callers may hold on to it).
  """

  def __init__(self, resize_shape=(84, 84), num_pooled_frames: int = 2,
               num_stacked_frames: int = 4, grayscaling: bool = True, device=None,
               device_observations: bool = False):
    if resize_shape is None or len(resize_shape) != 2:
      raise NotImplementedError(
          'the device pipeline resizes to a 2-D shape (resize_shape=None is not '
          'supported)')
    if not 1 <= num_pooled_frames <= 4:
      raise NotImplementedError('1 <= num_pooled_frames <= 4')
    if not 1 <= num_stacked_frames <= 8:
      raise NotImplementedError('1 <= num_stacked_frames <= 8')
    self._shape = (int(resize_shape[0]), int(resize_shape[1]))
    self._pooled = int(num_pooled_frames)
    self._stack = int(num_stacked_frames)
    self._gray = bool(grayscaling)
    self._device_arg = device
    self._device_out = bool(device_observations)
    self._ready = False
    self._count = 0     # frames in the stack
    self._slot = -1     # ring slot of the newest frame

  def reset(self) -> None:
    """Empties the frame stack (Deque.reset, processors.py:101-103)."""
    self._count, self._slot = 0, -1

  # -- lazy device set-up (so that building a processor needs no GPU) ----------
  def _setup(self, frame: np.ndarray) -> None:
    import torch  # pylint: disable=import-outside-toplevel
    from dqn_zoo_amd import _lib  # pylint: disable=import-outside-toplevel
    self._torch, self._lib_mod = torch, _lib
    self._lib = _lib.load()
    if not torch.cuda.is_available():
      raise _lib.HipLibraryError(
          'the Atari observation pipeline needs an AMD GPU; no CPU fallback')
    self._device = torch.device('cuda', torch.cuda.current_device()) \
        if self._device_arg is None else torch.device(self._device_arg)
    # [H, W, 3] RGB (grayscaled, or with grayscaling=False kept as three bands that
    # are resampled independently, like PIL's mode "RGB": processors.py:429,495) or,
    # with grayscaling=False, an already-gray [H, W] frame
    ok = (frame.ndim == 3 and frame.shape[2] == 3) or (frame.ndim == 2 and not self._gray)
    if not ok:
      raise ValueError('frames must be uint8 [H, W, 3]%s, got shape %s' % (
          '' if self._gray else ' or [H, W]', tuple(frame.shape)))
    self._in_shape = tuple(frame.shape)
    self._planes = 3 if (frame.ndim == 3 and not self._gray) else 1
    h, w = frame.shape[:2]
    oh, ow = self._shape
    xb, xk = resample_coeffs(w, ow)
    yb, yk = resample_coeffs(h, oh)
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(self._device)
    self._xb, self._xk, self._yb, self._yk = dev(xb), dev(xk), dev(yb), dev(yk)
    self._xks, self._yks = int(xk.shape[1]), int(yk.shape[1])
    self._ring = torch.zeros((self._stack, oh, ow, self._planes), dtype=torch.uint8,
                             device=self._device)
    # pinned staging for the raw frames: DEPTH launches may be in flight
    self._depth = 4
    self._pin = torch.empty((self._depth, self._pooled) + self._in_shape, dtype=torch.uint8,
                            pin_memory=True)
    self._raw = torch.empty((self._depth, self._pooled) + self._in_shape, dtype=torch.uint8,
                            device=self._device)
    self._events = [None] * self._depth
    self._stage = 0
    import ctypes  # pylint: disable=import-outside-toplevel
    self._ptrs = (ctypes.c_void_p * 4)()
    self._ready = True

  def __call__(self, frames: Sequence[Optional[np.ndarray]]):
    every = [f for f in frames if f is not None]
    if not every:
      raise ValueError('Must have at least one value which is not None.')
    # padding is all zeros and max(0, x) == x: only the real frames among the last
    # `num_pooled_frames` travel; a buffer cut short by LAST may leave none of them
    # there (M L ~ ~ pools two paddings): the pooled frame is then black, as in the
    # reference
    frames = [f for f in list(frames)[-self._pooled:] if f is not None]
    if not self._ready:
      self._setup(np.asarray(every[0]))
    torch = self._torch
    k = self._stage % self._depth
    self._stage += 1
    if self._events[k] is not None:
      self._events[k].synchronize()   # the upload that last used this slot is done
    stream = self._lib_mod.current_stream(self._device)
    for i, f in enumerate(frames):
      a = np.asarray(f)
      if a.dtype != np.uint8 or tuple(a.shape) != self._in_shape:
        raise ValueError('frame must be uint8 %s, got %s %s' % (
            self._in_shape, a.dtype, a.shape))
      self._pin[k, i].copy_(torch.from_numpy(np.ascontiguousarray(a)))
      self._raw[k, i].copy_(self._pin[k, i], non_blocking=True)
      self._ptrs[i] = self._raw[k, i].data_ptr()
    if self._events[k] is None:
      self._events[k] = torch.cuda.Event()
    self._events[k].record(stream)
    self._slot = (self._slot + 1) % self._stack
    self._count = min(self._count + 1, self._stack)
    oh, ow = self._shape
    shape = (oh, ow, self._stack) if self._planes == 1 else (oh, ow, 3, self._stack)
    obs = torch.empty(shape, dtype=torch.uint8, device=self._device)
    h, w = self._in_shape[:2]
    self._lib_mod.check(self._lib.dz_atari_observation(
        self._ptrs, len(frames), h, w, 3 if len(self._in_shape) == 3 else 1, int(self._gray),
        self._xb.data_ptr(), self._xk.data_ptr(), self._xks,
        self._yb.data_ptr(), self._yk.data_ptr(), self._yks, oh, ow,
        self._ring.data_ptr(), self._stack, self._slot, self._count, obs.data_ptr(),
        stream.cuda_stream), 'dz_atari_observation')
    return obs if self._device_out else obs.cpu().numpy()


def _device_single(array, resize_shape, grayscaling):
  pipe = ObservationPipeline(resize_shape, 1, 1, grayscaling)
  return pipe([array])[..., 0]


def rgb2y(array: np.ndarray) -> np.ndarray:
  """RGB uint8 [H, W, 3] -> grayscale uint8 [H, W] on the device: float64
  0.299 r + 0.587 g + (1 - (0.299 + 0.587)) b, truncated (ref: processors.py:367-371;
  identity resample tables)."""
  array = np.asarray(array)
  if array.ndim != 3:
    raise AssertionError('rgb2y expects a rank-3 array')
  return _device_single(array, array.shape[:2], True)


def resize(shape: Tuple[int, ...]):
  """Processor resizing a 2-D uint8 array to `shape` with Pillow's BILINEAR
  resample, on the device (ref: processors.py:374-387)."""
  if len(shape) != 2:
    raise ValueError('Resize shape has to be 2D, given: %s.' % str(shape))
  return lambda array: _device_single(np.asarray(array), tuple(shape), False)


# --------------------------------------------------------------------------- #
#  atari(): the standard DQN preprocessing as one state machine
# --------------------------------------------------------------------------- #
class AtariPreprocessor:
  """What `processors.atari()` of the reference does to a stream of raw
  timesteps whose observation is `(rgb_frame, lives)` (ref: processors.py:421-508):

    1. discount 0 on loss of life;                 5. resize (PIL bilinear);
    2. action repeats: a timestep comes out on     6. stack the last S frames,
       FIRST, on LAST and every R-th step, `None`     zeros for the missing ones;
       otherwise (= "repeat the action");          7. sum, then clip rewards;
    3. max-pool the last P raw frames;             8. multiply the discounts, times
    4. grayscale;                                     the additional discount.

      step type:   F   |  M   M   M   M  |  M   M   L  |  F
      frame:       A   |  B   C   D   E  |  F   G   H  |  I
      output:  max[0A] |  ~   ~   ~ max[DE] ~   ~ max[H0] max[0I]

  A buffer cut short by LAST is zero-padded like the reference's: its discount
  product is 0, and the pooled frame is the max over the last P slots of the
  PADDED buffer (black if both are padding, e.g. M L ~ ~).  Steps 3-6 are one
  launch on the GPU (`ObservationPipeline`)."""

  def __init__(self, additional_discount: float = 0.99,
               max_abs_reward: Optional[float] = 1.0,
               resize_shape: Optional[Tuple[int, int]] = (84, 84),
               num_action_repeats: int = 4, num_pooled_frames: int = 2,
               zero_discount_on_life_loss: bool = True, num_stacked_frames: int = 4,
               grayscaling: bool = True, device=None, device_observations: bool = False,
               observation_pipeline=None):
    self._additional_discount = additional_discount
    self._max_abs_reward = max_abs_reward
    self._repeats = int(num_action_repeats)
    self._life_loss = bool(zero_discount_on_life_loss)
    self._pooled = int(num_pooled_frames)
    self._pixels = observation_pipeline if observation_pipeline is not None else \
        ObservationPipeline(resize_shape, num_pooled_frames, num_stacked_frames,
                            grayscaling, device, device_observations)
    self._clock = _RepeatClock(self._repeats)
    self.reset()

  def reset(self) -> None:
    self._lives = None
    self._buffer = [None] * self._repeats   # FixedPaddedBuffer(length=R, initial_index=-1)
    self._pos = self._repeats - 1
    self._clock.reset()
    reset(self._pixels)

  def __call__(self, timestep):
    # 1. life loss (a MID step before any FIRST is refused by the clock below)
    rgb, lives = timestep.observation[0], timestep.observation[1]
    if (self._life_loss and timestep.mid() and self._lives is not None and
        lives < self._lives):
      timestep = timestep._replace(discount=0.0)
    self._lives = lives
    # 2. action-repeat buffer
    if self._pos == self._repeats:
      self._pos, self._buffer = 0, [None] * self._repeats
    self._buffer[self._pos] = timestep._replace(observation=rgb)
    self._pos += 1
    buf = self._buffer
    if not self._clock.tick(t.step_type for t in buf if t is not None):
      return None
    # step type: padding counts as FIRST (0), and only ever precedes a FIRST
    kinds = [0 if t is None else t.step_type for t in buf]
    step_type = reduce_step_type(kinds)
    # 7. rewards: None with a FIRST in the buffer, else the sum, clipped
    rewards = [0.0 if t is None else t.reward for t in buf]
    reward = None if None in rewards else sum(rewards)
    if reward is not None and self._max_abs_reward:
      reward = max(min(reward, self._max_abs_reward), -self._max_abs_reward)
    # 8. discounts: product over the PADDED buffer (padding contributes 0)
    discounts = [0.0 if t is None else t.discount for t in buf]
    if None in discounts:
      discount = None
    else:
      discount = 1
      for d in discounts:
        discount *= d
      discount = self._additional_discount * discount
    # 3-6. pixels
    observation = self._pixels([None if t is None else t.observation for t in buf])
    return dm_env.TimeStep(step_type=step_type, reward=reward, discount=discount,
                           observation=observation)


def atari(additional_discount: float = 0.99, max_abs_reward: Optional[float] = 1.0,
          resize_shape: Optional[Tuple[int, int]] = (84, 84), num_action_repeats: int = 4,
          num_pooled_frames: int = 2, zero_discount_on_life_loss: bool = True,
          num_stacked_frames: int = 4, grayscaling: bool = True, device=None,
          device_observations: bool = False) -> AtariPreprocessor:
  """Standard DQN preprocessing on Atari (ref: processors.py:421-508); same
  keyword arguments, plus `device` / `device_observations` (see
  ObservationPipeline)."""
  return AtariPreprocessor(additional_discount, max_abs_reward, resize_shape,
                           num_action_repeats, num_pooled_frames,
                           zero_discount_on_life_loss, num_stacked_frames, grayscaling,
                           device, device_observations)


class ObservationSpec(collections.namedtuple('ObservationSpec', 'shape dtype name')):
  """Stand-in for dm_env.specs.Array (shape, dtype, name)."""


class AtariEnvironmentWrapper:
  """Environment wrapper that applies `atari()` and performs the action repeats
  (ref: processors.py:511-601).  Expects an environment with (rgb, lives)
  observations, interleaved HWC pixels and zero-indexed actions."""

  def __init__(self, environment, additional_discount: float = 0.99,
               max_abs_reward: Optional[float] = 1.0,
               resize_shape: Optional[Tuple[int, int]] = (84, 84),
               num_action_repeats: int = 4, num_pooled_frames: int = 2,
               zero_discount_on_life_loss: bool = True, num_stacked_frames: int = 4,
               grayscaling: bool = True, device=None, device_observations: bool = False):
    rgb_spec, _ = environment.observation_spec()
    if rgb_spec.shape[2] != 3:
      raise ValueError('This wrapper assumes interleaved pixel observations with shape '
                       '(height, width, channels).')
    if int(environment.action_spec().minimum) != 0:
      raise ValueError('This wrapper assumes zero-indexed actions.')
    self._environment = environment
    self._processor = atari(additional_discount, max_abs_reward, resize_shape,
                            num_action_repeats, num_pooled_frames,
                            zero_discount_on_life_loss, num_stacked_frames, grayscaling,
                            device, device_observations)
    if grayscaling:
      self._obs_spec = ObservationSpec(tuple(resize_shape) + (num_stacked_frames,),
                                       np.uint8, 'grayscale')
    else:
      self._obs_spec = ObservationSpec(tuple(resize_shape) + (3, num_stacked_frames),
                                       np.uint8, 'RGB')
    self._needs_reset = True

  def reset(self):
    reset(self._processor)
    out = self._processor(self._environment.reset())
    assert out is not None
    self._needs_reset = False
    return out

  def step(self, action):
    """Repeats `action` until the processor emits a timestep."""
    if self._needs_reset:
      return self.reset()
    out = None
    while out is None:
      raw = self._environment.step(action)
      out = self._processor(raw)
      if raw.last():
        self._needs_reset = True
        assert out is not None
    return out

  def action_spec(self):
    return self._environment.action_spec()

  def observation_spec(self):
    return self._obs_spec
