"""Host logic of dqn_zoo_amd.processors (no GPU): the composable helpers against
the cases of the reference's processors_test.py:39-348 (ported, not copied), and
the `AtariPreprocessor` state machine against the LIVE reference `atari()` on
random episode streams (when /root/reference exists), with the pixel path of both
sides replaced by the CPU oracle so that only the control logic is compared."""

import collections

import numpy as np
import pytest

from dqn_zoo_amd import dm_env_shim as dm_env
from dqn_zoo_amd import processors
from oracle import processors_oracle as po
from oracle import ref_processors_loader as rpl

F, M, L = dm_env.StepType.FIRST, dm_env.StepType.MID, dm_env.StepType.LAST


def _ts(step_types):
  return [None if st is None else dm_env.TimeStep(
      st, None if st == F else 0, None if st == F else 0, 0) for st in step_types]


def test_fixed_padded_buffer():
  b = processors.FixedPaddedBuffer(length=4, initial_index=-1)
  outs = [list(b(i)) for i in range(1, 7)]
  assert outs == [[None, None, None, 1], [2, None, None, None], [2, 3, None, None],
                  [2, 3, 4, None], [2, 3, 4, 5], [6, None, None, None]]
  b.reset()
  assert list(b(-1)) == [None, None, None, -1]
  c = processors.FixedPaddedBuffer(length=3, initial_index=2)
  got = [''.join('~' if v is None else str(v) for v in c(i)) for i in range(7)]
  assert got == ['~~0', '1~~', '12~', '123', '4~~', '45~', '456']


def test_timestep_buffer_condition():
  cond = processors.TimestepBufferCondition(period=4)
  seq = [([None, None, None, F], True), ([M, None, None, None], False),
         ([M, M, None, None], False), ([M, M, M, None], False), ([M, M, M, M], True),
         ([M, None, None, None], False), ([M, M, None, None], False),
         ([M, M, M, None], False), ([M, M, M, M], True), ([M, None, None, None], False),
         ([M, L, None, None], True)]
  for st, want in seq:
    assert cond(_ts(st)) == want
  with pytest.raises(RuntimeError, match='Should have reset'):
    cond(_ts([M, L, F, None]))
  for bad in ([F, M, L], [F, M, F]):
    cond = processors.TimestepBufferCondition(period=3)
    for st in ([None, None, F], [F, None, None], [F, M, None]):
      cond(_ts(st))
    with pytest.raises(RuntimeError, match='at most one FIRST or LAST'):
      cond(_ts(bad))
  with pytest.raises(RuntimeError, match='should be FIRST'):
    processors.TimestepBufferCondition(period=3)(_ts([M, None, None]))


def _chain(n):
  return processors.Sequential(
      processors.FixedPaddedBuffer(length=n, initial_index=-1),
      processors.ConditionallySubsample(processors.TimestepBufferCondition(period=n)),
      processors.Maybe(processors.Sequential(processors.none_to_zero_pad,
                                             processors.named_tuple_sequence_stack)))


def _mk(kind, obs):
  return {'f': dm_env.restart, 'm': lambda observation: dm_env.transition(0, observation),
          'l': lambda observation: dm_env.termination(0, observation)}[kind](observation=obs)


def test_action_repeats_chain():
  proc = _chain(4)
  seq = [('f', '0001'), ('m', None), ('m', None), ('m', None), ('m', '2345'),
         ('m', None), ('l', '6700'), ('f', '0008'), ('m', None)]
  prev = None
  for i, (kind, want) in enumerate(seq, start=1):
    if prev is not None and prev.last():
      proc.reset()
    ts = _mk(kind, i)
    out = proc(ts)
    got = None if out is None else ''.join(str(o) for o in out.observation)
    assert got == want
    prev = ts
  proc = _chain(4)
  with pytest.raises(RuntimeError, match='reset'):
    for i, kind in enumerate('fmmmmmlfm', start=1):
      proc(_mk(kind, i))


def test_small_helpers():
  Pair = collections.namedtuple('Pair', ['a', 'b'])
  assert processors.ApplyToNamedTupleField('a', lambda x: x + 10)(Pair(1, 2)) == Pair(11, 2)
  for r, want in ((0, 0), (1, 1), (-1, -1), (-2.5, -2), (2.5, 2), (None, None)):
    assert processors.clip_reward(2)(r) == want
  for st, want in (([0, 0, 0, F], F), ([M, M, M, L], L), ([M, M, L, 0], L), ([M, M, M, M], M)):
    assert processors.reduce_step_type(np.asarray(st), debug=True) == want
  for st in ([0, 0, 0, M], [0, 0, 0, L], [M, 0, 0, 0], [L, 0, 0, M], [M, L, F, M]):
    with pytest.raises(ValueError):
      processors.reduce_step_type(np.asarray(st), debug=True)
  for r, want in (([None], None), ([0, 0, 0, None], None), ([0], 0), ([1, 2, 3], 6),
                  ([1, -2, 3], 2)):
    assert processors.aggregate_rewards(r, debug=True) == want
  for d, want in (([None], None), ([0, 0, None], None), ([1], 1), ([1, 1, 1], 1),
                  ([1, 1, 0], 0)):
    assert processors.aggregate_discounts(d, debug=True) == want
  for bad in ([1.0, None], [0.0, 1.0, None], [1.0, 0.0, None]):
    with pytest.raises(ValueError, match='None.*FIRST'):
      processors.aggregate_rewards(bad, debug=True)
    with pytest.raises(ValueError, match='None.*FIRST'):
      processors.aggregate_discounts(bad, debug=True)
  d = processors.Deque(3, initial_values=[7])
  d.reset()
  assert list(d(1)) == [7, 1] and list(d(2)) == [7, 1, 2] and list(d(3)) == [1, 2, 3]
  assert processors.apply_additional_discount(0.5)(None) is None
  assert processors.apply_additional_discount(0.5)(1.0) == 0.5
  pad = processors.trailing_zero_pad(3)([np.ones(2)])
  assert len(pad) == 3 and (pad[1] == 0).all()
  with pytest.raises(ValueError, match='at least one value'):
    processors.none_to_zero_pad([None, None])


@pytest.mark.parametrize('spec', [
    ('fmmmmm', 'n11111', '333333', 'n11111'), ('fmmmmm', 'n11111', '333222', 'n11011'),
    ('fmmmmm', 'n11111', '332211', 'n10101'), ('fmmlfm', '1110n1', '333355', '1110n1')])
def test_zero_discount_on_life_loss(spec):
  kinds, disc, lives, want = spec
  proc = processors.ZeroDiscountOnLifeLoss()
  for k, d, lv, w in zip(kinds, disc, lives, want):
    out = proc(dm_env.TimeStep({'f': F, 'm': M, 'l': L}[k], 8,
                               None if d == 'n' else float(d), (9, int(lv))))
    assert out.discount == (None if w == 'n' else float(w))


def test_resample_tables_match_the_oracle_and_pillow_geometry():
  for n_in, n_out in ((160, 84), (210, 84), (84, 84), (37, 84), (250, 42)):
    b, k = processors.resample_coeffs(n_in, n_out)
    ob, ok = po.resample_coeffs(n_in, n_out)
    np.testing.assert_array_equal(b, ob)
    np.testing.assert_array_equal(k, ok)
    assert (k.sum(axis=1) - (1 << 22)).__abs__().max() <= k.shape[1]  # ~normalised
  b, k = processors.resample_coeffs(84, 84)   # identity: one full-weight tap at x
  assert (k.max(axis=1) == 1 << 22).all()


class OraclePixels:
  """Pixel path of the test double: the CPU oracle (tests may use it)."""

  def __init__(self, pooled=2, stacked=4, grayscaling=True):
    self._pooled, self._stacked, self._gray = pooled, stacked, grayscaling
    self._frames = collections.deque(maxlen=stacked)

  def reset(self):
    self._frames.clear()

  def __call__(self, frames):
    shape = next(f for f in frames if f is not None).shape
    real = [f for f in list(frames)[-self._pooled:] if f is not None]
    self._frames.append(po.pooled_frame(real + [np.zeros(shape, np.uint8)],
                                        grayscaling=self._gray))
    return po.stack_frames(list(self._frames), self._stacked)


def random_episodes(rs, n_steps, shape=(210, 160, 3)):
  """Raw (rgb, lives) timesteps: episodes of random length with life losses."""
  out, t = [], 0
  while t < n_steps:
    length = int(rs.randint(1, 14))
    lives = 3
    out.append(dm_env.restart((rs.randint(0, 256, shape, dtype=np.uint8), lives)))
    for i in range(length):
      if rs.uniform() < 0.2 and lives > 0:
        lives -= 1
      obs = (rs.randint(0, 256, shape, dtype=np.uint8), lives)
      r = float(rs.choice([-3.0, -1.0, 0.0, 0.5, 2.0]))
      if i == length - 1:
        out.append(dm_env.termination(r, obs) if rs.uniform() < 0.5
                   else dm_env.truncation(r, obs, 1.0))
      else:
        out.append(dm_env.transition(r, obs, 1.0))
    t += length + 1
  return out


def drive(proc, timesteps):
  outs, prev = [], None
  for ts in timesteps:
    if prev is not None and prev.last():
      processors.reset(proc)
    outs.append(proc(ts))
    prev = ts
  return outs


def same_timestep(a, b):
  if a is None or b is None:
    return a is None and b is None
  return (a.step_type == b.step_type and a.reward == b.reward and
          a.discount == b.discount and np.array_equal(a.observation, b.observation))


@pytest.mark.skipif(not rpl.reference_available(),
                    reason='needs /root/reference (dev container only)')
@pytest.mark.parametrize('seed,repeats,pooled,life_loss,clip,gray', [
    (0, 4, 2, True, 1.0, True), (1, 4, 2, False, None, True), (2, 3, 1, True, 1.0, True),
    (3, 2, 2, True, 2.5, True), (4, 4, 2, True, 1.0, False)])
def test_state_machine_equals_live_reference(seed, repeats, pooled, life_loss, clip, gray):
  """(gray=False: the reference's atari(grayscaling=False), whose RGB frames go through
  PIL as mode "RGB" -- observations [84, 84, 3, 4].)"""
  ref = rpl.load_reference_processors()
  saved = ref.rgb2y
  ref.rgb2y = po.rgb2y   # un-fused float64 (this container's BLAS fuses: see the oracle)
  try:
    want_proc = ref.atari(additional_discount=0.99, max_abs_reward=clip,
                          num_action_repeats=repeats, num_pooled_frames=pooled,
                          zero_discount_on_life_loss=life_loss, grayscaling=gray)
  finally:
    ref.rgb2y = saved
  got_proc = processors.AtariPreprocessor(
      0.99, clip, (84, 84), repeats, pooled, life_loss, 4, gray,
      observation_pipeline=OraclePixels(pooled, 4, grayscaling=gray))
  stream = random_episodes(np.random.RandomState(seed), 120)
  want = drive(want_proc, stream)
  got = drive(got_proc, stream)
  assert len(want) == len(got)
  emitted = 0
  for w, g in zip(want, got):
    assert same_timestep(w, g), (w, g)
    emitted += w is not None
  assert emitted > 20
