"""Host-side logic of dqn_zoo_amd.replay (no GPU): closed forms against the
golden traces of the reference, accumulators against the reference's own test
bodies (replay_test.py:177-357), importance-weight pins."""

import itertools
import os

import numpy as np
import pytest

from dqn_zoo_amd import dm_env_shim as dm_env
from dqn_zoo_amd import replay as replay_lib
from tests.golden import protocol

GOLDEN = os.path.join(os.path.dirname(__file__), 'golden')


@pytest.mark.parametrize('case', protocol.UNIFORM_CASES,
                         ids=[c[0] for c in protocol.UNIFORM_CASES])
def test_position_to_id_closed_form(case):
  name, cap, fill, batch, steps, seed = case
  g = np.load(os.path.join(GOLDEN, 'replay_uni_%s.npz' % name))
  t = fill + 4 * steps
  size = min(t, cap)
  np.testing.assert_array_equal(
      replay_lib.position_to_id(np.arange(size), t, cap), g['pos_to_id'])


@pytest.mark.parametrize('case', protocol.PRIORITIZED_CASES,
                         ids=[c[0] for c in protocol.PRIORITIZED_CASES])
def test_tree_index_closed_forms(case):
  name, cap, fill, batch, steps = case[:5]
  g = np.load(os.path.join(GOLDEN, 'replay_prio_%s.npz' % name))
  t = int(g['final_t'])
  assert t == fill + 4 * steps
  np.testing.assert_array_equal(
      replay_lib.tree_index_of_id(g['live_ids'], cap), g['live_tree_index'])
  size = min(t, cap)
  ids_at_pos = replay_lib.position_to_id(np.arange(size), t, cap)
  np.testing.assert_array_equal(
      replay_lib.tree_index_of_id(ids_at_pos, cap), g['active_indices'])


def test_closed_forms_every_t_small():
  """Brute-force model of swap-remove + free stack for every t up to 5N."""
  for cap in (1, 2, 3, 5, 8):
    ids, free, idx_of = [], list(range(cap)), {}
    for t in range(1, 5 * cap + 1):
      new = t - 1
      if len(ids) == cap:
        old = new - cap
        j = ids.index(old)
        ids[j] = ids[-1]
        ids.pop()
        free.append(idx_of.pop(old))
      idx_of[new] = free.pop()
      ids.append(new)
      np.testing.assert_array_equal(
          replay_lib.position_to_id(np.arange(len(ids)), t, cap), ids)
      live = sorted(idx_of)
      np.testing.assert_array_equal(
          replay_lib.tree_index_of_id(live, cap), [idx_of[i] for i in live])


def test_importance_sampling_weights_pins():
  w = replay_lib.importance_sampling_weights(np.array([0.1, 0.2]), 0.1, 1.0,
                                             True)
  np.testing.assert_array_equal(w, [1.0, 0.5])
  w = replay_lib.importance_sampling_weights(np.array([0.1, 0.2]), 0.1, 0.0,
                                             False)
  np.testing.assert_array_equal(w, [1.0, 1.0])
  with pytest.raises(ValueError):
    replay_lib.importance_sampling_weights(np.array([0.1]), 0.1, -0.1, True)
  with pytest.raises(ValueError):
    replay_lib.importance_sampling_weights(np.array([0.1]), 1.1, 0.5, True)
  with pytest.raises(ValueError):
    replay_lib.importance_sampling_weights(np.array([0.0]), 0.1, 0.5, True)


# ---- accumulators: bodies of replay_test.py:177-357 -------------------------
class _Fixture:

  def __init__(self):
    self.n = 3
    self.acc = replay_lib.NStepTransitionAccumulator(self.n)
    self.T = 10
    self.step_types = [dm_env.StepType.FIRST] + [dm_env.StepType.MID] * 9
    self.states = list(range(self.T))
    self.discounts = np.linspace(0.9, 1.0, self.T, endpoint=False)
    self.rewards = np.linspace(-5, 5, self.T, endpoint=False)
    self.actions = [i % 4 for i in range(self.T)]
    self.out = []
    for i in range(self.T):
      ts = dm_env.TimeStep(step_type=self.step_types[i],
                           observation=self.states[i],
                           discount=self.discounts[i], reward=self.rewards[i])
      self.out.append(list(self.acc.step(ts, self.actions[i])))


def test_nstep_basic_accumulation():
  f = _Fixture()
  assert f.out[:f.n] == [[]] * f.n and f.out[f.n] != []
  flat = list(itertools.chain(*f.out))
  np.testing.assert_array_equal([t.s_tm1 for t in flat], f.states[:-f.n])
  np.testing.assert_array_equal([t.s_t for t in flat], f.states[f.n:])
  np.testing.assert_array_equal([t.a_tm1 for t in flat], f.actions[:-f.n])
  exp_d = [np.prod(f.discounts[i + 1:i + 1 + f.n]) for i in range(f.T - f.n)]
  np.testing.assert_allclose([t.discount_t for t in flat], exp_d)
  exp_r = []
  for i in range(f.T - f.n):
    d = np.concatenate([[1.0], f.discounts[i + 1:i + f.n]])
    exp_r.append(np.sum(np.cumprod(d) * f.rewards[i + 1:i + 1 + f.n]))
  np.testing.assert_allclose([t.r_t for t in flat], exp_r)


def test_nstep_reset_and_first_requirement():
  f = _Fixture()
  f.acc.reset()
  ts = dm_env.TimeStep(dm_env.StepType.FIRST, 3, 1.0, -1)
  assert list(f.acc.step(ts, 1)) == []
  acc = replay_lib.NStepTransitionAccumulator(2)
  with pytest.raises(ValueError, match='Expected FIRST timestep'):
    list(acc.step(dm_env.TimeStep(dm_env.StepType.MID, 0., 1., 0), 0))
  acc1 = replay_lib.TransitionAccumulator()
  with pytest.raises(ValueError, match='Expected FIRST timestep'):
    list(acc1.step(dm_env.TimeStep(dm_env.StepType.MID, 0., 1., 0), 0))


def test_nstep1_equals_transition_accumulator():
  f = _Fixture()
  a, b = replay_lib.NStepTransitionAccumulator(1), \
      replay_lib.TransitionAccumulator()
  for i in range(f.T):
    ts = dm_env.TimeStep(f.step_types[i], f.rewards[i], f.discounts[i],
                         f.states[i])
    assert list(a.step(ts, f.actions[i])) == list(b.step(ts, f.actions[i]))


def test_nstep_flush_on_last():
  F, M, L = dm_env.StepType.FIRST, dm_env.StepType.MID, dm_env.StepType.LAST
  n = 3
  acc = replay_lib.NStepTransitionAccumulator(n)
  types = [F, M, M, M, M, M, L, F, M, M, M, M, F, M]
  T = len(types)
  disc = np.arange(1, T + 1) / T
  out = []
  for i in range(T):
    ts = dm_env.TimeStep(types[i], 1.0, disc[i], i)
    out.append(list(acc.step(ts, T - i)))
  assert [len(o) for o in out] == [0, 0, 0, 1, 1, 1, n, 0, 0, 0, 1, 1, 0, 0]
  end = out[6]
  assert [(t.s_tm1, t.s_t) for t in end] == [(3, 6), (4, 6), (5, 6)]


def test_nstep_short_episode():
  F, M, L = dm_env.StepType.FIRST, dm_env.StepType.MID, dm_env.StepType.LAST
  acc = replay_lib.NStepTransitionAccumulator(4)
  out = []
  for i, st in enumerate([F, M, L]):
    out.append(list(acc.step(dm_env.TimeStep(st, 1.0, 1.0, i), 1.0)))
  assert [[(t.s_tm1, t.s_t) for t in o] for o in out] == [[], [],
                                                           [(0, 2), (1, 2)]]


def test_observation_cache_depth_follows_the_n_step_window():
  """device_obs.depth_for: slot reuse must be > n + 2 frames apart; the window comes
  from the public `window_size`, from the reference accumulator's private deque, and
  an accumulator that tells neither gets the conservative depth."""
  import collections
  from dqn_zoo_amd import device_obs
  assert device_obs.depth_for(replay_lib.TransitionAccumulator()) == 8
  assert device_obs.depth_for(replay_lib.NStepTransitionAccumulator(3)) == 8
  assert device_obs.depth_for(replay_lib.NStepTransitionAccumulator(7)) == 10
  assert replay_lib.NStepTransitionAccumulator(5).window_size == 5

  class RefLike:   # the reference's attribute name (replay.py:841)
    def __init__(self, n):
      self._transitions = collections.deque(maxlen=n)

  assert device_obs.depth_for(RefLike(20)) == 23

  class Wrapper:
    def __init__(self, inner):
      self.inner = inner

  assert device_obs.depth_for(Wrapper(replay_lib.NStepTransitionAccumulator(9))) == \
      device_obs.UNKNOWN_WINDOW_DEPTH


def test_random_sample_is_uniform_bit_for_bit():
  """`sample_device` / `prepare_next_sample` draw their float64 uniforms with
  `RandomState.random_sample(n)`; the reference calls `uniform(size=n)` (replay.py:551-566).
  The two are the same doubles (0.0 + 1.0 * x) from the same stream positions."""
  a, b = np.random.RandomState(11), np.random.RandomState(11)
  for n in (1, 7, 32, 64, 257):
    assert a.randint(1000, size=n).tobytes() == b.randint(1000, size=n).tobytes()
    assert a.uniform(size=n).tobytes() == b.random_sample(n).tobytes()
    assert a.uniform(size=n).tobytes() == b.random_sample(n).tobytes()
  assert a.randint(10 ** 9) == b.randint(10 ** 9)
