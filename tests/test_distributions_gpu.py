"""General `PrioritizedDistribution` (host bookkeeping + device SumTree) against
golden traces generated from the reference for usage patterns its replays never
produce: non-consecutive ids, removals in random order, capacity growth and a
bounded capacity, an all-zero tree (ref: replay.py:429-651; the cases of
replay_test.py:431-744 for the error paths)."""

import os

import numpy as np
import pytest

from tests.golden import gen_distribution_golden as gen
from tests.golden import protocol

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(__file__), 'golden')


@pytest.mark.parametrize('case', gen.CASES, ids=[c[0] for c in gen.CASES])
def test_prioritized_distribution_equals_reference_trace(case):
  from dqn_zoo_amd import replay as rl
  name, seed, expo, usp, cmin, cmax, steps = case
  g = np.load(os.path.join(GOLDEN, 'dist_general_%s.npz' % name))
  dist = rl.PrioritizedDistribution(expo, usp, np.random.RandomState(seed), cmin, cmax)
  ids_log, probs_log = [], []
  live = gen.run_program(dist, seed, steps, cmax,
                         lambda step, ids, probs: (ids_log.append(ids), probs_log.append(probs)),
                         zero_priorities=name == 'zero_tree')
  np.testing.assert_array_equal(np.stack(ids_log), g['ids'])
  np.testing.assert_array_equal(protocol.f64_bits(np.stack(probs_log)), g['probs_bits'])
  st = dist.get_state()
  assert list(st['active_indices']) == list(g['active_indices'])
  assert list(st['inactive_indices']) == list(g['inactive_indices'])
  np.testing.assert_array_equal(protocol.f64_bits(st['sum_tree']['storage']), g['tree_bits'])
  assert dist.capacity == int(g['capacity']) and sorted(dist.ids()) == list(g['live'])
  assert dist.check_valid() == (True, '')
  # state round trip into a fresh object
  other = rl.PrioritizedDistribution(expo, usp, np.random.RandomState(seed), cmin, cmax)
  other.set_state(dist.get_state())
  assert other.check_valid()[0] and sorted(other.ids()) == sorted(live)
  np.testing.assert_array_equal(other.get_exponentiated_priorities(live),
                                dist.get_exponentiated_priorities(live))


def test_prioritized_distribution_errors():
  from dqn_zoo_amd import replay as rl
  rs = np.random.RandomState(0)
  for args, msg in (((-1.0, 0.1), 'priority_exponent >= 0'), ((1.0, 1.5), 'uniform_sample_probability')):
    with pytest.raises(ValueError, match=msg):
      rl.PrioritizedDistribution(args[0], args[1], rs)
  with pytest.raises(ValueError, match='max_capacity >= min_capacity'):
    rl.PrioritizedDistribution(1.0, 0.1, rs, min_capacity=5, max_capacity=3)
  d = rl.PrioritizedDistribution(1.0, 0.1, rs, min_capacity=2, max_capacity=4)
  with pytest.raises(RuntimeError, match='No IDs to sample'):
    d.sample(1)
  d.add_priorities([7, 3], [1.0, 2.0])
  with pytest.raises(IndexError, match='ID 3 already exists'):
    d.add_priorities([9, 3], [1.0, 1.0])
  with pytest.raises(ValueError, match='max capacity would be exceeded'):
    d.add_priorities([10, 11, 12], [1.0, 1.0, 1.0])
  with pytest.raises(ValueError, match='cannot exceed max_capacity'):
    d.ensure_capacity(5)
  with pytest.raises(IndexError, match='ID 99 does not exist'):
    d.update_priorities([99], [1.0])
  with pytest.raises(KeyError):      # the reference's quirk
    d.remove_priorities([99])
  with pytest.raises(ValueError, match='finite and positive'):
    d.update_priorities([7], [np.nan])
  d.add_priorities([10, 11], [0.5, 0.5])      # grows 2 -> 4
  assert d.capacity == 4 and d.size == 4
  ids, probs = d.sample(64)
  assert set(ids) <= {7, 3, 10, 11} and abs(probs.sum() / 64 - probs.mean()) < 1e-12
