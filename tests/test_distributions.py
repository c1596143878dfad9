"""General `UniformDistribution` (host only: runs on the CPU) against the golden
trace generated from the reference (tests/golden/gen_distribution_golden.py), and
the error messages the reference's tests pin (replay_test.py:39-117)."""

import os

import numpy as np
import pytest

from dqn_zoo_amd import distributions
from tests.golden import gen_distribution_golden as gen

GOLDEN = os.path.join(os.path.dirname(__file__), 'golden')


def test_uniform_distribution_equals_reference_trace():
  g = np.load(os.path.join(GOLDEN, 'dist_general_uniform.npz'))
  dist = distributions.UniformDistribution(np.random.RandomState(21))
  log = []
  live = gen.run_uniform(dist, 21, 150, lambda step, ids: log.append(ids))
  np.testing.assert_array_equal(np.stack(log), g['ids'])
  assert list(dist.get_state()['ids']) == list(g['order'])
  assert sorted(dist.ids()) == sorted(live) == list(g['live'])
  assert dist.check_valid() == (True, '') and dist.size == len(live)


def test_uniform_distribution_errors_and_state():
  dist = distributions.UniformDistribution(np.random.RandomState(0))
  dist.add([20, 19, 18])
  with pytest.raises(IndexError, match='Cannot add ID 19, it already exists'):
    dist.add([5, 19])
  assert dist.size == 3          # nothing added by the failed call
  with pytest.raises(IndexError, match='Cannot remove ID 7, it does not exist'):
    dist.remove([20, 7])
  assert dist.size == 3
  dist.remove([20])
  assert sorted(dist.ids()) == [18, 19]
  other = distributions.UniformDistribution(np.random.RandomState(0))
  other.set_state(dist.get_state())
  assert other.check_valid()[0] and sorted(other.ids()) == [18, 19]
  assert set(other.sample(50)) == {18, 19}
