"""End-to-end sanity beyond parity: agents driven through parts.run_loop on a
two-context bandit (observation tells which action pays) must end up preferring
the paying action in both contexts.  Exercises act -> accumulate -> device
insert -> sample -> learner step -> (priority write-back) -> target sync as one
system."""

import itertools

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

A = 3


class ContextBandit:
  """Episodes of one decision: FIRST(obs of context c) -> LAST(reward 1 if the
  action equals c else 0)."""

  def __init__(self, seed):
    self.rs = np.random.RandomState(seed)
    self.obs = []
    for c in range(2):
      o = np.zeros((84, 84, 4), np.uint8)
      o[:, 42 * c:42 * (c + 1), :] = 255
      self.obs.append(o)

  def reset(self):
    from dqn_zoo_amd import dm_env_shim as dm_env
    self.c = int(self.rs.randint(2))
    return dm_env.restart(self.obs[self.c])

  def step(self, action):
    from dqn_zoo_amd import dm_env_shim as dm_env
    return dm_env.termination(1.0 if action == self.c else 0.0, self.obs[1 - self.c])


def _q(agent, env, kind):
  out = []
  for c in range(2):
    x = torch.from_numpy(env.obs[c][None]).cuda()
    if kind == 'rainbow':
      q, _, _ = agent._learner.apply(x, resample_noise=False)  # pylint: disable=protected-access
      out.append(q[0].cpu().numpy())
    else:
      o, q, _, _ = agent._learner.apply(x)  # pylint: disable=protected-access
      out.append(q[0].cpu().numpy())
  return np.stack(out)


@pytest.mark.parametrize('kind', ['dqn', 'rainbow'])
def test_agent_learns_context_bandit(kind):
  from dqn_zoo_amd import learner, networks, parts, processors
  from dqn_zoo_amd import replay as rl
  T = rl.Transition(None, None, None, None, None)
  rs = np.random.RandomState(0)
  eps = parts.LinearSchedule(begin_t=0, decay_steps=600, begin_value=1.0, end_value=0.1)
  common = dict(preprocessor=processors.Identity(),
                sample_network_input=np.zeros((84, 84, 4), np.uint8),
                batch_size=32, min_replay_capacity_fraction=0.05, learn_period=1,
                target_network_update_period=50, rng_key=3)
  if kind == 'dqn':
    from dqn_zoo_amd.dqn import agent as m
    ag = m.Dqn(network=networks.DenseNetwork('dqn', A),
               optimizer=learner.RmsPropConfig(learning_rate=0.0005, decay=0.95, eps=1e-5),
               transition_accumulator=rl.TransitionAccumulator(),
               replay=rl.TransitionReplay(2000, T, rs), exploration_epsilon=eps,
               grad_error_bound=1.0 / 32, **common)
  else:
    from dqn_zoo_amd.rainbow import agent as m
    support = np.linspace(-2.0, 2.0, 51).astype(np.float32)
    rep = rl.PrioritizedTransitionReplay(
        2000, T, 0.5, parts.LinearSchedule(begin_t=0, end_t=2000, begin_value=0.4,
                                           end_value=1.0), 1e-3, True, rs)
    ag = m.Rainbow(network=networks.RainbowNetwork(A, support, 0.1), support=support,
                   optimizer=learner.AdamConfig(learning_rate=0.0005, eps=0.005 / 32),
                   transition_accumulator=rl.NStepTransitionAccumulator(1), replay=rep,
                   **common)
  env = ContextBandit(1)
  q0 = _q(ag, env, kind)
  # every episode is FIRST + LAST = 2 agent steps
  for _ in itertools.islice(parts.run_loop(ag, env, max_steps_per_episode=0), 1600):
    pass
  torch.cuda.synchronize()
  q1 = _q(ag, env, kind)
  assert np.isfinite(q1).all()
  for c in range(2):
    assert int(np.argmax(q1[c])) == c, (kind, c, q0, q1)
    wrong = np.delete(q1[c], c)
    assert q1[c, c] - wrong.max() > 0.3, (kind, c, q1)   # reward gap is 1.0
  if kind == 'rainbow':
    ag._replay.check_status()  # pylint: disable=protected-access
