"""Drop-in behaviour of the five dense-head agents on the GPU: run_loop smoke
mirroring the reference's `<agent>/run_atari_test.py` (replay 1000, batch 10,
learn_period 2), learning gates, epsilon-greedy policy, evaluation actor."""

import itertools

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

A = 4
SUPPORT = np.linspace(-10.0, 10.0, 51).astype(np.float32)
QUANTILES = ((np.arange(201) + 0.5) / 201).astype(np.float32)


def _env(seed):
  from tests.test_agent_gpu import SyntheticEnv
  return SyntheticEnv(seed)


def _build(name, seed=1):
  from dqn_zoo_amd import learner, networks, parts, processors
  from dqn_zoo_amd import replay as rl
  T = rl.Transition(None, None, None, None, None)
  rs = np.random.RandomState(seed)
  common = dict(preprocessor=processors.Identity(),
                sample_network_input=np.zeros((84, 84, 4), np.uint8),
                transition_accumulator=rl.TransitionAccumulator(), batch_size=10,
                exploration_epsilon=parts.LinearSchedule(
                    begin_t=50, decay_steps=50, begin_value=1.0, end_value=0.1),
                min_replay_capacity_fraction=0.05, learn_period=2,
                target_network_update_period=40, rng_key=seed)
  rms = learner.RmsPropConfig()
  adam = learner.AdamConfig(learning_rate=0.00025, eps=0.01 / 32)
  if name == 'dqn':
    from dqn_zoo_amd.dqn import agent as m
    net = networks.DenseNetwork('dqn', A)
    return m.Dqn(network=net, optimizer=rms, replay=rl.TransitionReplay(1000, T, rs),
                 grad_error_bound=1 / 32, **common), net
  if name == 'double_q':
    from dqn_zoo_amd.double_q import agent as m
    net = networks.DenseNetwork('double_dqn', A)
    return m.DoubleDqn(network=net, optimizer=rms,
                       replay=rl.TransitionReplay(1000, T, rs),
                       grad_error_bound=1 / 32, **common), net
  if name == 'prioritized':
    from dqn_zoo_amd.prioritized import agent as m
    net = networks.DenseNetwork('double_dqn', A)
    rep = rl.PrioritizedTransitionReplay(
        1000, T, 0.6, parts.LinearSchedule(begin_t=50, end_t=500, begin_value=0.4,
                                           end_value=1.0), 1e-3, True, rs)
    return m.PrioritizedDqn(network=net, optimizer=rms, replay=rep,
                            grad_error_bound=1 / 32, **common), net
  if name == 'c51':
    from dqn_zoo_amd.c51 import agent as m
    net = networks.DenseNetwork('c51', A, support=SUPPORT)
    return m.C51(network=net, support=SUPPORT, optimizer=adam,
                 replay=rl.TransitionReplay(1000, T, rs), **common), net
  from dqn_zoo_amd.qrdqn import agent as m
  net = networks.DenseNetwork('qr', A, quantiles=QUANTILES)
  return m.QrDqn(network=net, quantiles=QUANTILES, optimizer=adam,
                 replay=rl.TransitionReplay(1000, T, rs), huber_param=1.0,
                 **common), net


@pytest.mark.parametrize('name', ['dqn', 'double_q', 'prioritized', 'c51', 'qrdqn'])
def test_agent_run_loop_smoke(name):
  from dqn_zoo_amd import parts, processors
  ag, net = _build(name)
  p0 = ag.online_params
  assert ag.exploration_epsilon == 1.0
  seq = itertools.islice(parts.run_loop(ag, _env(3), max_steps_per_episode=50), 130)
  stats = parts.generate_statistics(parts.make_default_trackers(ag), seq)
  assert stats['num_steps_since_reset'] == 130 and np.isfinite(stats['state_value'])
  assert abs(ag.exploration_epsilon - 0.1) < 1e-12      # schedule finished
  p1 = ag.online_params
  assert any(np.abs(p1[k] - p0[k]).max() > 0 for k in p1)
  assert all(np.isfinite(v).all() for v in p1.values())
  if name == 'prioritized':
    ag._replay.check_status()
    assert ag.max_seen_priority >= 1.0
    assert 0.4 <= ag.importance_sampling_exponent <= 1.0
    assert ag._replay.check_valid()[0]
  # state round trip through get_state/set_state
  st = ag.get_state()
  ag2, _ = _build(name, seed=7)
  ag2.set_state(st)
  for k, v in ag2.online_params.items():
    np.testing.assert_array_equal(v, p1[k])
  # evaluation actor with externally set parameters (parts.py:342-411)
  actor = parts.EpsilonGreedyActor(processors.Identity(), net, 0.0, rng_key=5)
  with pytest.raises(RuntimeError):
    actor.step(_env(0).reset())
  actor.network_params = p1
  env = _env(9)
  ts = env.reset()
  acts = [actor.step(ts) for _ in range(3)]
  assert all(0 <= a < A for a in acts) and len(set(acts)) == 1  # greedy, same obs


def test_epsilon_greedy_distribution():
  from dqn_zoo_amd import dense_agent
  rs = np.random.RandomState(0)
  q = np.array([1.0, 3.0, 3.0, 0.0])
  draws = [dense_agent.epsilon_greedy_sample(q, 0.2, rs) for _ in range(20000)]
  freq = np.bincount(draws, minlength=4) / 20000.0
  np.testing.assert_allclose(freq, [0.05, 0.45, 0.45, 0.05], atol=0.01)
  assert dense_agent.epsilon_greedy_sample(q, 0.0, rs) in (1, 2)


def test_rainbow_eval_actor():
  from dqn_zoo_amd import networks, parts, processors
  from tests.test_agent_gpu import _make_agent
  ag, _ = _make_agent()
  net = networks.RainbowNetwork(A, SUPPORT, 0.1)
  actor = parts.EpsilonGreedyActor(processors.Identity(), net, 0.0, rng_key=3)
  actor.network_params = ag.online_params
  ts = _env(1).reset()
  assert 0 <= actor.step(ts) < A
  st = actor.get_state()
  actor.set_state(st)
  actor.reset()
