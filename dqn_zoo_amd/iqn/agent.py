"""IQN agent, drop-in for `dqn_zoo/iqn/agent.py` (classes Iqn and
IqnEpsilonGreedyActor): uniform replay, three tau draws per update
(iqn/agent.py:176-186), target network for both the greedy-action selector and
the target distribution (187-198), vmapped rlax.quantile_q_learning (199-209),
Adam without clipping (iqn/run_atari.py:213-215)."""

from typing import Any, Mapping

import numpy as np
import torch

from dqn_zoo_amd import dense_agent
from dqn_zoo_amd import device_obs
from dqn_zoo_amd import learner as learner_lib
from dqn_zoo_amd import networks
from dqn_zoo_amd import parts
from dqn_zoo_amd import processors

IqnInputs = networks.IqnInputs


class IqnEpsilonGreedyActor(parts.Agent):
  """Acts epsilon-greedily on the mean of `tau_samples` quantile samples with
  externally set network parameters (ref: iqn/agent.py:54-127)."""

  def __init__(self, preprocessor, network: networks.IqnNetwork,
               exploration_epsilon: float, tau_samples: int, rng_key: int):
    self._preprocessor = preprocessor
    self._epsilon = exploration_epsilon
    self._tau_samples = int(tau_samples)
    self._rng = np.random.RandomState(int(rng_key) % (2 ** 32))
    self._net = learner_lib.InferenceNet(network, seed=int(rng_key))
    self._obs = torch.empty((1, 84, 84, 4), dtype=torch.uint8,
                            device=self._net.device)
    self._action = None
    self._network_params = None

  @property
  def network_params(self):
    return self._network_params

  @network_params.setter
  def network_params(self, params) -> None:
    self._network_params = params
    if params is not None:
      self._net.set_params(params)

  def step(self, timestep) -> parts.Action:
    timestep = self._preprocessor(timestep)
    if timestep is None:  # repeat action
      if self._action is None:
        raise RuntimeError('Cannot repeat if action has never been selected.')
      return self._action
    obs = np.ascontiguousarray(timestep.observation, dtype=np.uint8)
    self._obs[0].copy_(torch.from_numpy(obs))
    q = self._net.iqn_q_values(self._obs, self._tau_samples)
    self._action = parts.Action(
        dense_agent.epsilon_greedy_sample(q, self._epsilon, self._rng))
    return self._action

  def reset(self) -> None:
    processors.reset(self._preprocessor)
    self._action = None

  def get_state(self) -> Mapping[str, Any]:
    return {'rng_key': self._rng.get_state(),
            'network_params': self._network_params}

  def set_state(self, state: Mapping[str, Any]) -> None:
    self._rng.set_state(state['rng_key'])
    self.network_params = state['network_params']

  @property
  def statistics(self) -> Mapping[str, float]:
    return {}


class Iqn(dense_agent.DenseAgent):
  """Implicit Quantile Network agent (ref: iqn/agent.py:130-325)."""

  def __init__(self, preprocessor, sample_network_input, network, optimizer,
               transition_accumulator, replay, batch_size, exploration_epsilon,
               min_replay_capacity_fraction, learn_period,
               target_network_update_period, huber_param, tau_samples_policy,
               tau_samples_s_tm1, tau_samples_s_t, rng_key):
    # pylint: disable=super-init-not-called
    state = getattr(sample_network_input, 'state', sample_network_input)
    if tuple(np.shape(state)) != (84, 84, 4):
      raise ValueError('sample_network_input.state must have shape (84, 84, 4)')
    if not isinstance(network, networks.IqnNetwork):
      raise TypeError('network must be a networks.IqnNetwork descriptor')
    self._preprocessor = preprocessor
    self._replay = replay
    self._transition_accumulator = transition_accumulator
    self._batch_size = batch_size
    self._exploration_epsilon = exploration_epsilon
    self._min_replay_capacity = min_replay_capacity_fraction * replay.capacity
    self._learn_period = learn_period
    self._target_network_update_period = target_network_update_period
    self._network = network
    self._tau_samples_policy = int(tau_samples_policy)
    self._learner = learner_lib.IqnLearner(
        network, optimizer, batch_size,
        tau_samples=(tau_samples_s_tm1, tau_samples_policy, tau_samples_s_t),
        huber_param=huber_param, seed=int(rng_key),
        device=replay._device)  # pylint: disable=protected-access
    # The learner step is enqueued eagerly: a frame that learns is GPU-bound (the decision kernel
    # covers the host's enqueue time) and a hipGraph replay of the same launches runs 3-6 % slower
    # on the device -- measured on the drop-in loop: Rainbow +3 %, DQN +4.5 %, IQN +2.5 % agent
    # steps/s against graph replay (EXPERIMENTS.md R6-15).  `learner.use_graphs = True` replays.
    self._learner.use_graphs = False
    self._device = self._learner.device
    self._policy_rng = np.random.RandomState(int(rng_key) % (2 ** 32))
    self._action = None
    self._frame_t = -1
    self._statistics = {'state_value': np.nan}
    self._obs = device_obs.ObservationCache(
        self._device, depth=device_obs.depth_for(transition_accumulator))
    self._act_taus = torch.empty((1, self._tau_samples_policy),
                                 dtype=torch.float32, device=self._device)
    self._act_counter = 0
    self._act_seed = (int(rng_key) * 0x9E3779B97F4A7C15 + 1) & 0xFFFFFFFFFFFFFFFF

  act_one_launch = True   # the decision for one state is ONE launch (tau_samples_policy <= 32)

  def _act(self, timestep) -> parts.PendingAction:
    """ref: iqn/agent.py:234-247 select_action: tau_samples_policy fresh draws,
    epsilon-greedy on the sample mean."""
    ln = self._learner
    obs_d = self._obs.upload(timestep.observation)
    n = self._tau_samples_policy
    if self.act_one_launch and ln.can_decide_in_one_launch(n):
      # ONE launch per decision (csrc/dz_iqn_act.h): the taus are drawn inside the kernel at the
      # same stream position dz_uniform_fill would use, the q-values polled in a pinned slot
      read = ln.q_async(obs_d, n, self._act_seed, self._act_counter)
      self._act_counter += n
      return parts.PendingAction(self._deferred_policy(read, self.exploration_epsilon))
    learner_lib._lib.check(ln._lib.dz_uniform_fill(  # pylint: disable=protected-access
        self._act_taus.data_ptr(), n, self._act_seed, self._act_counter, None,
        learner_lib._lib.stream_ptr(self._device)), 'dz_uniform_fill')
    self._act_counter += n
    _, q, _, _ = ln.apply(obs_d, self._act_taus)
    # Q-values to pinned host memory asynchronously; epsilon-greedy on the host when
    # step() resolves the action (dense_agent.DenseAgent._deferred_policy)
    if getattr(self, '_q_host', None) is None:
      self._q_host = torch.empty((8, q.shape[1]), dtype=torch.float32).pin_memory()
      self._q_events = [torch.cuda.Event() for _ in range(8)]
      self._q_pos = 0
    k = self._q_pos % 8
    self._q_pos += 1
    slot, ev = self._q_host[k], self._q_events[k]
    slot.copy_(q[0], non_blocking=True)
    ev.record(learner_lib._lib.current_stream(self._device))

    def read():
      ev.synchronize()
      return slot.numpy().copy()

    return parts.PendingAction(self._deferred_policy(read, self.exploration_epsilon))

  def q_values(self, head_out):   # the IQN apply already returns sample-mean Q-values
    return head_out

  def _learn(self) -> None:
    t, _ = self._replay.sample_device(self._batch_size)
    self._learner.step(t.s_tm1, t.a_tm1, t.r_t, t.discount_t, t.s_t)

  def get_state(self) -> Mapping[str, Any]:
    state = dict(super().get_state())
    state['act_counter'] = self._act_counter
    return state

  def set_state(self, state: Mapping[str, Any]) -> None:
    super().set_state(state)
    self._act_counter = state.get('act_counter', 0)
