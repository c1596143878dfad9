"""Shared body of the dense-head agents (DQN, double-Q, prioritized, C51,
QR-DQN).  The five agent modules (`dqn_zoo_amd/<agent>/agent.py`) are thin
subclasses that keep the reference constructors' keyword names; everything
that is common to the reference's five `agent.py` files lives here:

  step / reset / _act / get_state / set_state   ref: dqn/agent.py:133-229
  epsilon-greedy behaviour policy                ref: dqn/agent.py:121-131
                                                 (distrax.EpsilonGreedy)
  learning gates (min replay, learn_period, target sync)  ref: dqn/agent.py:149-156

`_learn()` enqueues sample -> update (-> priority write-back) on the device with
no host synchronisation.
"""

from typing import Any, Mapping

import numpy as np

from dqn_zoo_amd import device_obs
from dqn_zoo_amd import learner as learner_lib
from dqn_zoo_amd import networks
from dqn_zoo_amd import parts
from dqn_zoo_amd import processors


def epsilon_greedy_sample(q_values: np.ndarray, epsilon: float,
                          random_state: np.random.RandomState) -> int:
  """distrax.EpsilonGreedy(q, eps).sample(): probs = eps/A + (1-eps) *
  1[q == max] / #argmax.  The JAX key stream itself
  cannot be reproduced; the distribution is the same."""
  q = np.asarray(q_values, dtype=np.float64)
  a = q.shape[0]
  greedy = (q == q.max())
  probs = epsilon / a + (1.0 - epsilon) * greedy / greedy.sum()
  # RandomState.choice(a, p=p) for one draw, without its argument checks (8 of the 15 us this
  # function cost per frame): one uniform, normalised cumulative sum, searchsorted -- the same
  # arithmetic and the same use of the stream, so the same actions for the same seed
  cdf = (probs / probs.sum()).cumsum()
  cdf /= cdf[-1]
  return int(cdf.searchsorted(random_state.random_sample(), side='right'))


class DenseAgent(parts.Agent):
  """Common machinery; subclasses fix the loss and the replay flavour."""

  LOSS = 'q'
  PRIORITIZED = False

  def __init__(self, preprocessor, sample_network_input, network, optimizer,
               transition_accumulator, replay, batch_size, exploration_epsilon,
               min_replay_capacity_fraction, learn_period,
               target_network_update_period, rng_key, grad_error_bound=1.0 / 32,
               huber_param=1.0):
    if tuple(np.shape(sample_network_input)) != (84, 84, 4):
      raise ValueError('sample_network_input must have shape (84, 84, 4)')
    if not isinstance(network, networks.DenseNetwork):
      raise TypeError('network must be a networks.DenseNetwork descriptor')
    self._preprocessor = preprocessor
    self._replay = replay
    self._transition_accumulator = transition_accumulator
    self._batch_size = batch_size
    self._exploration_epsilon = exploration_epsilon
    self._min_replay_capacity = min_replay_capacity_fraction * replay.capacity
    self._learn_period = learn_period
    self._target_network_update_period = target_network_update_period
    self._network = network
    self._learner = learner_lib.DenseLearner(
        network, self.LOSS, optimizer, batch_size,
        grad_error_bound=grad_error_bound, huber_param=huber_param,
        seed=int(rng_key), device=replay._device)  # pylint: disable=protected-access
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

  # -- stepping ---------------------------------------------------------------
  def step(self, timestep) -> parts.Action:
    self._frame_t += 1
    timestep = self._preprocessor(timestep)
    if timestep is None:
      if self._action is None:
        raise RuntimeError('Cannot repeat if action has never been selected.')
      action = self._action
    else:
      # enqueued, not awaited (parts.PendingAction): the accumulator only stores a_t
      action = self._act(timestep)
      for transition in self._transition_accumulator.step(timestep, action):
        self._add(transition)
    if self._replay.size >= self._min_replay_capacity:
      if self._frame_t % self._learn_period == 0:
        self._learn()
        # a NaN/inf/negative priority or weight flagged by an earlier step's kernels
        # (sticky word in pinned host memory: a plain load, nothing is awaited)
        self._replay.poll_status()
      if self._frame_t % self._target_network_update_period == 0:
        self._learner.sync_target()
        # the reference raises at the offending call when a priority or weight
        # goes NaN/inf/negative (replay.py:233-242,281-282); here those land in a
        # sticky word; polled without waiting at every learner step, and definitively
        # (one host sync) once per target period
        self._replay.check_status()
    if isinstance(action, parts.PendingAction):
      # the frame's device work is queued; wait for the acting launches only
      pending, action = action, parts.Action(action.resolve())
      self._statistics['state_value'] = pending.state_value
    self._action = action
    return action

  def reset(self) -> None:
    self._transition_accumulator.reset()
    processors.reset(self._preprocessor)
    self._action = None

  def _add(self, transition) -> None:
    # both states are already in HBM (uploaded for acting): no re-upload
    transition = self._obs.on_device(transition)
    if self.PRIORITIZED:  # priority = running max (prioritized/agent.py:152-153)
      self._replay.add_with_device_priority(transition)
    else:
      self._replay.add(transition)

  def q_values(self, head_out: np.ndarray) -> np.ndarray:
    """Q-values from one state's head outputs (networks.py:305-363)."""
    net = self._network
    if net.kind == 'c51':   # expectation of softmax over atoms
      lg = head_out.reshape(net.num_actions, net.num_atoms).astype(np.float64)
      lg -= lg.max(axis=1, keepdims=True)
      p = np.exp(lg)
      p /= p.sum(axis=1, keepdims=True)
      return (p * net.support[None, :]).sum(axis=1)
    if net.kind == 'qr':    # mean over quantiles, quantile-major layout
      return head_out.reshape(net.num_atoms, net.num_actions).mean(axis=0)
    return head_out

  def _act(self, timestep) -> parts.PendingAction:
    """Epsilon-greedy action from the online network's Q-values
    (ref: dqn/agent.py:121-131, 162-170).  The apply is enqueued and its head
    outputs copied to pinned host memory asynchronously; the host part (Q-values,
    the policy's RNG draw) runs when `step()` resolves the action, after the rest
    of the frame's device work has been queued."""
    read = self._learner.head_async(self._obs.upload(timestep.observation))
    return parts.PendingAction(self._deferred_policy(read, self.exploration_epsilon))

  def _deferred_policy(self, read_head, epsilon):
    """`read_head() -> host array` of head outputs (or of Q-values, IQN); the
    policy's host part runs when the action is resolved."""

    def resolve():
      q = self.q_values(read_head())
      a_t = epsilon_greedy_sample(q, epsilon, self._policy_rng)
      return a_t, float(np.max(q))

    return resolve

  def _learn(self) -> None:
    ln = self._learner
    if self.PRIORITIZED:
      s = self._replay.sample_device(self._batch_size)
      t = s.transitions
      # priorities = |td| (prioritized/agent.py:202) go into the sum tree (and the
      # running max) inside the step's backward launches
      if self._batch_size <= 256:
        ln.step(t.s_tm1, t.a_tm1, t.r_t, t.discount_t, t.s_t, s.weights32,
                priority_sink=self._replay.priority_sink(s.ids))
      else:  # the side block handles up to 256 leaves
        ln.step(t.s_tm1, t.a_tm1, t.r_t, t.discount_t, t.s_t, s.weights32)
        self._replay.update_priorities(s.ids, ln.priorities)
    else:
      t, _ = self._replay.sample_device(self._batch_size)
      ln.step(t.s_tm1, t.a_tm1, t.r_t, t.discount_t, t.s_t, None)

  # -- properties ---------------------------------------------------------------
  @property
  def online_params(self) -> Mapping[str, np.ndarray]:
    return self._learner.get_params('online')

  @property
  def statistics(self) -> Mapping[str, float]:
    return self._statistics

  @property
  def exploration_epsilon(self) -> float:
    return self._exploration_epsilon(self._frame_t)

  @property
  def learner(self) -> learner_lib.DenseLearner:
    return self._learner

  def get_state(self) -> Mapping[str, Any]:
    state = {
        'rng_key': self._policy_rng.get_state(),
        'frame_t': self._frame_t,
        'opt_state': self._learner.get_opt_state(),
        'online_params': self._learner.get_params('online'),
        'target_params': self._learner.get_params('target'),
        'replay': self._replay.get_state(),
    }
    if self.PRIORITIZED:
      state['max_seen_priority'] = self.max_seen_priority
    return state

  def set_state(self, state: Mapping[str, Any]) -> None:
    self._policy_rng.set_state(state['rng_key'])
    self._frame_t = state['frame_t']
    self._learner.set_opt_state(state['opt_state'])
    self._learner.set_params(state['online_params'], 'online')
    self._learner.set_params(state['target_params'], 'target')
    self._replay.set_state(state['replay'])
    if self.PRIORITIZED:
      self._replay.max_seen_priority_device.fill_(state['max_seen_priority'])
