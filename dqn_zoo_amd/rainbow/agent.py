"""Rainbow agent with the surface of `dqn_zoo/rainbow/agent.py`, learning on one
MI355X: prioritized replay in HBM, the whole update enqueued as HIP kernels.

Drop-in notes: the constructor keeps the reference's keyword
names; `network` is a `networks.RainbowNetwork` descriptor instead of an
`hk.Transformed`, `optimizer` a `learner.AdamConfig` instead of an
`optax.GradientTransformation`, `rng_key` an integer seed, and `replay` a
`dqn_zoo_amd.replay.PrioritizedTransitionReplay`.  Methods, properties,
statistics keys, learning gates and error behaviour follow
ref: rainbow/agent.py:41-245 line by line (cited below).

Difference that matters for speed: `_learn()` never synchronises with the
host -- sampled ids, importance weights, losses, priorities and the running
max priority all stay on the device (the reference does two host<->device
round trips per learner step, rainbow/agent.py:184-198).
"""

from typing import Any, Mapping

import numpy as np

from dqn_zoo_amd import device_obs
from dqn_zoo_amd import learner as learner_lib
from dqn_zoo_amd import networks
from dqn_zoo_amd import parts
from dqn_zoo_amd import processors
from dqn_zoo_amd import replay as replay_lib


class Rainbow(parts.Agent):
  """Rainbow agent (ref: rainbow/agent.py:41)."""

  def __init__(
      self,
      preprocessor: processors.Processor,
      sample_network_input: np.ndarray,
      network: networks.RainbowNetwork,
      support: np.ndarray,
      optimizer: learner_lib.AdamConfig,
      transition_accumulator: Any,
      replay: replay_lib.PrioritizedTransitionReplay,
      batch_size: int,
      min_replay_capacity_fraction: float,
      learn_period: int,
      target_network_update_period: int,
      rng_key: int,
  ):
    if tuple(np.shape(sample_network_input)) != (84, 84, 4):
      raise ValueError('sample_network_input must have shape (84, 84, 4)')
    if not np.array_equal(np.asarray(support, np.float32), network.support):
      raise ValueError('support differs from the network descriptor\'s')
    self._preprocessor = preprocessor
    self._replay = replay
    self._transition_accumulator = transition_accumulator
    self._batch_size = batch_size
    self._min_replay_capacity = min_replay_capacity_fraction * replay.capacity
    self._learn_period = learn_period
    self._target_network_update_period = target_network_update_period

    # parameters, target copy and optimizer state live in the learner
    # (ref: rainbow/agent.py:67-73).
    self._learner = learner_lib.RainbowLearner(
        network, optimizer, batch_size, seed=int(rng_key),
        device=replay._device)  # pylint: disable=protected-access
    # The learner step is enqueued eagerly: a frame that learns is GPU-bound (the decision kernel
    # covers the host's enqueue time) and a hipGraph replay of the same launches runs 3-6 % slower
    # on the device -- measured on the drop-in loop: Rainbow +3 %, DQN +4.5 %, IQN +2.5 % agent
    # steps/s against graph replay (EXPERIMENTS.md R6-15).  `learner.use_graphs = True` replays.
    self._learner.use_graphs = False
    self._device = self._learner.device

    self._action = None
    self._frame_t = -1
    self._statistics = {'state_value': np.nan}
    self._obs = device_obs.ObservationCache(
        self._device, depth=device_obs.depth_for(transition_accumulator))

  # -- acting / stepping -------------------------------------------------------
  def step(self, timestep) -> parts.Action:
    """Selects action given timestep and potentially learns
    (ref: rainbow/agent.py:135-160)."""
    self._frame_t += 1
    timestep = self._preprocessor(timestep)

    if timestep is None:  # repeat action
      if self._action is None:
        raise RuntimeError('Cannot repeat if action has never been selected.')
      action = self._action
    else:
      # The reference's order, act -> add -> learn (rainbow/agent.py:141-155), as ENQUEUED work:
      # the decision kernel first (greedy action w.r.t. a freshly-noised online network,
      # rainbow/agent.py:171-179: enqueued, not awaited; the (action, value) pair lands in pinned
      # host memory), then the inserts -- whose host time runs under the decision kernel --
      # then the learner step.  The accumulator only STORES a_t (parts.PendingAction) for
      # transitions it emits on later steps.
      obs_d = self._obs.upload(timestep.observation)
      action = parts.PendingAction(self._learner.apply_async(obs_d))
      for transition in self._transition_accumulator.step(timestep, action):
        # priority = running max priority, kept on the device (agent.py:149)
        # both states are already in HBM (uploaded for acting): no re-upload
        self._replay.add_with_device_priority(self._obs.on_device(transition))

    if self._replay.size >= self._min_replay_capacity:
      if self._frame_t % self._learn_period == 0:
        self._learn()
        # a NaN/inf/negative priority or weight flagged by an earlier step's kernels
        # (sticky word in pinned host memory: a plain load, nothing is awaited)
        try:
          self._replay.poll_status()
        except replay_lib.ChainTimeoutError as e:
          self._recover_from_chain_timeout(e)

      if self._frame_t % self._target_network_update_period == 0:
        self._learner.sync_target()
        # the reference raises at the offending call when a priority or weight
        # goes NaN/inf/negative (replay.py:233-242,281-282); here those land in a
        # sticky word; polled without waiting at every learner step, and definitively
        # (one host sync) once per target period
        try:
          self._replay.check_status()
          self._learner.check_status()
        except replay_lib.ChainTimeoutError as e:
          self._recover_from_chain_timeout(e)

    if isinstance(action, parts.PendingAction):
      # everything of this frame is queued; now wait for the acting launches only
      pending, action = action, parts.Action(action.resolve())
      self._statistics['state_value'] = pending.state_value
    self._action = action
    return action

  def reset(self) -> None:
    """Resets episodic state (ref: rainbow/agent.py:162-169)."""
    self._transition_accumulator.reset()
    processors.reset(self._preprocessor)
    self._action = None

  def _recover_from_chain_timeout(self, err) -> None:
    """The head launch of a learner step gave up on an in-launch seam (HIP does not promise the
    dispatch order its liveness argument uses).  The step was void on the device (no parameter,
    moment, count or priority change), so nothing is lost but that step: clear both sticky words,
    use the four-launch form of the head from now on, say so once, carry on."""
    import warnings
    try:
      self._learner.check_status(fallback=True)   # clears ws_scalars[DZ_SC_CHAIN_FAIL]
    except replay_lib.ChainTimeoutError:
      pass
    self.chain_timeouts = getattr(self, 'chain_timeouts', 0) + 1
    warnings.warn('%s -- continuing with separate launches (%d so far)' % (err, self.chain_timeouts),
                  RuntimeWarning)

  def _learn(self) -> None:
    """Samples a batch and learns from it, entirely on the device
    (ref: rainbow/agent.py:181-198)."""
    s = self._replay.sample_device(self._batch_size)
    t = s.transitions
    # priorities = clip(|losses|, 0, 100) are written by the loss kernel and go
    # straight into the sum tree (and the running max priority) inside the
    # step's backward launches: no separate update_priorities kernel.
    if self._batch_size <= 256:
      self._learner.step(t.s_tm1, t.a_tm1, t.r_t, t.discount_t, t.s_t, s.weights32,
                         priority_sink=self._replay.priority_sink(s.ids))
    else:  # the side block handles up to 256 leaves
      self._learner.step(t.s_tm1, t.a_tm1, t.r_t, t.discount_t, t.s_t, s.weights32)
      self._replay.update_priorities(s.ids, self._learner.priorities)

  # -- properties ----------------------------------------------------------------
  @property
  def online_params(self) -> Mapping[str, np.ndarray]:
    """Current Q-network parameters as Haiku-shaped host arrays."""
    return self._learner.get_params('online')

  @property
  def statistics(self) -> Mapping[str, float]:
    return self._statistics

  @property
  def importance_sampling_exponent(self) -> float:
    return self._replay.importance_sampling_exponent

  @property
  def max_seen_priority(self) -> float:
    return float(self._replay.max_seen_priority_device.item())

  @property
  def learner(self) -> learner_lib.RainbowLearner:
    return self._learner

  # -- (de)serialisation (ref: rainbow/agent.py:224-245) ---------------------------
  def get_state(self) -> Mapping[str, Any]:
    ln = self._learner
    return {
        # (learner noise seed, learner stream position, actor stream position)
        'rng_key': (ln._noise_seed, ln._noise_counter, ln.act_step()),  # pylint: disable=protected-access
        'frame_t': self._frame_t,
        'opt_state': ln.get_opt_state(),
        'online_params': ln.get_params('online'),
        'target_params': ln.get_params('target'),
        'replay': self._replay.get_state(),
        'max_seen_priority': self.max_seen_priority,
    }

  def set_state(self, state: Mapping[str, Any]) -> None:
    ln = self._learner
    ln.set_noise_state(*state['rng_key'])
    self._frame_t = state['frame_t']
    ln.set_opt_state(state['opt_state'])
    ln.set_params(state['online_params'], 'online')
    ln.set_params(state['target_params'], 'target')
    self._replay.set_state(state['replay'])
    self._replay.max_seen_priority_device.fill_(state['max_seen_priority'])
