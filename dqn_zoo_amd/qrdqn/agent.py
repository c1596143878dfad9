"""QR-DQN agent, drop-in for `dqn_zoo/qrdqn/agent.py` (class QrDqn): uniform
replay, rlax.quantile_q_learning with Huber parameter (qrdqn/agent.py:88-110),
clip_by_global_norm + Adam (qrdqn/run_atari.py:213-219)."""

import numpy as np

from dqn_zoo_amd import dense_agent


class QrDqn(dense_agent.DenseAgent):
  LOSS = 'quantile'

  def __init__(self, preprocessor, sample_network_input, network, quantiles,
               optimizer, transition_accumulator, replay, batch_size,
               exploration_epsilon, min_replay_capacity_fraction, learn_period,
               target_network_update_period, huber_param, rng_key):
    if not np.array_equal(np.asarray(quantiles, np.float32), network.quantiles):
      raise ValueError('quantiles differ from the network descriptor\'s')
    super().__init__(preprocessor, sample_network_input, network, optimizer,
                     transition_accumulator, replay, batch_size,
                     exploration_epsilon, min_replay_capacity_fraction,
                     learn_period, target_network_update_period, rng_key,
                     huber_param=huber_param)
