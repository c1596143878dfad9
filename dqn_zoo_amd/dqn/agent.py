"""DQN agent, drop-in for `dqn_zoo/dqn/agent.py` (class Dqn, :40-229):
uniform replay, rlax.q_learning with gradient clipping, centred RMSProp."""

from dqn_zoo_amd import dense_agent


class Dqn(dense_agent.DenseAgent):
  """ref: dqn/agent.py:43-58 (constructor keywords kept)."""
  LOSS = 'q'

  def __init__(self, preprocessor, sample_network_input, network, optimizer,
               transition_accumulator, replay, batch_size, exploration_epsilon,
               min_replay_capacity_fraction, learn_period,
               target_network_update_period, grad_error_bound, rng_key):
    super().__init__(preprocessor, sample_network_input, network, optimizer,
                     transition_accumulator, replay, batch_size,
                     exploration_epsilon, min_replay_capacity_fraction,
                     learn_period, target_network_update_period, rng_key,
                     grad_error_bound=grad_error_bound)
