"""Double-DQN agent, drop-in for `dqn_zoo/double_q/agent.py` (class DoubleDqn):
uniform replay, rlax.double_q_learning (online(s_t) selects, target evaluates;
double_q/agent.py:85-111), shared-bias head, centred RMSProp."""

from dqn_zoo_amd import dense_agent


class DoubleDqn(dense_agent.DenseAgent):
  LOSS = 'double_q'

  def __init__(self, preprocessor, sample_network_input, network, optimizer,
               transition_accumulator, replay, batch_size, exploration_epsilon,
               min_replay_capacity_fraction, learn_period,
               target_network_update_period, grad_error_bound, rng_key):
    super().__init__(preprocessor, sample_network_input, network, optimizer,
                     transition_accumulator, replay, batch_size,
                     exploration_epsilon, min_replay_capacity_fraction,
                     learn_period, target_network_update_period, rng_key,
                     grad_error_bound=grad_error_bound)
