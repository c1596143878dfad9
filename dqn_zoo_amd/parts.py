"""Host-side loop and statistics glue with the surface of dqn_zoo's `parts.py`.

Out of scope as compute (pure Python in the reference too), but the
agents are drop-ins only if this surface is kept: `Agent`, `run_loop`
(ref: parts.py:70-122), `generate_statistics` (125-147), the three trackers
(150-329), `make_default_trackers` (332-339), `LinearSchedule` (414-430),
`NullWriter`/`CsvWriter` (433-493), `NullCheckpoint` (496-513),
`AttributeDict` (516-526).  Behaviour is pinned by tests/test_parts.py, which
replays the bodies of the reference's parts_test.py.
"""

import abc
import collections
import csv
import os
import timeit
from typing import Any, Iterable, Mapping, Optional, Sequence, Tuple

import numpy as np

from dqn_zoo_amd import dm_env_shim as dm_env

Action = int


class PendingAction:
  """An action (and its value estimate) that the GPU is still computing.

  `agent.step()` enqueues the acting network's launches, then the replay insert
  and -- every `learn_period` frames -- the learner step, and only THEN reads the
  action back: none of that later work needs the new action's value (the
  transition accumulators only store a_t for transitions they emit on later
  steps, ref: replay.py:771-892), so the ~50 us of host bookkeeping per frame and
  the device work overlap instead of alternating.  The object stands in for the
  int wherever the action is merely carried (`int(x)`, `np.int64(x)`, indexing and
  `==` resolve it; resolving waits for the acting launches only, not for the
  learner step queued behind them)."""

  __slots__ = ('_read', '_value', 'state_value')

  def __init__(self, read):
    self._read = read          # () -> (int action, float value); synchronises
    self._value = None
    self.state_value = None

  def resolve(self) -> int:
    if self._value is None:
      a, v = self._read()
      self._value, self.state_value, self._read = int(a), v, None
    return self._value

  __int__ = __index__ = resolve

  def __eq__(self, other):
    return self.resolve() == other

  def __hash__(self):
    return hash(self.resolve())

  def __repr__(self):
    return 'PendingAction(%s)' % ('?' if self._value is None else self._value)


class Agent(abc.ABC):
  """Agent interface (ref: parts.py:42-67)."""

  @abc.abstractmethod
  def step(self, timestep) -> Action:
    """Selects action given timestep and potentially learns."""

  @abc.abstractmethod
  def reset(self) -> None:
    """Resets the agent's episodic state; call at the start of every episode."""

  @abc.abstractmethod
  def get_state(self) -> Mapping[str, Any]:
    """Retrieves agent state as a dictionary (e.g. for serialization)."""

  @abc.abstractmethod
  def set_state(self, state: Mapping[str, Any]) -> None:
    """Sets agent state from a (potentially de-serialized) dictionary."""

  @property
  @abc.abstractmethod
  def statistics(self) -> Mapping[str, float]:
    """Returns current agent statistics as a dictionary."""


def run_loop(agent, environment, max_steps_per_episode: int = 0,
             yield_before_reset: bool = False
             ) -> Iterable[Tuple[Any, Optional[Any], Any, Optional[Action]]]:
  """Alternates environment and agent steps forever (ref: parts.py:70-122).

  Yields `(environment, timestep_t, agent, a_t)` after every `agent.step`.  An
  episode ends on a LAST timestep (or after `max_steps_per_episode` steps, by
  relabelling the timestep as LAST); the agent still sees that timestep, its
  action is dropped and `None` is yielded in its place.
  """
  while True:
    if yield_before_reset:
      yield environment, None, agent, None
    agent.reset()
    timestep = environment.reset()
    steps_taken = 0
    while True:
      action = agent.step(timestep)
      yield environment, timestep, agent, action
      steps_taken += 1
      timestep = environment.step(action)
      if 0 < max_steps_per_episode <= steps_taken:
        assert steps_taken == max_steps_per_episode
        timestep = timestep._replace(step_type=dm_env.StepType.LAST)
      if timestep.last():
        agent.step(timestep)  # the agent observes the end; action unused.
        yield environment, timestep, agent, None
        break


def generate_statistics(trackers: Sequence[Any],
                        timestep_action_sequence) -> Mapping[str, Any]:
  """Feeds a (timestep, action) stream to trackers and merges their reports;
  earlier trackers win on duplicate keys (ref: parts.py:125-147)."""
  for tr in trackers:
    tr.reset()  # once, not per episode.
  for environment, timestep_t, agent, a_t in timestep_action_sequence:
    for tr in trackers:
      tr.step(environment, timestep_t, agent, a_t)
  return dict(collections.ChainMap(*(tr.get() for tr in trackers)))


class EpisodeTracker:
  """Episode returns and step counts (ref: parts.py:150-247)."""

  def __init__(self):
    self._live = False

  def reset(self) -> None:
    self._live = True
    self._steps = 0                 # since reset
    self._steps_in_done_episodes = 0
    self._returns = []              # completed episodes
    self._rewards = []              # current episode
    self._episode_step = 0

  def step(self, environment, timestep_t, agent, a_t) -> None:
    del environment, agent, a_t
    if not self._live:
      raise RuntimeError('reset() must be called before first call to step().')
    if timestep_t.first():
      if self._rewards:
        raise ValueError('Current episode reward list should be empty.')
      if self._episode_step != 0:
        raise ValueError('Current episode step should be zero.')
    else:
      self._rewards.append(timestep_t.reward)  # a FIRST reward is meaningless.
    self._steps += 1
    self._episode_step += 1
    if timestep_t.last():
      self._returns.append(sum(self._rewards))
      self._steps_in_done_episodes += self._episode_step
      self._rewards, self._episode_step = [], 0

  def get(self) -> Mapping[str, Any]:
    """`episode_return` is the mean over completed episodes if there is one,
    else the running return of the current episode, else NaN."""
    if not self._live:
      raise RuntimeError('reset() must be called before first call to get().')
    if self._returns:
      mean_return = np.array(self._returns).mean()
      current = sum(self._rewards)
      headline = mean_return
    else:
      mean_return = np.nan
      current = sum(self._rewards) if self._steps > 0 else np.nan
      headline = current
    return {
        'mean_episode_return': mean_return,
        'current_episode_return': current,
        'episode_return': headline,
        'num_episodes': len(self._returns),
        'num_steps_over_episodes': self._steps_in_done_episodes,
        'current_episode_step': self._episode_step,
        'num_steps_since_reset': self._steps,
    }


class StepRateTracker:
  """Steps per second since reset (ref: parts.py:250-284)."""

  def __init__(self):
    self._steps = None
    self._t0 = None

  def step(self, environment, timestep_t, agent, a_t) -> None:
    del environment, timestep_t, agent, a_t
    self._steps += 1

  def reset(self) -> None:
    self._steps = 0
    self._t0 = timeit.default_timer()

  def get(self) -> Mapping[str, float]:
    if self._steps is None or self._t0 is None:
      raise RuntimeError('reset() must be called before first call to get().')
    elapsed = timeit.default_timer() - self._t0
    return {
        'step_rate': self._steps / elapsed if self._steps > 0 else np.nan,
        'num_steps': self._steps,
        'duration': elapsed,
    }


class UnbiasedExponentialWeightedAverageAgentTracker:
  """Sutton & Barto's unbiased constant-step-size average of
  `agent.statistics` (ref: parts.py:287-329)."""

  def __init__(self, step_size: float, initial_agent):
    self._initial = dict(initial_agent.statistics)
    self._alpha = step_size
    self.trace = 0.0
    self._stats = dict(self._initial)

  def step(self, environment, timestep_t, agent, a_t) -> None:
    del environment, timestep_t, a_t
    self.trace = (1 - self._alpha) * self.trace + self._alpha
    beta = self._alpha / self.trace
    assert 0 <= beta <= 1
    fresh = agent.statistics
    if beta == 1:  # first step: initial values are typically NaN.
      self._stats = dict(fresh)
    else:
      self._stats = {k: (1 - beta) * self._stats[k] + beta * fresh[k]
                     for k in self._stats}

  def reset(self) -> None:
    self.trace = 0.0
    self._stats = dict(self._initial)

  def get(self) -> Mapping[str, float]:
    return self._stats


def make_default_trackers(initial_agent) -> Sequence[Any]:
  return [
      EpisodeTracker(),
      StepRateTracker(),
      UnbiasedExponentialWeightedAverageAgentTracker(
          step_size=1e-3, initial_agent=initial_agent),
  ]


class LinearSchedule:
  """Linear interpolation begin_value -> end_value over [begin_t, end_t],
  constant outside (ref: parts.py:414-430)."""

  def __init__(self, begin_value, end_value, begin_t, end_t=None,
               decay_steps=None):
    if (end_t is None) == (decay_steps is None):
      raise ValueError('Exactly one of end_t, decay_steps must be provided.')
    self._decay_steps = decay_steps if end_t is None else end_t - begin_t
    self._begin_t = begin_t
    self._begin_value = begin_value
    self._end_value = end_value

  def __call__(self, t):
    frac = min(max(t - self._begin_t, 0), self._decay_steps) / self._decay_steps
    return (1 - frac) * self._begin_value + frac * self._end_value


class NullWriter:
  """Logging placeholder (ref: parts.py:433-440)."""

  def write(self, *args, **kwargs) -> None:
    pass

  def close(self) -> None:
    pass


class CsvWriter:
  """Appends one CSV row per `write(OrderedDict)`; the first call fixes the
  columns and writes the header (ref: parts.py:443-493)."""

  def __init__(self, fname: str):
    folder = os.path.dirname(fname)
    if not os.path.exists(folder):
      os.makedirs(folder)
    self._fname = fname
    self._header_written = False
    self._fieldnames = None

  def write(self, values) -> None:
    if self._fieldnames is None:
      self._fieldnames = values.keys()
    # append mode: logging continues in the same file after a restart.
    with open(self._fname, 'a') as f:
      w = csv.DictWriter(f, fieldnames=self._fieldnames)  # checks the keys.
      if not self._header_written:
        w.writeheader()
        self._header_written = True
      w.writerow(values)

  def close(self) -> None:
    pass

  def get_state(self) -> Mapping[str, Any]:
    return {'header_written': self._header_written,
            'fieldnames': self._fieldnames}

  def set_state(self, state: Mapping[str, Any]) -> None:
    self._header_written = state['header_written']
    self._fieldnames = state['fieldnames']


class AttributeDict(dict):
  """dict with attribute access (ref: parts.py:516-526)."""

  def __getattr__(self, key):
    return self[key]

  def __setattr__(self, key, value):
    self[key] = value

  def __delattr__(self, key):
    del self[key]


class NullCheckpoint:
  """Checkpoint placeholder holding state in memory (ref: parts.py:496-513)."""

  def __init__(self):
    self.state = AttributeDict()

  def save(self) -> None:
    pass

  def can_be_restored(self) -> bool:
    return False

  def restore(self) -> None:
    pass


class EpsilonGreedyActor(Agent):
  """Agent that acts epsilon-greedily with externally set network parameters
  (the evaluation actor; ref: parts.py:342-411).  `network` is a descriptor
  (`networks.RainbowNetwork` / `networks.DenseNetwork`), `rng_key` an int seed;
  assign `network_params = train_agent.online_params` before stepping
  (rainbow/run_atari.py:284-288)."""

  def __init__(self, preprocessor, network, exploration_epsilon: float,
               rng_key: int):
    import torch  # pylint: disable=import-outside-toplevel
    from dqn_zoo_amd import learner as learner_lib  # pylint: disable=import-outside-toplevel
    self._preprocessor = preprocessor
    self._epsilon = exploration_epsilon
    self._rng = np.random.RandomState(int(rng_key) % (2 ** 32))
    self._net = learner_lib.InferenceNet(network, seed=int(rng_key))
    self._obs = torch.empty((1, 84, 84, 4), dtype=torch.uint8,
                            device=self._net.device)
    self._torch = torch
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

  def step(self, timestep) -> Action:
    timestep = self._preprocessor(timestep)
    if timestep is None:  # repeat action
      if self._action is None:
        raise RuntimeError('Cannot repeat if action has never been selected.')
      return self._action
    obs = np.ascontiguousarray(timestep.observation, dtype=np.uint8)
    self._obs[0].copy_(self._torch.from_numpy(obs))
    q = np.asarray(self._net.q_values(self._obs), dtype=np.float64)
    greedy = (q == q.max())
    probs = self._epsilon / len(q) + (1.0 - self._epsilon) * greedy / greedy.sum()
    self._action = Action(self._rng.choice(len(q), p=probs / probs.sum()))
    return self._action

  def reset(self) -> None:
    from dqn_zoo_amd import processors  # pylint: disable=import-outside-toplevel
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
