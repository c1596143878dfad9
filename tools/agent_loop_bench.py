"""Frames/s of the full agent loop (parts.run_loop: act -> accumulate -> add ->
learn every `learn_period` frames) for the Rainbow agent on a synthetic
pre-processed environment, with a wall-clock breakdown of agent.step()."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from dqn_zoo_amd import learner, networks, parts, processors
from dqn_zoo_amd import replay as replay_lib
from dqn_zoo_amd.rainbow import agent as agent_lib
from dqn_zoo_amd import dm_env_shim as dm_env

A = 6
SUPPORT = np.linspace(-10.0, 10.0, 51).astype(np.float32)


class Env:
  def __init__(self, seed, n=1000):
    rs = np.random.RandomState(seed)
    self.pool = rs.randint(0, 256, (64, 84, 84, 4)).astype(np.uint8)
    self.rs, self.n = rs, n
  def _obs(self):
    return self.pool[self.rs.randint(64)]
  def reset(self):
    self.t = 0
    return dm_env.restart(self._obs())
  def step(self, action):
    self.t += 1
    r = float(self.rs.randint(-1, 2))
    if self.t == self.n:
      return dm_env.termination(r, self._obs())
    return dm_env.transition(r, self._obs(), 0.99)


def main():
  frames = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
  which = sys.argv[2] if len(sys.argv) > 2 else 'rainbow'
  torch.cuda.set_stream(torch.cuda.Stream())
  if which == 'dqn':
    from dqn_zoo_amd.dqn import agent as dqn_lib
    rep = replay_lib.TransitionReplay(
        100000, replay_lib.Transition(None, None, None, None, None), np.random.RandomState(1))
    ag = dqn_lib.Dqn(
        preprocessor=processors.Identity(), sample_network_input=np.zeros((84, 84, 4), np.uint8),
        network=networks.DenseNetwork('dqn', A), optimizer=learner.RmsPropConfig(),
        transition_accumulator=replay_lib.TransitionAccumulator(), replay=rep, batch_size=32,
        exploration_epsilon=lambda t: 0.1, min_replay_capacity_fraction=0.005, learn_period=4,
        target_network_update_period=2000, rng_key=1, grad_error_bound=1.0 / 32)
    add_name = 'add'
  else:
    add_name = 'add_with_device_priority'
    rep = replay_lib.PrioritizedTransitionReplay(
        100000, replay_lib.Transition(None, None, None, None, None), 0.5,
        parts.LinearSchedule(begin_t=2000, end_t=10 ** 7, begin_value=0.4, end_value=1.0),
        1e-3, True, np.random.RandomState(1))
  if which != 'dqn':
   ag = agent_lib.Rainbow(
      preprocessor=processors.Identity(),
      sample_network_input=np.zeros((84, 84, 4), np.uint8),
      network=networks.RainbowNetwork(A, SUPPORT, 0.1), support=SUPPORT,
      optimizer=learner.AdamConfig(),
      transition_accumulator=replay_lib.NStepTransitionAccumulator(3), replay=rep,
      batch_size=32, min_replay_capacity_fraction=0.005,
      learn_period=int(os.environ.get('LEARN_PERIOD', 4)),
      target_network_update_period=2000, rng_key=1)
  if 'eager-learn' in sys.argv[3:]:
    ag._learner.use_graphs = False   # learner launches eager, acting applies still from graphs  # pylint: disable=protected-access
  # instrument
  acc = {'act': 0.0, 'add': 0.0, 'learn': 0.0}
  def wrap(obj, name, key):
    f = getattr(obj, name)
    def g(*a, **k):
      t0 = time.perf_counter()
      r = f(*a, **k)
      acc[key] += time.perf_counter() - t0
      return r
    setattr(obj, name, g)
  if hasattr(ag, '_act'):
    wrap(ag, '_act', 'act')
  else:   # Rainbow: the acting apply is enqueued inside step() (it may carry the sample)
    wrap(ag._learner, 'apply_async', 'act')   # pylint: disable=protected-access
  wrap(ag, '_learn', 'learn'); wrap(rep, add_name, 'add')
  if 'nofuse' in sys.argv[3:]:
    ag.fuse_sample_into_acting = False
  if 'fuse' in sys.argv[3:]:
    ag.fuse_sample_into_acting = True
  if which != 'dqn':   # A/B switches of the acting path (defaults: both on)
    ag._learner.poll_action_slot = 'event-wait' not in sys.argv[3:]   # pylint: disable=protected-access
    ag._learner.act_direct = 'act-graph' not in sys.argv[3:]          # pylint: disable=protected-access
  env = Env(3)
  loop = parts.run_loop(ag, env, max_steps_per_episode=0)
  for _ in range(1000):   # fill past min replay, warm up
    next(loop)
  torch.cuda.synchronize()
  for k in acc: acc[k] = 0.0
  t0 = time.perf_counter()
  for _ in range(frames):
    next(loop)
  torch.cuda.synchronize()
  dt = time.perf_counter() - t0
  print(which + ' agent loop: %.0f agent steps/s (%.1f us/step); per step: act %.1f us, add %.1f us, '
        'learn(enqueue, every 4th) %.1f us, other %.1f us' % (
            frames / dt, 1e6 * dt / frames, 1e6 * acc['act'] / frames, 1e6 * acc['add'] / frames,
            1e6 * acc['learn'] / frames,
            1e6 * (dt - sum(acc.values())) / frames))
  rep.check_status()
  if 'json' in sys.argv[3:]:
    import json
    print(json.dumps({'agent': which, 'agent_steps_per_sec': round(frames / dt, 1),
                      'us_per_agent_step': round(1e6 * dt / frames, 2), 'learn_period': 4,
                      'learner_steps_per_sec': round(frames / dt / 4, 1),
                      'sample_carried_by_acting_apply': bool(getattr(ag, 'fuse_sample_into_acting', False))}))


if __name__ == '__main__':
  main()
