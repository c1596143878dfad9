"""Frames/s of the full agent loop (parts.run_loop: act -> accumulate -> add ->
learn every `learn_period` frames) for the Rainbow agent on a synthetic
pre-processed environment, with a wall-clock breakdown of agent.step()."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench


def main():
  frames = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
  which = sys.argv[2] if len(sys.argv) > 2 else 'rainbow'
  torch.cuda.set_stream(torch.cuda.Stream())
  acc = {'act': 0.0, 'add': 0.0, 'learn': 0.0}
  out = bench.measure_agent_loop(which, frames, int(os.environ.get('LEARN_PERIOD', 4)),
                                 setup=lambda ag, rep: instrument(ag, rep, which, acc),
                                 on_warm=lambda: acc.update(act=0.0, add=0.0, learn=0.0))
  dt = frames / out['agent_steps_per_sec']
  print(which + ' agent loop: %.0f agent steps/s (%.1f us/step); per step: act %.1f us, add %.1f us, '
        'learn(enqueue, every 4th) %.1f us, other %.1f us' % (
            frames / dt, 1e6 * dt / frames, 1e6 * acc['act'] / frames, 1e6 * acc['add'] / frames,
            1e6 * acc['learn'] / frames, 1e6 * (dt - sum(acc.values())) / frames))
  if 'json' in sys.argv[3:]:
    import json
    print(json.dumps(out))


def instrument(ag, rep, which, acc):
  add_name = 'add_with_device_priority' if which == 'rainbow' else 'add'
  if 'graph-learn' in sys.argv[3:]:
    ag._learner.use_graphs = True   # learner step replayed from a hipGraph (the agents' default is eager)  # pylint: disable=protected-access
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
  else:   # Rainbow: the acting apply is enqueued inside step() 
    wrap(ag._learner, 'apply_async', 'act')   # pylint: disable=protected-access
  wrap(ag, '_learn', 'learn'); wrap(rep, add_name, 'add')
  if which == 'iqn':
    ag.act_one_launch = 'multi-launch-act' not in sys.argv[3:]
  if which == 'rainbow':   # A/B switches of the acting path (defaults: both on)
    ag._learner.poll_action_slot = 'event-wait' not in sys.argv[3:]   # pylint: disable=protected-access
    ag._learner.act_direct = 'act-graph' not in sys.argv[3:]          # pylint: disable=protected-access


if __name__ == '__main__':
  main()
