"""Stress of the one-launch decisions: many decisions interleaved with learner steps and inserts (the
agent loops), checking every returned action / q-value for NaN and the seam words at the end."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np, torch
import agent_loop_bench as b
from dqn_zoo_amd import parts

def run(which, frames):
  sys.argv = ['x', str(frames), which]
  torch.cuda.set_stream(torch.cuda.Stream())
  # build the same agents as the bench, but iterate ourselves and check every action
  import types
  out = {}
  orig = parts.run_loop
  def checked(agent, env, max_steps_per_episode=0):
    for item in orig(agent, env, max_steps_per_episode):
      a = item[3]
      v = agent.statistics.get('state_value', 0.0)
      if a is not None and not (0 <= int(a) < b.A):
        raise SystemExit('bad action %r' % (a,))
      if v != v:
        raise SystemExit('NaN state value at frame')
      out['n'] = out.get('n', 0) + 1
      yield item
  parts.run_loop = checked
  try:
    b.main()
  finally:
    parts.run_loop = orig
  return out['n']

for which in ('rainbow', 'dqn'):
  t0 = time.time()
  n = run(which, int(sys.argv[1]) if len(sys.argv) > 1 else 100000)
  print(which, 'frames checked', n, 'in %.1f s' % (time.time() - t0))
