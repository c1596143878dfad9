"""world_size-2 gloo test of the only collective on the path: the packed
statistics all-reduce.  Runs on CPU."""

import os
import socket
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
  s = socket.socket()
  s.bind(('127.0.0.1', 0))
  p = s.getsockname()[1]
  s.close()
  return p


def _worker(rank, world, port, q):
  sys.path.insert(0, ROOT)
  os.environ['MASTER_ADDR'] = '127.0.0.1'
  os.environ['MASTER_PORT'] = str(port)
  dist.init_process_group('gloo', rank=rank, world_size=world)
  from dqn_zoo_amd import distributed as dz
  st = dz.ReplicaStats('cpu')
  # replica r: (r+1) episodes of return 10*(r+1), 100*(r+1) steps, duration r+1
  st.add_tracker_statistics({
      'num_episodes': rank + 1, 'mean_episode_return': 10.0 * (rank + 1),
      'num_steps_over_episodes': 90 * (rank + 1),
      'num_steps_since_reset': 100 * (rank + 1), 'duration': float(rank + 1)})
  st.add(grad_steps=7, loss_sum=torch.tensor(0.5 * (rank + 1)))
  out = st.all_reduce()
  q.put((rank, dict(out)))
  dist.destroy_process_group()


def test_stats_all_reduce_world2():
  ctx = mp.get_context('spawn')
  q = ctx.Queue()
  port = _free_port()
  procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
  for p in procs:
    p.start()
  res = dict(q.get(timeout=120) for _ in range(2))
  for p in procs:
    p.join(timeout=60)
    assert p.exitcode == 0
  for r in (0, 1):
    o = res[r]
    assert o['replicas'] == 2 and o['grad_steps'] == 14
    assert o['num_episodes'] == 3 and o['episode_return_sum'] == 10 + 40
    assert abs(o['mean_episode_return'] - 50.0 / 3) < 1e-12   # exact mean
    assert o['num_steps_since_reset'] == 300 and o['duration'] == 2.0
    assert o['step_rate'] == 150.0 and o['loss_sum'] == 1.5
  assert res[0] == res[1]


def test_stats_identity_without_process_group():
  from dqn_zoo_amd import distributed as dz
  st = dz.ReplicaStats('cpu')
  st.add(grad_steps=3, duration=2.0, num_steps_since_reset=10)
  o = st.all_reduce()
  assert o['grad_steps'] == 3 and o['replicas'] == 1 and o['step_rate'] == 5.0
