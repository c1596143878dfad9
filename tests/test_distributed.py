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


def _replica_worker(rank, world, port, q):
  """What one bench.py replica does around its timed region, on CPU: seed derivation,
  host-core slice, its own replay id stream, then bench's statistics reduction."""
  sys.path.insert(0, ROOT)
  os.environ['MASTER_ADDR'] = '127.0.0.1'
  os.environ['MASTER_PORT'] = str(port)
  import numpy as np
  dist.init_process_group('gloo', rank=rank, world_size=world)
  from dqn_zoo_amd import distributed as dz
  from oracle import replay_oracle as ro
  from tests.golden import protocol
  seed = dz.replica_seed(1, rank)
  cpus = dz.rank_cpus(rank, world, available=range(16))
  rep = ro.PrioritizedReplayOracle(64, protocol.Item(None, None), 0.5, protocol.beta_schedule(64),
                                   1e-3, True, np.random.RandomState(seed))
  for i in range(64):
    rep.add(protocol.Item(i, -i), 1.0)
  ids = np.concatenate([rep.sample_ids(32)[0] for _ in range(4)])
  steps = 20 + rank   # replicas need not finish the same number of steps
  out = dz.reduce_run(dz.ReplicaStats('cpu'), seconds=0.004 * (rank + 1), grad_steps=steps,
                      loss_sum=torch.tensor(2.0 * (rank + 1)))
  q.put((rank, seed, list(cpus), ids.tolist(), out))
  dist.destroy_process_group()


def test_two_replicas_seeds_cpu_slices_and_reduction():
  """world-size-2 gloo run of the replica plumbing bench.py uses for `--gpus N`
  (VERDICT r2 next #7): replicas draw different id streams from seeds seed+1000*rank,
  get disjoint host-core slices, and bench's `reduce_run` reports replicas == 2, the
  SUM of steps and the MAX of the two durations -- identically on both ranks."""
  ctx = mp.get_context('spawn')
  q = ctx.Queue()
  port = _free_port()
  procs = [ctx.Process(target=_replica_worker, args=(r, 2, port, q)) for r in range(2)]
  for p in procs:
    p.start()
  res = {r[0]: r[1:] for r in (q.get(timeout=120) for _ in range(2))}
  for p in procs:
    p.join(timeout=60)
    assert p.exitcode == 0
  (seed0, cpus0, ids0, out0), (seed1, cpus1, ids1, out1) = res[0], res[1]
  assert (seed0, seed1) == (1, 1001)
  assert cpus0 == list(range(0, 8)) and cpus1 == list(range(8, 16))
  assert ids0 != ids1 and len(ids0) == len(ids1) == 128
  for o in (out0, out1):
    assert o['replicas'] == 2 and o['grad_steps'] == 41 and o['loss_sum'] == 6.0
    assert o['seconds_max'] == 0.008
    assert abs(o['steps_per_second'] - 41 / 0.008) < 1e-6
  assert out0.keys() == out1.keys()
  for k in out0:   # (means over zero episodes are NaN on both ranks)
    assert out0[k] == out1[k] or (out0[k] != out0[k] and out1[k] != out1[k]), k


def test_rank_cpus_edge_cases():
  from dqn_zoo_amd import distributed as dz
  assert list(dz.rank_cpus(0, 1, available=[3, 1, 2])) == [1, 2, 3]
  assert list(dz.rank_cpus(3, 8, available=range(128))) == list(range(48, 64))
  assert list(dz.rank_cpus(1, 4, available=[0, 1])) == [0, 1]   # fewer cores than replicas: no pinning
  slices = [set(dz.rank_cpus(r, 8, available=range(20))) for r in range(8)]
  assert all(len(s) == 2 for s in slices) and len(set.union(*slices)) == 16
