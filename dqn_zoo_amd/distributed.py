"""Multi-GPU: independent actor-learner replicas, one process per GPU, and ONE
collective -- an all-reduce of packed statistics sums over RCCL/xGMI.

The reference is single-process (README.md:93-95) and has no collectives
(SURVEY.md 2, 5).  The learner path does not shard: batch 32 is one sequential
learner coupled to its own replay ("replicas only"), so there is
no gradient or replay exchange.  What is reduced is what the reference's
trackers report per process (keys of EpisodeTracker.get / StepRateTracker.get,
ref: parts.py:239-247, 280-284): SUMS and COUNTS, so that cross-replica means
are exact, plus MAX of durations.

Works with any initialised torch.distributed backend ("nccl" == RCCL on ROCm;
"gloo" for the CPU tests) and degrades to the identity when uninitialised.
"""

import collections
import os
from typing import Mapping, Optional, Sequence

import torch
import torch.distributed as dist

SUM_KEYS = ('episode_return_sum', 'num_episodes', 'num_steps_over_episodes',
            'num_steps_since_reset', 'state_value_sum', 'grad_steps',
            'loss_sum', 'replicas')
MAX_KEYS = ('duration',)


def world_size() -> int:
  return dist.get_world_size() if dist.is_available() and dist.is_initialized() \
      else 1


class ReplicaStats:
  """Accumulates per-replica sums in one float64 device vector and reduces
  them with a single all-reduce(SUM) (+ one all-reduce(MAX) for durations)."""

  def __init__(self, device='cpu'):
    self._device = torch.device(device)
    self._sum = torch.zeros(len(SUM_KEYS), dtype=torch.float64,
                            device=self._device)
    self._max = torch.zeros(len(MAX_KEYS), dtype=torch.float64,
                            device=self._device)
    self._sum[SUM_KEYS.index('replicas')] = 1.0

  def add(self, **values) -> None:
    """Adds to the named sums / maxes; values may be floats or 0-d tensors
    (device tensors are consumed without a host sync)."""
    for k, v in values.items():
      v = v.to(self._device, torch.float64) if isinstance(v, torch.Tensor) \
          else float(v)
      if k in SUM_KEYS:
        self._sum[SUM_KEYS.index(k)] += v
      elif k in MAX_KEYS:
        i = MAX_KEYS.index(k)
        self._max[i] = torch.maximum(self._max[i], torch.as_tensor(
            v, dtype=torch.float64, device=self._device))
      else:
        raise KeyError(k)

  def add_tracker_statistics(self, stats: Mapping[str, float]) -> None:
    """Folds in one replica's `generate_statistics` output (parts.py:125-147)."""
    n = stats.get('num_episodes', 0)
    mean_ret = stats.get('mean_episode_return', float('nan'))
    self.add(num_episodes=n,
             episode_return_sum=(mean_ret * n) if n else 0.0,
             num_steps_over_episodes=stats.get('num_steps_over_episodes', 0),
             num_steps_since_reset=stats.get('num_steps_since_reset', 0),
             duration=stats.get('duration', 0.0))

  def all_reduce(self) -> Mapping[str, float]:
    """One SUM all-reduce (and one MAX) across replicas; returns host floats
    including the derived cross-replica means."""
    if world_size() > 1:
      dist.all_reduce(self._sum, op=dist.ReduceOp.SUM)
      dist.all_reduce(self._max, op=dist.ReduceOp.MAX)
    s = self._sum.cpu().tolist()
    m = self._max.cpu().tolist()
    out = collections.OrderedDict(zip(SUM_KEYS, s))
    out.update(zip(MAX_KEYS, m))
    ne = out['num_episodes']
    out['mean_episode_return'] = out['episode_return_sum'] / ne if ne else float('nan')
    out['step_rate'] = (out['num_steps_since_reset'] / out['duration']
                        if out['duration'] > 0 else float('nan'))
    return out


# ---- replica set-up (one process per GPU) ------------------------------------------
def replica_seed(seed: int, rank: int) -> int:
  """Seed of replica `rank`: replicas are independent runs (different replay RNG,
  network init and noise streams), not shards of one run."""
  return int(seed) + 1000 * int(rank)


def rank_cpus(local_rank: int, local_world: int,
              available: Optional[Sequence[int]] = None) -> Sequence[int]:
  """Host cores for one of `local_world` replicas: a contiguous, disjoint slice of the
  cores this process may use (sorted ids: on a two-socket MI355X node the lower half
  of the ids -- and GPUs 0-3 -- sit on socket 0, so contiguous slices keep a replica's
  host thread on the socket its GPU hangs off).  Every replica's host thread draws
  RNG numbers and enqueues the step's 11 launches every ~150 us; unpinned, the 8 threads migrate
  and share cores with each other's runtime helper threads."""
  cpus = sorted(os.sched_getaffinity(0)) if available is None else sorted(available)
  if local_world <= 1 or len(cpus) < local_world:
    return cpus
  per = len(cpus) // local_world
  return cpus[local_rank * per:(local_rank + 1) * per]


def pin_rank(local_rank: int, local_world: int) -> Sequence[int]:
  """Applies `rank_cpus` to this process; returns the core list (empty list: the
  platform has no affinity control and nothing was done)."""
  try:
    cpus = rank_cpus(local_rank, local_world)
    if local_world > 1 and cpus:
      os.sched_setaffinity(0, cpus)
    return list(cpus)
  except (AttributeError, OSError):
    return []


def reduce_run(stats: 'ReplicaStats', seconds: float, grad_steps: int, loss_sum,
               device='cpu') -> Mapping[str, float]:
  """The statistics boundary of a replicated run (bench.py after its timed region):
  ONE packed SUM all-reduce of the replicas' counts and sums, and the MAX over ranks of
  the elapsed time, which is what whole-job throughput divides by.  Returns the
  reduced dict plus `seconds_max` and `steps_per_second` = all replicas' steps / the
  slowest replica's time."""
  stats.add(grad_steps=grad_steps, loss_sum=loss_sum, duration=seconds)
  out = dict(stats.all_reduce())
  out['seconds_max'] = out['duration']
  out['steps_per_second'] = out['grad_steps'] / out['duration'] if out['duration'] > 0 \
      else float('nan')
  return out
