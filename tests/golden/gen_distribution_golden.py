"""Golden trace of the reference's GENERAL distributions (arbitrary ids, removals
in random order, capacity growth): tests/golden/dist_general_*.npz.

    python tests/golden/gen_distribution_golden.py     (needs /root/reference)

A seeded random program of add / remove / update / sample operations is run on
`dqn_zoo.replay.PrioritizedDistribution` and `UniformDistribution`; the file
records the program and, per sample, the ids and the probability bits, plus the
final tables.  `run_program` is shared with the tests, which replay the same
program on the HIP-backed classes."""

import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, '..', '..'))

from oracle import ref_loader  # noqa: E402
from tests.golden import protocol  # noqa: E402

# (name, seed, priority_exponent, usp, min_capacity, max_capacity, steps)
CASES = [('grow', 11, 0.5, 0.1, 0, None, 120), ('bounded', 12, 1.0, 0.25, 4, 24, 120),
         ('zero_tree', 13, 1.0, 0.0, 8, None, 40)]


def run_program(dist, seed, steps, max_capacity, on_sample, zero_priorities=False):
  """Random program over a PrioritizedDistribution-like `dist`."""
  rs = np.random.RandomState(seed + 500)
  live, next_id = [], 0
  for step in range(steps):
    op = rs.choice(['add', 'add', 'remove', 'update', 'sample', 'sample'])
    if op == 'add' or not live:
      n = int(rs.randint(1, 6))
      if max_capacity is not None and len(live) + n > max_capacity:
        continue
      # ids are NOT consecutive: gaps and a shuffled order
      ids = [next_id + 3 * j + int(rs.randint(0, 3)) for j in range(n)]
      next_id = max(ids) + 1 + int(rs.randint(0, 4))
      rs.shuffle(ids)
      pr = np.zeros(n) if zero_priorities else np.abs(rs.standard_cauchy(n)).clip(0, 50)
      dist.add_priorities(ids, pr)
      live.extend(ids)
    elif op == 'remove':
      n = int(rs.randint(1, min(4, len(live)) + 1))
      pick = [live[j] for j in rs.choice(len(live), size=n, replace=False)]
      dist.remove_priorities(pick)
      for i in pick:
        live.remove(i)
    elif op == 'update':
      n = int(rs.randint(1, min(5, len(live)) + 1))
      pick = [live[j] for j in rs.choice(len(live), size=n, replace=False)]
      pr = np.zeros(n) if zero_priorities else np.abs(rs.standard_cauchy(n)).clip(0, 50)
      dist.update_priorities(pick, pr)
    else:
      ids, probs = dist.sample(7)
      on_sample(step, np.asarray(ids), np.asarray(probs))
  return live


def run_uniform(dist, seed, steps, on_sample):
  rs = np.random.RandomState(seed + 900)
  live, next_id = [], 0
  for step in range(steps):
    op = rs.choice(['add', 'add', 'remove', 'sample'])
    if op == 'add' or not live:
      ids = [next_id + 2 * j for j in range(int(rs.randint(1, 5)))]
      next_id = ids[-1] + 1 + int(rs.randint(0, 3))
      dist.add(ids)
      live.extend(ids)
    elif op == 'remove':
      pick = [live[j] for j in rs.choice(len(live), size=int(rs.randint(1, min(3, len(live)) + 1)),
                                         replace=False)]
      dist.remove(pick)
      for i in pick:
        live.remove(i)
    else:
      on_sample(step, np.asarray(dist.sample(6)))
  return live


def main():
  ref = ref_loader.load_reference_replay()
  if ref is None:
    raise SystemExit('reference not available')
  for name, seed, expo, usp, cmin, cmax, steps in CASES:
    dist = ref.PrioritizedDistribution(expo, usp, np.random.RandomState(seed), cmin, cmax)
    ids_log, probs_log, step_log = [], [], []

    def on_sample(step, ids, probs):
      step_log.append(step)
      ids_log.append(ids.astype(np.int64))
      probs_log.append(protocol.f64_bits(probs))

    live = run_program(dist, seed, steps, cmax, on_sample, zero_priorities=name == 'zero_tree')
    st = dist.get_state()
    np.savez_compressed(
        os.path.join(HERE, 'dist_general_%s.npz' % name),
        steps=np.array(step_log), ids=np.stack(ids_log), probs_bits=np.stack(probs_log),
        live=np.array(sorted(live), dtype=np.int64), capacity=np.int64(dist.capacity),
        active_indices=np.array(st['active_indices'], dtype=np.int64),
        inactive_indices=np.array(st['inactive_indices'], dtype=np.int64),
        tree_bits=protocol.f64_bits(st['sum_tree']['storage']))
    print('wrote', name, len(step_log), 'samples, capacity', dist.capacity)
  uni = ref.UniformDistribution(np.random.RandomState(21))
  log = []
  live = run_uniform(uni, 21, 150, lambda step, ids: log.append(ids.astype(np.int64)))
  np.savez_compressed(os.path.join(HERE, 'dist_general_uniform.npz'), ids=np.stack(log),
                      order=np.array(uni.get_state()['ids'], dtype=np.int64),
                      live=np.array(sorted(live), dtype=np.int64))
  print('wrote uniform', len(log), 'samples')


if __name__ == '__main__':
  main()
