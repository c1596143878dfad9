"""Generates tests/golden/replay_*.npz from the REAL reference replay module.

Run in the dev container (needs /root/reference):
    python tests/golden/gen_replay_golden.py [--big]

Each trace records, per protocol step (see protocol.py): sampled ids (int64),
sampling probabilities / IS weights / sum-tree root as raw float64 bit
patterns, and at the end the sha256 of the whole sum-tree storage plus the
private id<->index tables the closed forms are checked against.
"""

import copy
import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, '..', '..'))

from oracle import ref_loader  # noqa: E402
from tests.golden import protocol  # noqa: E402


def gen_prioritized(ref, case):
  name, cap, fill, batch, steps, seed, expo, usp, norm = case
  rs = np.random.RandomState(seed)
  replay = ref.PrioritizedTransitionReplay(
      capacity=cap, structure=protocol.Item(None, None),
      priority_exponent=expo,
      importance_sampling_exponent=protocol.beta_schedule(cap),
      uniform_sample_probability=usp, normalize_weights=norm, random_state=rs)

  probs_log = []
  orig = ref.importance_sampling_weights

  def spy(probabilities, **kw):
    probs_log.append(np.array(probabilities, dtype=np.float64))
    return orig(probabilities, **kw)

  ref.importance_sampling_weights = spy
  ids_log, w_log, root_log = [], [], []
  tree = replay._distribution._sum_tree  # pylint: disable=protected-access

  def on_sample(k, ids, weights):
    ids_log.append(ids.astype(np.int64))
    w_log.append(weights.astype(np.float64))
    root_log.append(tree.root())

  snap_at = protocol.STATE_SNAPSHOTS.get(name)

  def on_snapshot(rep, t, max_seen):
    # the reference's own get_state(), frozen (it returns live references)
    st = copy.deepcopy(rep.get_state())
    packed = protocol.pack_state(st, prioritized=True)
    packed['snapshot_step'] = np.int64(snap_at)
    packed['max_seen'] = np.float64(max_seen)
    key = rs.get_state()
    packed['rng_key'] = np.asarray(key[1], dtype=np.uint32)
    packed['rng_pos'] = np.int64(key[2])
    np.savez_compressed(os.path.join(HERE, 'state_prio_%s.npz' % name), **packed)
    # round trip through the reference itself: a fresh object restored from the
    # unpacked fixture must continue identically (checked by the trace below
    # being recorded on the restored object)
    fresh = ref.PrioritizedTransitionReplay(
        capacity=cap, structure=protocol.Item(None, None), priority_exponent=expo,
        importance_sampling_exponent=protocol.beta_schedule(cap),
        uniform_sample_probability=usp, normalize_weights=norm, random_state=rs)
    fresh.set_state(protocol.unpack_state(packed, prioritized=True))
    nonlocal tree, replay
    replay = fresh
    tree = fresh._distribution._sum_tree  # pylint: disable=protected-access
    return fresh

  try:
    protocol.drive_prioritized(replay, cap, fill, batch, steps, seed, on_sample,
                               snapshot_at=snap_at, on_snapshot=on_snapshot)
  finally:
    ref.importance_sampling_weights = orig

  dist = replay._distribution  # pylint: disable=protected-access
  storage = tree.get_state()['storage']
  out = dict(
      ids=np.stack(ids_log), probs_bits=protocol.f64_bits(np.stack(probs_log)),
      weights_bits=protocol.f64_bits(np.stack(w_log)),
      root_bits=protocol.f64_bits(np.array(root_log)),
      tree_sha256=np.frombuffer(
          hashlib.sha256(np.ascontiguousarray(storage).tobytes()).digest(),
          dtype=np.uint8),
      final_t=np.int64(replay._t),  # pylint: disable=protected-access
  )
  if cap <= 1000:
    out['tree_storage_bits'] = protocol.f64_bits(storage)
    out['active_indices'] = np.array(dist._active_indices, dtype=np.int64)
    id_sorted = sorted(dist._id_to_index.keys())
    out['live_ids'] = np.array(id_sorted, dtype=np.int64)
    out['live_tree_index'] = np.array(
        [dist._id_to_index[i] for i in id_sorted], dtype=np.int64)
  np.savez_compressed(os.path.join(HERE, 'replay_prio_%s.npz' % name), **out)
  print('wrote', name, 'steps', steps)


def gen_uniform(ref, case):
  name, cap, fill, batch, steps, seed = case
  rs = np.random.RandomState(seed)
  replay = ref.TransitionReplay(cap, protocol.Item(None, None), rs)
  ids_log = []

  def on_sample(k, s):
    ids_log.append(np.asarray(s.a, dtype=np.int64))

  snap_at = protocol.UNIFORM_STATE_SNAPSHOTS.get(name)

  def on_snapshot(rep, t):
    st = copy.deepcopy(rep.get_state())
    packed = protocol.pack_state(st, prioritized=False)
    packed['snapshot_step'] = np.int64(snap_at)
    key = rs.get_state()
    packed['rng_key'] = np.asarray(key[1], dtype=np.uint32)
    packed['rng_pos'] = np.int64(key[2])
    np.savez_compressed(os.path.join(HERE, 'state_uni_%s.npz' % name), **packed)
    fresh = ref.TransitionReplay(cap, protocol.Item(None, None), rs)
    fresh.set_state(protocol.unpack_state(packed, prioritized=False))
    nonlocal replay
    replay = fresh
    return fresh

  protocol.drive_uniform(replay, cap, fill, batch, steps, seed, on_sample,
                         snapshot_at=snap_at, on_snapshot=on_snapshot)
  out = dict(ids=np.stack(ids_log))
  if cap <= 1000:
    out['pos_to_id'] = np.array(
        replay._distribution._ids, dtype=np.int64)  # pylint: disable=protected-access
  np.savez_compressed(os.path.join(HERE, 'replay_uni_%s.npz' % name), **out)
  print('wrote', name)


def main():
  ref = ref_loader.load_reference_replay()
  if ref is None:
    raise SystemExit('reference not available')
  for c in protocol.PRIORITIZED_CASES:
    gen_prioritized(ref, c)
  for c in protocol.UNIFORM_CASES:
    gen_uniform(ref, c)
  if '--big' in sys.argv:
    gen_prioritized(ref, protocol.PRIORITIZED_BIG)
    gen_uniform(ref, protocol.UNIFORM_BIG)


if __name__ == '__main__':
  main()
