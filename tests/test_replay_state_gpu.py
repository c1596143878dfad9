"""Row f3 on the GPU: `get_state()` emits the REFERENCE's dictionary and
`set_state()` loads one (ref: replay.py:178-191, 747-760).

For each fixture frozen from the unmodified reference (tests/golden/
state_*.npz, written by gen_replay_golden.py at a mid-trace step):
  1. the HIP replay driven to that step returns a state EQUAL to the reference's
     (storage items and order, every id/index table, sum-tree float64 bits);
  2. a fresh HIP replay loaded from the reference's state (it never saw the
     first half of the trace) continues the protocol bit-identically to the
     reference's continuation: ids, probability bits, root bits, final tree."""

import os

import numpy as np
import pytest
import torch

from tests.golden import protocol

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(__file__), 'golden')
PRIO = {c[0]: c for c in protocol.PRIORITIZED_CASES}
UNI = {c[0]: c for c in protocol.UNIFORM_CASES}


@pytest.fixture(scope='module')
def rl():
  from dqn_zoo_amd import replay as replay_lib
  return replay_lib


def _make_prio(rl, case, rs):
  name, cap, fill, batch, steps, seed, expo, usp, norm = case
  return rl.PrioritizedTransitionReplay(
      cap, protocol.Item(None, None), expo, protocol.beta_schedule(cap), usp,
      norm, rs)


def _restored_rng(z):
  rs = np.random.RandomState(0)
  rs.set_state(('MT19937', z['rng_key'], int(z['rng_pos'])))
  return rs


@pytest.mark.parametrize('name', sorted(protocol.STATE_SNAPSHOTS))
def test_prioritized_state_equals_reference_and_restores(rl, name):
  case = PRIO[name]
  _, cap, fill, batch, steps, seed, expo, usp, norm = case
  z = np.load(os.path.join(GOLDEN, 'state_prio_%s.npz' % name))
  g = np.load(os.path.join(GOLDEN, 'replay_prio_%s.npz' % name))
  want = protocol.unpack_state(z, prioritized=True)
  k0 = int(z['snapshot_step'])
  rs = np.random.RandomState(seed)
  rep = _make_prio(rl, case, rs)
  ids_log = []
  seen = {}

  def on_snapshot(r, t, max_seen):
    got = r.get_state()
    assert protocol.states_equal(got, want, prioritized=True) == ''
    assert max_seen == float(z['max_seen'])
    # native compact formats round-trip too (host arrays / HBM clones)
    for mode in (True, 'device'):
      c = _make_prio(rl, case, rs)
      c.set_state(r.get_state(compact=mode))
      assert protocol.states_equal(c.get_state(), want, prioritized=True) == ''
    # continue on a restored copy that shares the RNG stream
    fresh = _make_prio(rl, case, rs)
    fresh.set_state(got)
    fresh.max_seen_priority_device.fill_(max_seen)
    seen['fresh'] = fresh
    return fresh

  protocol.drive_prioritized(rep, cap, fill, batch, steps, seed,
                             lambda k, ids, w: ids_log.append(ids),
                             snapshot_at=k0, on_snapshot=on_snapshot)
  np.testing.assert_array_equal(np.stack(ids_log), g['ids'])
  np.testing.assert_array_equal(
      protocol.f64_bits(seen['fresh'].tree_storage.cpu().numpy()),
      g['tree_storage_bits'])


@pytest.mark.parametrize('name', sorted(protocol.STATE_SNAPSHOTS))
def test_prioritized_continues_from_reference_state(rl, name):
  """A replay that never ran the first half: loaded from the REFERENCE's state."""
  case = PRIO[name]
  _, cap, fill, batch, steps, seed, expo, usp, norm = case
  z = np.load(os.path.join(GOLDEN, 'state_prio_%s.npz' % name))
  g = np.load(os.path.join(GOLDEN, 'replay_prio_%s.npz' % name))
  k0 = int(z['snapshot_step'])
  assert steps - k0 >= 20
  rep = _make_prio(rl, case, _restored_rng(z))
  rep.set_state(protocol.unpack_state(z, prioritized=True))
  assert rep.check_valid()[0] and rep.size == len(z['storage_ids'])
  probs_log, ids_log, w_log, root_log = [], [], [], []
  orig = rl.importance_sampling_weights

  def spy(probabilities, **kw):
    probs_log.append(np.array(probabilities, dtype=np.float64))
    return orig(probabilities, **kw)

  rl.importance_sampling_weights = spy
  try:
    def on_sample(k, ids, w):
      ids_log.append(ids)
      w_log.append(w)
      root_log.append(float(rep.tree_storage[1].item()))
    protocol.drive_prioritized(rep, cap, fill, batch, steps, seed, on_sample,
                               resume=(k0, int(z['t']), float(z['max_seen'])))
  finally:
    rl.importance_sampling_weights = orig
  np.testing.assert_array_equal(np.stack(ids_log), g['ids'][k0:])
  np.testing.assert_array_equal(protocol.f64_bits(np.stack(probs_log)),
                                g['probs_bits'][k0:])
  np.testing.assert_array_equal(protocol.f64_bits(np.array(root_log)),
                                g['root_bits'][k0:])
  np.testing.assert_array_equal(
      protocol.f64_bits(rep.tree_storage.cpu().numpy()), g['tree_storage_bits'])
  assert int(rep._t) == int(g['final_t'])  # pylint: disable=protected-access
  # stored items survived: Item.a carries the id
  for i, item in zip(rep.ids(), rep.get(rep.ids())):
    assert int(item.a) == i


@pytest.mark.parametrize('name', sorted(protocol.UNIFORM_STATE_SNAPSHOTS))
def test_uniform_state_equals_reference_and_restores(rl, name):
  _, cap, fill, batch, steps, seed = UNI[name]
  z = np.load(os.path.join(GOLDEN, 'state_uni_%s.npz' % name))
  g = np.load(os.path.join(GOLDEN, 'replay_uni_%s.npz' % name))
  want = protocol.unpack_state(z, prioritized=False)
  k0 = int(z['snapshot_step'])
  rs = np.random.RandomState(seed)
  rep = rl.TransitionReplay(cap, protocol.Item(None, None), rs)
  log = []

  def on_snapshot(r, t):
    got = r.get_state()
    assert protocol.states_equal(got, want, prioritized=False) == ''
    c = rl.TransitionReplay(cap, protocol.Item(None, None), rs)
    c.set_state(r.get_state(compact=True))
    assert protocol.states_equal(c.get_state(), want, prioritized=False) == ''
    fresh = rl.TransitionReplay(cap, protocol.Item(None, None), rs)
    fresh.set_state(got)
    return fresh

  protocol.drive_uniform(rep, cap, fill, batch, steps, seed,
                         lambda k, s: log.append(np.asarray(s.a, dtype=np.int64)),
                         snapshot_at=k0, on_snapshot=on_snapshot)
  np.testing.assert_array_equal(np.stack(log), g['ids'])
  # ... and from the reference's state alone
  rep2 = rl.TransitionReplay(cap, protocol.Item(None, None), _restored_rng(z))
  rep2.set_state(want)
  log2 = []
  protocol.drive_uniform(rep2, cap, fill, batch, steps, seed,
                         lambda k, s: log2.append(np.asarray(s.a, dtype=np.int64)),
                         resume=(k0, int(z['t'])))
  np.testing.assert_array_equal(np.stack(log2), g['ids'][k0:])


def test_set_state_invalidates_cached_sample_slots(rl):
  """sample -> set_state -> sample on the SAME object must gather
  from the re-allocated field arrays, not from the freed ones."""
  cap, b = 64, 16
  rs = np.random.RandomState(3)
  rep = rl.TransitionReplay(cap, protocol.Item(None, None), rs)
  for i in range(cap):
    rep.add(protocol.Item(i, -i))
  out, ids = rep.sample_device(b)   # builds the cached descriptor ring
  other = rl.TransitionReplay(cap, protocol.Item(None, None), rs)
  for i in range(cap + 10):
    other.add(protocol.Item(1000 + i, 7))
  for state in (other.get_state(), other.get_state(compact=True),
                other.get_state(compact='device')):
    junk = [torch.full((cap * 4,), -1, dtype=torch.int64, device='cuda')
            for _ in range(4)]  # recycle freed blocks with poison
    rep.set_state(state)
    del junk
    out, ids = rep.sample_device(b)
    np.testing.assert_array_equal(out.a.cpu().numpy(), 1000 + ids.cpu().numpy())
    assert (out.b.cpu().numpy() == 7).all()


def test_rejects_state_outside_the_closed_forms(rl):
  z = np.load(os.path.join(GOLDEN, 'state_prio_n8_wrap.npz'))
  st = protocol.unpack_state(z, prioritized=True)
  a = st['distribution']['active_indices']
  a[0], a[1] = a[1], a[0]
  rep = _make_prio(rl, PRIO['n8_wrap'], np.random.RandomState(0))
  with pytest.raises(ValueError, match='active_indices'):
    rep.set_state(st)
