"""GPU parity tests for the replay half of the hot path (through the C ABI).

Compares the HIP-backed classes of dqn_zoo_amd.replay with (a) the golden
traces generated from the real reference, (b) the CPU oracle on fresh seeds,
(c) the reference's known-answer tables.  Integer/index work and float64 tree
contents are compared BIT-EXACT.
"""

import os

import numpy as np
import pytest
import torch

from oracle import replay_oracle as ro
from tests.golden import protocol

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(__file__), 'golden')


@pytest.fixture(scope='module')
def rl():
  from dqn_zoo_amd import replay as replay_lib
  return replay_lib


def _bits(x):
  return protocol.f64_bits(x)


# ---- SumTree known answers (replay_test.py:820-1045) ------------------------
@pytest.mark.parametrize('expected,target', [
    (0, 0.0), (0, 3.0 - 0.1), (1, 3.0), (1, 4.0 - 0.1), (2, 4.0),
    (2, 6.0 - 0.1), (3, 6.0), (3, 11.0 - 0.1)])
def test_query_typical(rl, expected, target):
  t = rl.SumTree()
  t.set_all([3.0, 1.0, 2.0, 5.0])
  assert t.query([target]) == [expected]


def test_sumtree_surface(rl):
  t = rl.SumTree()
  assert t.size == 0 and np.isnan(t.root()) and t.check_valid()[0]
  t.resize(3)
  assert t.size == 3 and list(t.get([0, 1, 2])) == [0, 0, 0]
  t.set_all([4.0, 5.0, 3.0])
  assert t.capacity == 4 and t.check_valid()[0]
  np.testing.assert_array_equal(t.values, [4.0, 5.0, 3.0])
  t.resize(8)
  np.testing.assert_array_equal(t.values, [4, 5, 3, 0, 0, 0, 0, 0])
  assert t.capacity == 8 and t.check_valid()[0]
  t.resize(2)
  np.testing.assert_array_equal(t.values, [4, 5])
  assert t.root() == 9.0 and t.check_valid()[0]
  t.set_all([4, 5, 3, 9])
  t.set([2, 0], [99, 88])
  np.testing.assert_array_equal(t.values, [88, 5, 99, 9])
  t.set([1, 1], [1.0, 7.0])  # duplicates: last wins (replay.py:283)
  assert t.get([1])[0] == 7.0 and t.root() == 88 + 7 + 99 + 9
  assert t.query([2.9 + 88, 95.0]) == [1, 2]
  for bad in (-1, np.nan, np.inf):
    with pytest.raises(ValueError):
      t.set([1], [bad])
  with pytest.raises(ValueError):
    rl.SumTree().set_all([1, -1])
  for i in (-1, 4):
    with pytest.raises(IndexError):
      t.get([i])
  for target in (-1.0, t.root(), t.root() + 1):
    with pytest.raises(ValueError):
      t.query([target])
  s = t.get_state()
  u = rl.SumTree()
  u.set_state(s)
  np.testing.assert_array_equal(u.values, t.values)
  assert u.check_valid()[0] and u.capacity == t.capacity


@pytest.mark.parametrize(
    'target', [0, 0.1, 0.9, 1, 1.1, 3.9, 4, 4.1, 5.9, 6, 6.1, 8.9, 8.999999])
def test_query_never_returns_zero_leaf(rl, target):
  v = np.array([0, 1, 0, 0, 3, 0, 2, 0, 3, 0], dtype=np.float64)
  t = rl.SumTree()
  t.set_all(v)
  assert v[t.query([target])[0]] != 0


@pytest.mark.parametrize('seed', range(10))
def test_sumtree_random_ops_vs_oracle(rl, seed):
  """replay_test.py:1120-1161 protocol, HIP tree vs oracle tree, bit-exact."""

  def ops(tree, leaves):
    rs = np.random.RandomState(seed)
    out = []
    for _ in range(12):
      tree.resize(int(rs.randint(10, 40)))
      tree.set(rs.randint(tree.size, size=3), np.abs(rs.standard_cauchy(3)))
      out.append(list(tree.query(rs.uniform(0, tree.root(), size=4))))
      tree.set_all(np.abs(rs.standard_cauchy(int(rs.randint(10, 40)))))
      tree.set(rs.randint(tree.size, size=4), np.abs(rs.standard_cauchy(4)))
      out.append(list(tree.query(rs.uniform(0, tree.root(), size=3))))
      out.append(_bits(leaves(tree)).tolist())
      out.append(_bits([tree.root()]).tolist())
    return out

  a = ops(rl.SumTree(), lambda t: t.values)
  b = ops(ro.SumTreeOracle(), lambda t: t.leaves())
  assert a == b


def test_device_float_ops_are_correctly_rounded(rl):
  """sqrt (f32 and f64) and the tree's f64 adds must round like NumPy."""
  n = 1024
  rs = np.random.RandomState(7)
  for rep in range(8):
    p = np.abs(rs.standard_cauchy(n)).astype(np.float64)
    r = rl.PrioritizedTransitionReplay(
        n, protocol.Item(None, None), 0.5, lambda t: 1.0, 0.0, True,
        np.random.RandomState(0))
    r.bulk_fill([torch.zeros(n, dtype=torch.int64, device='cuda'),
                 torch.zeros(n, dtype=torch.int64, device='cuda')])
    ids = torch.arange(n, device='cuda')
    r.update_priorities(ids, torch.from_numpy(p).cuda())
    leaf = r.tree_storage[r._cap_pow2:2 * r._cap_pow2].cpu().numpy()
    exp = np.sqrt(p)[::-1]  # tree index = n-1-id
    np.testing.assert_array_equal(_bits(leaf), _bits(exp))
    p32 = p.astype(np.float32)
    r.update_priorities(ids, torch.from_numpy(p32).cuda())
    leaf = r.tree_storage[r._cap_pow2:2 * r._cap_pow2].cpu().numpy()
    exp = ro.power_zero_safe(p32, 0.5)[::-1].astype(np.float64)
    np.testing.assert_array_equal(_bits(leaf), _bits(exp))
    t = ro.SumTreeOracle()
    t.set_all(exp)
    np.testing.assert_array_equal(_bits(r.tree_storage.cpu().numpy()),
                                  _bits(t.node))
    assert float(r.max_seen_priority_device.item()) == max(1.0, p.max(), float(p32.max()))


# ---- golden traces ------------------------------------------------------------
def _device_bulk(replay, n):
  a = torch.arange(n, dtype=torch.int64, device='cuda')
  replay.bulk_fill([a, -a])


def _run_prio(rl, case, bulk):
  name, cap, fill, batch, steps, seed, expo, usp, norm = case
  rep = rl.PrioritizedTransitionReplay(
      cap, protocol.Item(None, None), expo, protocol.beta_schedule(cap), usp,
      norm, np.random.RandomState(seed))
  ids_log, w_log, root_log, probs_log = [], [], [], []

  def on_sample(k, ids, w):
    ids_log.append(ids)
    w_log.append(w)
    root_log.append(float(rep.tree_storage[1].item()))

  # the device's sampling probabilities, as `sample()` hands them to the host
  # weight computation (same tap point as the golden generator's spy)
  orig = rl.importance_sampling_weights

  def spy(probabilities, **kw):
    probs_log.append(np.array(probabilities, dtype=np.float64))
    return orig(probabilities, **kw)

  rl.importance_sampling_weights = spy
  try:
    protocol.drive_prioritized(rep, cap, fill, batch, steps, seed, on_sample,
                               bulk_fill=_device_bulk if bulk else None)
  finally:
    rl.importance_sampling_weights = orig
  rep.golden_probs = np.stack(probs_log)
  return rep, np.stack(ids_log), np.stack(w_log), np.array(root_log)


@pytest.mark.parametrize('case', protocol.PRIORITIZED_CASES,
                         ids=[c[0] for c in protocol.PRIORITIZED_CASES])
def test_prioritized_golden(rl, case):
  g = np.load(os.path.join(GOLDEN, 'replay_prio_%s.npz' % case[0]))
  rep, ids, w, roots = _run_prio(rl, case, bulk=False)
  np.testing.assert_array_equal(ids, g['ids'])
  np.testing.assert_array_equal(_bits(roots), g['root_bits'])
  # sampling probabilities: BITWISE equal to the reference's
  np.testing.assert_array_equal(_bits(rep.golden_probs), g['probs_bits'])
  # weights = NumPy `**beta` of those probabilities on this host: the last bit of
  # NumPy's SIMD pow may differ from the machine that generated the fixture
  np.testing.assert_allclose(w, g['weights_bits'].view(np.float64), rtol=4e-16)
  np.testing.assert_array_equal(_bits(rep.tree_storage.cpu().numpy()),
                                g['tree_storage_bits'])
  assert rep.check_valid()[0]
  assert list(rep.ids()) == list(g['live_ids'])


def test_prioritized_golden_1m(rl):
  """BASELINE full size: 1M capacity, 2^20-leaf tree, reference trace."""
  import hashlib
  case = protocol.PRIORITIZED_BIG
  g = np.load(os.path.join(GOLDEN, 'replay_prio_%s.npz' % case[0]))
  rep, ids, w, roots = _run_prio(rl, case, bulk=True)
  np.testing.assert_array_equal(ids, g['ids'])
  np.testing.assert_array_equal(_bits(roots), g['root_bits'])
  np.testing.assert_array_equal(_bits(rep.golden_probs), g['probs_bits'])
  np.testing.assert_allclose(w, g['weights_bits'].view(np.float64), rtol=4e-16)
  sha = hashlib.sha256(rep.tree_storage.cpu().numpy().tobytes()).digest()
  assert sha == g['tree_sha256'].tobytes()
  assert rep.check_valid()[0]


@pytest.mark.parametrize('case', protocol.UNIFORM_CASES + [protocol.UNIFORM_BIG],
                         ids=[c[0] for c in protocol.UNIFORM_CASES] + ['u1m'])
def test_uniform_golden(rl, case):
  name, cap, fill, batch, steps, seed = case
  g = np.load(os.path.join(GOLDEN, 'replay_uni_%s.npz' % name))
  rep = rl.TransitionReplay(cap, protocol.Item(None, None),
                            np.random.RandomState(seed))
  log = []
  protocol.drive_uniform(rep, cap, fill, batch, steps, seed,
                         lambda k, s: log.append(np.asarray(s.a)),
                         bulk_fill=_device_bulk if cap > 10000 else None)
  np.testing.assert_array_equal(np.stack(log), g['ids'])
  # FIFO window invariant (replay_test.py:129-147)
  ids = list(rep.ids())
  assert ids == list(range(ids[0], ids[0] + rep.size)) and rep.check_valid()[0]


# ---- pipelined path vs oracle (fresh seeds) ----------------------------------
@pytest.mark.parametrize('seed,cap,expo,usp', [
    (11, 50, 0.5, 1e-3), (12, 1000, 0.5, 0.1), (13, 333, 1.0, 1e-3),
    (14, 4096, 0.5, 1e-3), (15, 100, 0.0, 0.5)])
def test_pipelined_sample_update_vs_oracle(rl, seed, cap, expo, usp):
  """sample_device() + update_priorities(device f32) + device-priority adds,
  as the on-device learner drives them; oracle gets the same f32 values."""
  batch = 32
  beta = protocol.beta_schedule(cap)
  dev = rl.PrioritizedTransitionReplay(
      cap, protocol.Item(None, None), expo, beta, usp, True,
      np.random.RandomState(seed))
  cpu = ro.PrioritizedReplayOracle(cap, protocol.Item(None, None), expo, beta,
                                   usp, True, np.random.RandomState(seed))
  prs = np.random.RandomState(seed + 99)
  max_seen = 1.0
  mismatched = 0
  t = 0
  for _ in range(cap - 3):
    dev.add(protocol.Item(t, -t), 1.0) if cap <= 100 else None
    cpu.add(protocol.Item(t, -t), 1.0)
    t += 1
  if cap > 100:
    _device_bulk(dev, cap - 3)
  for k in range(60):
    s = dev.sample_device(batch)
    ids_c, probs_c, w_c = cpu.sample_ids(batch)
    np.testing.assert_array_equal(s.ids.cpu().numpy(), ids_c)
    np.testing.assert_array_equal(_bits(s.probabilities.cpu().numpy()),
                                  _bits(probs_c))
    np.testing.assert_allclose(s.weights.cpu().numpy(), w_c, rtol=1e-14)
    # float32 weights (what the learner consumes): the cast of a float64 value that
    # agrees with NumPy's to 1e-14 -- identical unless that value sits within 1e-14
    # (relative) of a float32 rounding boundary, then one ulp apart
    w32 = s.weights32.cpu().numpy()
    ref32 = w_c.astype(np.float32)
    bad = w32 != ref32
    mismatched += int(bad.sum())
    assert np.all(np.abs(w32[bad] - ref32[bad]) <= np.spacing(ref32[bad]))
    np.testing.assert_array_equal(s.transitions.a.cpu().numpy(), ids_c)
    p32 = np.clip(np.abs(prs.standard_cauchy(batch)), 0, 100).astype(np.float32)
    p32[prs.uniform(size=batch) < 0.05] = 0.0
    dev.update_priorities(s.ids, torch.from_numpy(p32).cuda())
    cpu.update_priorities(ids_c, p32)
    max_seen = np.max([max_seen, p32.max()])
    for _ in range(4):
      dev.add_with_device_priority(protocol.Item(t, k))
      cpu.add(protocol.Item(t, k), max_seen)
      t += 1
  dev.check_status()
  assert mismatched <= 1, mismatched   # of 60 x 32 float32 weights
  assert float(dev.max_seen_priority_device.item()) == float(max_seen)
  np.testing.assert_array_equal(_bits(dev.tree_storage.cpu().numpy()),
                                _bits(cpu.dist.tree.node))


@pytest.mark.parametrize('batch', [64, 200, 255, 256, 300])
@pytest.mark.parametrize('cap', [37, 700])
def test_wide_batch_write_back_vs_oracle(rl, batch, cap):
  """update_priorities(device ids) with up to 256+ ids, many of them duplicated (cap 37)
  and all partner levels populated (cap 700): the two-round-trip walk serves n <= 255
  (its partner table stores batch positions as bytes, 0xFF = none -- ADVICE r3: at
  n == 256 position 255 collided with the mark), the level-by-level form the rest; both
  must leave the oracle's tree, bit for bit."""
  dev = rl.PrioritizedTransitionReplay(
      cap, protocol.Item(None, None), 0.5, protocol.beta_schedule(cap), 1e-3, True,
      np.random.RandomState(5))
  cpu = ro.PrioritizedReplayOracle(cap, protocol.Item(None, None), 0.5,
                                   protocol.beta_schedule(cap), 1e-3, True,
                                   np.random.RandomState(5))
  for t in range(cap + 5):
    dev.add(protocol.Item(t, -t), 1.0)
    cpu.add(protocol.Item(t, -t), 1.0)
  prs = np.random.RandomState(batch * 1000 + cap)
  for _ in range(6):
    ids = prs.randint(5, cap + 5, size=batch).astype(np.int64)
    ids[-1] = ids[0]                       # the last position always has a duplicate
    p32 = np.abs(prs.standard_cauchy(batch)).astype(np.float32)
    dev.update_priorities(torch.from_numpy(ids).cuda(), torch.from_numpy(p32).cuda())
    cpu.update_priorities(ids, p32)
    dev.check_status()
    np.testing.assert_array_equal(_bits(dev.tree_storage.cpu().numpy()),
                                  _bits(cpu.dist.tree.node))


# ---- gather ------------------------------------------------------------------
def test_gather_bytes_and_dtypes(rl):
  cap, batch = 300, 32
  rs = np.random.RandomState(3)
  for cls in ('uniform', 'prio'):
    if cls == 'uniform':
      rep = rl.TransitionReplay(cap, rl.Transition(None, None, None, None, None),
                                np.random.RandomState(5))
    else:
      rep = rl.PrioritizedTransitionReplay(
          cap, rl.Transition(None, None, None, None, None), 0.5,
          lambda t: 0.6, 1e-3, True, np.random.RandomState(5))
    host = {}
    for i in range(cap + 57):  # wraps around
      tr = rl.Transition(
          s_tm1=rs.randint(0, 256, (84, 84, 4)).astype(np.uint8),
          a_tm1=int(rs.randint(6)), r_t=float(rs.choice([-1., 0., 1.])),
          discount_t=float(rs.choice([0.0, 0.99 ** 3])),
          s_t=rs.randint(0, 256, (84, 84, 4)).astype(np.uint8))
      host[i] = tr
      rep.add(tr) if cls == 'uniform' else rep.add(tr, 1.0 + i % 3)
    for _ in range(5):
      if cls == 'uniform':
        ids = rep.sample_ids_device(batch)
        outs = rep._ring.gather(ids, batch, rep._stream())
        out = rep._to_host(outs)
        ids = ids.cpu().numpy()
      else:
        out, ids, w = rep.sample(batch)
        assert w.dtype == np.float64 and ids.dtype == np.int64
      assert out.s_tm1.dtype == np.uint8 and out.s_tm1.shape == (batch, 84, 84, 4)
      assert out.a_tm1.dtype == np.int64 and out.r_t.dtype == np.float64
      assert out.discount_t.dtype == np.float64
      for b, i in enumerate(ids):
        assert i >= 57
        np.testing.assert_array_equal(out.s_tm1[b], host[i].s_tm1)
        np.testing.assert_array_equal(out.s_t[b], host[i].s_t)
        assert out.a_tm1[b] == host[i].a_tm1 and out.r_t[b] == host[i].r_t
        assert out.discount_t[b] == host[i].discount_t
    got = list(rep.get([60, cap + 56]))
    np.testing.assert_array_equal(got[0].s_t, host[60].s_t)
    assert got[1].a_tm1 == host[cap + 56].a_tm1


def test_sample_gather_ragged_field_sizes(rl):
  """The one-launch sample+gather with a generic structure whose fields mix a wide
  16-byte-aligned row (several chunk blocks), rows of 1000 and 600 bytes (16-byte path
  impossible: byte copies, still more than one block's 512 bytes) and a scalar: every
  byte of every sampled row arrives (the early exit of the surplus chunk
  blocks used the vector path's unit for byte-path fields)."""
  import collections
  import torch
  S = collections.namedtuple('S', ['wide', 'odd', 'odd2', 'k'])
  cap, batch = 64, 16
  rep = rl.PrioritizedTransitionReplay(cap, S(None, None, None, None), 0.5, lambda t: 0.6,
                                       1e-3, True, np.random.RandomState(9))
  rs = np.random.RandomState(4)
  host = {}
  for i in range(cap + 9):
    it = S(rs.randint(0, 256, (84, 84, 4)).astype(np.uint8),
           rs.randint(0, 256, (1000,)).astype(np.uint8),
           rs.randint(0, 256, (3, 200)).astype(np.uint8), int(i))
    host[i] = it
    rep.add(it, 1.0 + i % 5)
  for _ in range(4):
    s = rep.sample_device(batch)
    torch.cuda.synchronize()
    ids = s.ids.cpu().numpy()
    t = s.transitions
    for b, i in enumerate(ids):
      np.testing.assert_array_equal(t.wide[b].cpu().numpy(), host[i].wide)
      np.testing.assert_array_equal(t.odd[b].cpu().numpy(), host[i].odd)
      np.testing.assert_array_equal(t.odd2[b].cpu().numpy(), host[i].odd2)
      assert int(t.k[b]) == host[i].k
  rep.check_status()


def test_error_behaviour(rl):
  S = protocol.Item(None, None)
  with pytest.raises(ValueError, match='priority_exponent'):
    rl.PrioritizedTransitionReplay(8, S, -1.0, lambda t: 1., 0.1, True,
                                   np.random.RandomState(0))
  with pytest.raises(ValueError, match='uniform_sample_probability'):
    rl.PrioritizedTransitionReplay(8, S, 1.0, lambda t: 1., 1.5, True,
                                   np.random.RandomState(0))
  r = rl.PrioritizedTransitionReplay(8, S, 1.0, lambda t: 1., 0.1, True,
                                     np.random.RandomState(0))
  with pytest.raises(RuntimeError, match='No IDs to sample.'):
    r.sample(2)
  for i in range(10):
    r.add(protocol.Item(i, i), 1.0)
  with pytest.raises(IndexError, match='ID 1 does not exist.'):
    r.update_priorities([1], [1.0])      # evicted
  with pytest.raises(IndexError, match='ID 10 does not exist.'):
    r.update_priorities([10], [1.0])     # never added
  with pytest.raises(ValueError):
    r.update_priorities([5], [-1.0])
  with pytest.raises(ValueError):
    r.add(protocol.Item(0, 0), np.nan)
  # device-side detection of the same conditions (sticky status word)
  r.update_priorities(torch.tensor([1], device='cuda'),
                      torch.tensor([1.0], dtype=torch.float64, device='cuda'))
  with pytest.raises(IndexError):
    r.check_status()
  r.update_priorities(torch.tensor([5], device='cuda'),
                      torch.tensor([-1.0], dtype=torch.float64, device='cuda'))
  with pytest.raises(ValueError):
    r.check_status()
  # all-zero priorities: uniform fallback with the reference's RNG order
  z = rl.PrioritizedTransitionReplay(8, S, 1.0, lambda t: 1., 0.1, True,
                                     np.random.RandomState(4))
  zo = ro.PrioritizedReplayOracle(8, S, 1.0, lambda t: 1., 0.1, True,
                                  np.random.RandomState(4))
  for i in range(8):
    z.add(protocol.Item(i, i), 0.0)
    zo.add(protocol.Item(i, i), 0.0)
  for _ in range(3):
    _, ids, w = z.sample(6)
    ids_o, _, w_o = zo.sample_ids(6)
    np.testing.assert_array_equal(ids, ids_o)
    np.testing.assert_array_equal(_bits(w), _bits(w_o))
