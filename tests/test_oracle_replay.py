"""Pins the replay oracle (oracle/replay_oracle.py) to the reference.

Three anchors (task §③): the committed golden traces generated from the real
reference module, the reference's own known-answer tables
(replay_test.py:939-987), and -- when /root/reference is present -- a live
randomized comparison against the unmodified reference module.
"""

import glob
import hashlib
import os

import numpy as np
import pytest

from oracle import ref_loader
from oracle import replay_oracle as ro
from tests.golden import protocol

GOLDEN = os.path.join(os.path.dirname(__file__), 'golden')


def _load(kind, name):
  return np.load(os.path.join(GOLDEN, 'replay_%s_%s.npz' % (kind, name)))


# ---- known-answer tables from the reference's own tests --------------------
@pytest.mark.parametrize('expected,target', [
    (0, 0.0), (0, 3.0 - 0.1), (1, 3.0), (1, 4.0 - 0.1), (2, 4.0),
    (2, 6.0 - 0.1), (3, 6.0), (3, 11.0 - 0.1)])
def test_query_typical(expected, target):  # replay_test.py:939-953
  t = ro.SumTreeOracle()
  t.set_all([3.0, 1.0, 2.0, 5.0])
  assert t.query([target]) == [expected]


def test_query_multiple():  # replay_test.py:972-976
  t = ro.SumTreeOracle()
  t.set_all([3.0, 1.0, 2.0, 5.0])
  assert t.query([2.9, 3.0, 4]) == [0, 1, 2]


@pytest.mark.parametrize(
    'target', [0, 0.1, 0.9, 1, 1.1, 3.9, 4, 4.1, 5.9, 6, 6.1, 8.9, 8.999999])
def test_query_never_returns_zero_leaf(target):  # replay_test.py:978-987
  v = np.array([0, 1, 0, 0, 3, 0, 2, 0, 3, 0], dtype=np.float64)
  t = ro.SumTreeOracle()
  t.set_all(v)
  assert v[t.query([target])[0]] != 0


def test_query_out_of_range():  # replay_test.py:955-970
  t = ro.SumTreeOracle()
  t.set_all([3.0, 1.0, 2.0, 5.0])
  for bad in (-1.0, 11.0, 12.0, t.root()):
    with pytest.raises(ValueError):
      t.query([bad])


def test_set_duplicates_last_wins_and_errors():
  t = ro.SumTreeOracle()
  t.set_all([0, 1, 2])
  t.set([1, 1], [5.0, 7.0])
  assert t.get([1])[0] == 7.0 and t.root() == 9.0 and t.consistent()
  for bad in (-1, np.nan, np.inf):  # replay_test.py:995-1005
    with pytest.raises(ValueError):
      t.set([1], [bad])
  with pytest.raises(IndexError):  # replay_test.py:910-916
    t.get([3])
  assert np.isnan(ro.SumTreeOracle().root())  # replay_test.py:822-826


def test_capacity_is_pow2():  # replay_test.py:864-872
  t = ro.SumTreeOracle()
  t.set_all([4.0, 5.0, 3.0, 2.0])
  assert t.cap == 4
  t = ro.SumTreeOracle()
  t.set_all([4.0, 5.0, 3.0, 2.0, 9])
  assert t.cap == 8


def test_power_and_weights_pins():
  assert ro.power_zero_safe([0.0, 4.0], 0.0).tolist() == [0.0, 1.0]
  assert ro.power_zero_safe([0.0, 4.0], 0.5).tolist() == [0.0, 2.0]
  w = ro.is_weights(np.array([0.1, 0.2]), 0.1, 1.0, True)
  np.testing.assert_array_equal(w, [1.0, 0.5])
  with pytest.raises(ValueError):
    ro.is_weights(np.array([0.1]), 0.1, 1.5, True)
  with pytest.raises(ValueError):
    ro.is_weights(np.array([0.0]), 0.1, 0.5, False)


# ---- golden traces ----------------------------------------------------------
def _run_oracle_prioritized(case):
  name, cap, fill, batch, steps, seed, expo, usp, norm = case
  rs = np.random.RandomState(seed)
  rep = ro.PrioritizedReplayOracle(
      cap, protocol.Item(None, None), expo, protocol.beta_schedule(cap), usp,
      norm, rs)
  ids_log, w_log, root_log, probs_log = [], [], [], []
  orig = rep.sample_ids

  def spy(size):
    ids, probs, w = orig(size)
    probs_log.append(probs)
    return ids, probs, w

  rep.sample_ids = spy

  def on_sample(k, ids, w):
    ids_log.append(ids)
    w_log.append(w)
    root_log.append(rep.dist.tree.root())

  protocol.drive_prioritized(rep, cap, fill, batch, steps, seed, on_sample)
  return rep, np.stack(ids_log), np.stack(probs_log), np.stack(w_log), np.array(
      root_log)


@pytest.mark.parametrize('case', protocol.PRIORITIZED_CASES,
                         ids=[c[0] for c in protocol.PRIORITIZED_CASES])
def test_oracle_matches_golden_prioritized(case):
  g = _load('prio', case[0])
  rep, ids, probs, w, roots = _run_oracle_prioritized(case)
  np.testing.assert_array_equal(ids, g['ids'])
  np.testing.assert_array_equal(protocol.f64_bits(probs), g['probs_bits'])
  np.testing.assert_array_equal(protocol.f64_bits(roots), g['root_bits'])
  # weights go through a vectorised pow whose last bit is CPU dependent.
  np.testing.assert_allclose(w, g['weights_bits'].view(np.float64), rtol=4e-16)
  tree = rep.dist.tree
  np.testing.assert_array_equal(protocol.f64_bits(tree.node),
                                g['tree_storage_bits'])
  np.testing.assert_array_equal(np.array(rep.dist.active), g['active_indices'])
  live = sorted(rep.dist.index_of)
  np.testing.assert_array_equal(live, g['live_ids'])
  np.testing.assert_array_equal([rep.dist.index_of[i] for i in live],
                                g['live_tree_index'])
  assert rep.t == int(g['final_t'])


@pytest.mark.parametrize('case', protocol.UNIFORM_CASES,
                         ids=[c[0] for c in protocol.UNIFORM_CASES])
def test_oracle_matches_golden_uniform(case):
  name, cap, fill, batch, steps, seed = case
  g = _load('uni', name)
  rep = ro.UniformReplayOracle(cap, protocol.Item(None, None),
                               np.random.RandomState(seed))
  log = []
  protocol.drive_uniform(rep, cap, fill, batch, steps, seed,
                         lambda k, s: log.append(np.asarray(s.a)))
  np.testing.assert_array_equal(np.stack(log), g['ids'])
  np.testing.assert_array_equal(np.array(rep.ids), g['pos_to_id'])


def test_golden_files_complete():
  names = {os.path.basename(p) for p in glob.glob(os.path.join(GOLDEN, '*.npz'))}
  for c in protocol.PRIORITIZED_CASES + [protocol.PRIORITIZED_BIG]:
    assert 'replay_prio_%s.npz' % c[0] in names
  for c in protocol.UNIFORM_CASES + [protocol.UNIFORM_BIG]:
    assert 'replay_uni_%s.npz' % c[0] in names


# ---- live comparison against the unmodified reference module ---------------
@pytest.mark.skipif(not ref_loader.reference_available(),
                    reason='/root/reference not present (GPU box)')
@pytest.mark.parametrize('seed', range(6))
def test_oracle_vs_live_reference(seed):
  ref = ref_loader.load_reference_replay()
  rs = np.random.RandomState(100 + seed)
  cap = int(rs.randint(3, 70))
  fill = int(rs.randint(1, cap + 1))
  batch = int(rs.randint(1, 40))
  expo = [0.0, 0.5, 0.6, 1.0, 0.3, 2.0][seed]   # 0.6/0.3: same-machine pow.
  usp = [1e-3, 0.0, 1.0, 0.3, 0.5, 1e-3][seed]
  norm = bool(seed % 2)
  beta = protocol.beta_schedule(cap)
  a = ref.PrioritizedTransitionReplay(
      capacity=cap, structure=protocol.Item(None, None),
      priority_exponent=expo, importance_sampling_exponent=beta,
      uniform_sample_probability=usp, normalize_weights=norm,
      random_state=np.random.RandomState(seed))
  b = ro.PrioritizedReplayOracle(cap, protocol.Item(None, None), expo, beta,
                                 usp, norm, np.random.RandomState(seed))
  la, lb = [], []
  protocol.drive_prioritized(a, cap, fill, batch, 50, seed,
                             lambda k, i, w: la.append((i, w)))
  protocol.drive_prioritized(b, cap, fill, batch, 50, seed,
                             lambda k, i, w: lb.append((i, w)))
  for (ia, wa), (ib, wb) in zip(la, lb):
    np.testing.assert_array_equal(ia, ib)
    np.testing.assert_array_equal(protocol.f64_bits(wa), protocol.f64_bits(wb))
  np.testing.assert_array_equal(
      protocol.f64_bits(a._distribution._sum_tree._storage),
      protocol.f64_bits(b.dist.tree.node))


@pytest.mark.skipif(not ref_loader.reference_available(),
                    reason='/root/reference not present (GPU box)')
@pytest.mark.parametrize('seed', range(10))
def test_sumtree_random_ops_vs_live_reference(seed):
  """The reference's NaiveSumTree equivalence protocol (replay_test.py:
  1120-1161), run against the reference SumTree itself, bit-exact."""
  ref = ref_loader.load_reference_replay()

  def ops(tree, size_of, root_of):
    rs = np.random.RandomState(seed)
    out = []
    for _ in range(20):
      tree.resize(int(rs.randint(10, 40)))
      tree.set(rs.randint(size_of(tree), size=3),
               np.abs(rs.standard_cauchy(3)))
      out.append(tree.query(rs.uniform(0, root_of(tree), size=4)))
      tree.set_all(np.abs(rs.standard_cauchy(int(rs.randint(10, 40)))))
      tree.set(rs.randint(size_of(tree), size=4),
               np.abs(rs.standard_cauchy(4)))
      out.append(tree.query(rs.uniform(0, root_of(tree), size=3)))
    return out

  ta, tb = ref.SumTree(), ro.SumTreeOracle()
  oa = ops(ta, lambda t: t.size, lambda t: t.root())
  ob = ops(tb, lambda t: t.size, lambda t: t.root())
  assert oa == ob
  np.testing.assert_array_equal(protocol.f64_bits(ta.values),
                                protocol.f64_bits(tb.leaves()))
  assert ta.root() == tb.root()


def test_oracle_bulk_fill_equals_sequential_adds():
  for cap, n in ((8, 8), (8, 5), (37, 37), (64, 20)):
    mk = lambda: ro.PrioritizedReplayOracle(
        cap, protocol.Item(None, None), 0.5, protocol.beta_schedule(cap), 1e-3,
        True, np.random.RandomState(3))
    a, b = mk(), mk()
    for i in range(n):
      a.add(protocol.Item(i, -i), 2.0)
    b.bulk_fill(n, lambda i: protocol.Item(i, -i), 2.0)
    np.testing.assert_array_equal(a.dist.tree.node, b.dist.tree.node)
    assert a.dist.free == b.dist.free and a.dist.active == b.dist.active
    assert a.dist.where == b.dist.where and a.dist.index_of == b.dist.index_of
    assert a.dist.id_of == b.dist.id_of and a.store.items == b.store.items
    for _ in range(5):
      ia, _, wa = a.sample_ids(4)
      ib, _, wb = b.sample_ids(4)
      np.testing.assert_array_equal(ia, ib)
      a.add(protocol.Item(0, 0), 1.5)
      b.add(protocol.Item(0, 0), 1.5)
    np.testing.assert_array_equal(a.dist.tree.node, b.dist.tree.node)
