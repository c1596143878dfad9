"""Device-side insert path (dz_replay_insert): one launch
writes a transition's rows from HBM-resident observations / immediates and its
sum-tree leaf.  Checked against the host-array path and the replay oracle."""

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _transition(rs, T):
  return T(s_tm1=rs.randint(0, 256, (84, 84, 4)).astype(np.uint8),
           a_tm1=int(rs.randint(6)), r_t=float(rs.choice([-1.0, 0.0, 1.0])),
           discount_t=float(rs.choice([0.0, 0.99 ** 3])),
           s_t=rs.randint(0, 256, (84, 84, 4)).astype(np.uint8))


def test_insert_device_rows_equal_host_rows_and_oracle_tree():
  from dqn_zoo_amd import parts
  from dqn_zoo_amd import replay as rl
  from oracle import replay_oracle as ro
  cap = 37
  T = rl.Transition
  beta = parts.LinearSchedule(begin_t=0, end_t=100, begin_value=0.4, end_value=1.0)
  mk = lambda: rl.PrioritizedTransitionReplay(cap, T(None, None, None, None, None), 0.5,
                                              beta, 1e-3, True, np.random.RandomState(3))
  host, dev = mk(), mk()
  orc = ro.PrioritizedReplayOracle(cap, T(None, None, None, None, None), 0.5, beta, 1e-3,
                                   True, np.random.RandomState(3))
  rs = np.random.RandomState(0)
  for i in range(3 * cap + 5):            # wraps around: evictions included
    tr = _transition(rs, T)
    host.add_with_device_priority(tr)
    dev.add_with_device_priority(tr._replace(
        s_tm1=torch.from_numpy(tr.s_tm1).cuda(), s_t=torch.from_numpy(tr.s_t).cuda()))
    orc.add(tr, 1.0)
  torch.cuda.synchronize()
  host.check_status(); dev.check_status()
  orc_ids = sorted(orc.dist.index_of)
  assert sorted(dev.ids()) == sorted(host.ids()) == orc_ids
  for a, b in zip(host._ring.fields, dev._ring.fields):   # pylint: disable=protected-access
    assert torch.equal(a, b)
  want = orc.store.stack(orc_ids[:5], T(None, None, None, None, None))
  got = list(dev.get(orc_ids[:5]))
  for j, g in enumerate(got):
    for x, y in zip(g, want):
      np.testing.assert_array_equal(np.asarray(x), np.asarray(y)[j])
    assert np.asarray(g.a_tm1).dtype == np.int64 and np.asarray(g.r_t).dtype == np.float64
  np.testing.assert_array_equal(dev.tree_storage.cpu().numpy(), host.tree_storage.cpu().numpy())
  np.testing.assert_array_equal(dev.tree_storage.cpu().numpy()[:2 * 64], orc.dist.tree.node)
  with pytest.raises(ValueError):          # wrong dtype on the device path is refused
    dev.add_with_device_priority(_transition(rs, T)._replace(
        s_t=torch.zeros((84, 84, 4), dtype=torch.float32, device='cuda')))


def test_uniform_replay_insert_and_observation_cache():
  from dqn_zoo_amd import device_obs
  from dqn_zoo_amd import replay as rl
  T = rl.Transition
  rep = rl.TransitionReplay(16, T(None, None, None, None, None), np.random.RandomState(1))
  cache = device_obs.ObservationCache(torch.device('cuda', 0), depth=4)
  rs = np.random.RandomState(5)
  obs = [rs.randint(0, 256, (84, 84, 4)).astype(np.uint8) for _ in range(7)]
  for t in range(1, 7):
    d = cache.upload(obs[t])
    assert d.shape == (1, 84, 84, 4)
    torch.cuda.synchronize()
    tr = T(obs[t - 1], t, float(t), 0.5, obs[t])
    trd = cache.on_device(tr)
    if t >= 2:   # both observations were uploaded by earlier steps
      assert isinstance(trd.s_tm1, torch.Tensor) and isinstance(trd.s_t, torch.Tensor)
    else:        # obs[0] never went through the cache: host path for that field
      assert trd.s_tm1 is obs[0] and isinstance(trd.s_t, torch.Tensor)
    rep.add(trd)
  got = list(rep.get(sorted(rep.ids())))
  for t, g in zip(range(1, 7), got):
    np.testing.assert_array_equal(g.s_tm1, obs[t - 1])
    np.testing.assert_array_equal(g.s_t, obs[t])
    assert g.a_tm1 == t and g.r_t == float(t) and g.discount_t == 0.5
  assert cache.lookup(obs[1]) is None        # evicted from the 4-deep ring
  assert cache.lookup(obs[6]) is not None
  assert cache.lookup(obs[6].copy()) is None  # identity, not equality


@pytest.mark.parametrize('cap,n_add', [(23, 10), (23, 23), (23, 61), (1, 4)])
def test_uniform_sample_one_launch_equals_two_step_path_and_oracle(cap, n_add):
  """dz_replay_sample_uniform (positions in kernel arguments -> ids -> gather)
  against the pos->id kernel + gather path and the oracle's id list, while
  filling, exactly full and after wrap-around (ref: replay.py:52-82, 152-163)."""
  from dqn_zoo_amd import replay as rl
  from oracle import replay_oracle as ro
  T = rl.Transition
  a = rl.TransitionReplay(cap, T(None, None, None, None, None), np.random.RandomState(9))
  b = rl.TransitionReplay(cap, T(None, None, None, None, None), np.random.RandomState(9))
  o = ro.UniformReplayOracle(cap, T(None, None, None, None, None), np.random.RandomState(9))
  rs = np.random.RandomState(1)
  for _ in range(n_add):
    tr = _transition(rs, T)
    a.add(tr); b.add(tr); o.add(tr)
  for _ in range(6):
    outs, ids = a.sample_device(8)            # one launch
    ids2 = b.sample_ids_device(8)             # reference-order two-step path
    outs2 = b._ring.gather(ids2, 8, b._stream())  # pylint: disable=protected-access
    want = o.sample_ids(8)
    torch.cuda.synchronize()
    np.testing.assert_array_equal(ids.cpu().numpy(), ids2.cpu().numpy())
    np.testing.assert_array_equal(ids.cpu().numpy(), np.asarray(want))
    for x, y in zip(outs, outs2):
      assert torch.equal(x, y)


@pytest.mark.parametrize('batch', [32, 64, 100])
def test_prioritized_sample_device_both_draw_paths_match_oracle(batch):
  """sample_device(): draws in kernel arguments (batch <= 64) and the pinned
  async-copy path (batch > 64) give the oracle's ids and probabilities."""
  from dqn_zoo_amd import parts
  from dqn_zoo_amd import replay as rl
  from oracle import replay_oracle as ro
  cap = 257
  T = rl.Transition
  beta = parts.LinearSchedule(begin_t=0, end_t=100, begin_value=0.4, end_value=1.0)
  dev = rl.PrioritizedTransitionReplay(cap, T(None, None, None, None, None), 0.5, beta, 1e-3,
                                       True, np.random.RandomState(17))
  orc = ro.PrioritizedReplayOracle(cap, T(None, None, None, None, None), 0.5, beta, 1e-3, True,
                                   np.random.RandomState(17))
  rs = np.random.RandomState(2)
  for i in range(cap + 30):
    tr = T(rs.randint(0, 256, (84, 84, 4)).astype(np.uint8), int(rs.randint(6)), 1.0, 0.99,
           rs.randint(0, 256, (84, 84, 4)).astype(np.uint8))
    p = 0.5 + (i % 11)
    dev.add(tr, p); orc.add(tr, p)
  for _ in range(4):
    s = dev.sample_device(batch)
    ids, probs = orc.dist.sample(batch)
    torch.cuda.synchronize()
    np.testing.assert_array_equal(s.ids.cpu().numpy(), np.asarray(ids))
    np.testing.assert_array_equal(s.probabilities.cpu().numpy(), np.asarray(probs))
  dev.check_status()
