"""The device noise / tau generators as STREAMS, not just as moments.

The reference draws its noise from JAX's threefry keys (one key per network apply:
rainbow/agent.py:87, networks.py:139-150; three tau sets per IQN step:
iqn/agent.py:182-187), which cannot be reproduced without JAX; parity tests therefore
inject noise.  What CAN be pinned, and what an injected-noise test never sees, is that
the device generator (a) hands every apply of every step its own, non-overlapping
range of one counter-based stream, (b) has the reference's distribution.
"""

import ctypes

import numpy as np
import pytest
import scipy.stats
import torch

pytestmark = pytest.mark.gpu

SUP = np.linspace(-10, 10, 51).astype(np.float32)


def _fill(lib, n, seed, counter):
  from dqn_zoo_amd import _lib
  out = torch.empty(n, dtype=torch.float32, device='cuda')
  _lib.check(lib.dz_noise_fill(out.data_ptr(), n, seed, counter, None), 'dz_noise_fill')
  torch.cuda.synchronize()
  return out.cpu().numpy()


def _batch(rs, A, B):
  dev = 'cuda'
  return (torch.from_numpy(rs.randint(0, 256, (B, 84, 84, 4)).astype(np.uint8)).to(dev),
          torch.from_numpy(rs.randint(0, A, B).astype(np.int64)).to(dev),
          torch.from_numpy(rs.randint(-1, 2, B).astype(np.float64)).to(dev),
          torch.from_numpy((rs.randint(0, 2, B) * 0.97).astype(np.float64)).to(dev),
          torch.from_numpy(rs.randint(0, 256, (B, 84, 84, 4)).astype(np.uint8)).to(dev),
          torch.from_numpy(rs.uniform(0.3, 1.0, B).astype(np.float32)).to(dev))


def test_learner_noise_stream_positions():
  """Step c (optimiser count c before the step) uses EXACTLY positions
  [0x5eed + 3 c stride, 0x5eed + 3 (c+1) stride) of the (seed) stream: block g of the
  three applies is the g-th third.  Consecutive steps and the three applies of a step
  therefore never share a stream position; and no two of the 3 x 6 noise vectors of a
  step are equal or correlated."""
  from dqn_zoo_amd import _lib, learner as ll, networks
  A, B, seed = 4, 8, 12345
  lib = _lib.load()
  ln = ll.RainbowLearner(networks.RainbowNetwork(A, SUP, 0.1), ll.AdamConfig(), B, seed=seed)
  ln.use_graphs = False
  st = int(ln.layout.noise_stride)
  rs = np.random.RandomState(0)
  seen = []
  for c in range(3):
    ln.step(*_batch(rs, A, B))
    torch.cuda.synchronize()
    got = ln.noise.cpu().numpy()
    want = _fill(lib, 3 * st, seed, 0x5eed + c * 3 * st)
    np.testing.assert_array_equal(got, want)
    seen.append(got)
    assert int(ln.adam_count.item()) == c + 1
  # a step's blocks / consecutive steps: different ranges of one stream -> different values
  allv = np.concatenate(seen)
  blocks = allv.reshape(9, st)
  for i in range(9):
    for j in range(i + 1, 9):
      assert (blocks[i] != blocks[j]).mean() > 0.99
      assert abs(np.corrcoef(blocks[i], blocks[j])[0, 1]) < 0.05
  # the 6 vectors of one apply (eps_in / eps_out of the four noisy layers;
  # networks.py:151-176) are distinct slices of that range
  L = ln.layout.c
  offs = [int(L.n_adv1_in), int(L.n_val1_in), int(L.n_fc1_out), int(L.n_adv2_in),
          int(L.n_val2_in), int(L.n_fc2_out), st]
  assert offs == sorted(offs) and len(set(offs)) == len(offs)
  vecs = [blocks[0][offs[k]:offs[k + 1]] for k in range(6)]
  for i in range(6):
    for j in range(i + 1, 6):
      m = min(len(vecs[i]), len(vecs[j]))
      assert abs(np.corrcoef(vecs[i][:m], vecs[j][:m])[0, 1]) < 0.15


def test_acting_noise_is_a_separate_stream():
  """The actor's applies (rainbow/agent.py:171-179: a fresh key per decision) draw block
  k at positions [k stride, (k+1) stride) of ANOTHER seed's stream -- never a position
  of the learner's stream with the same seed -- and advance by themselves."""
  from dqn_zoo_amd import _lib, learner as ll, networks
  A, seed = 4, 777
  lib = _lib.load()
  ln = ll.RainbowLearner(networks.RainbowNetwork(A, SUP, 0.1), ll.AdamConfig(), 8, seed=seed)
  st = int(ln.layout.noise_stride)
  s = torch.zeros((1, 84, 84, 4), dtype=torch.uint8, device='cuda')
  for k in range(3):
    ln.apply(s)
    torch.cuda.synchronize()
    got = ln._act_noise.cpu().numpy()  # pylint: disable=protected-access
    np.testing.assert_array_equal(got, _fill(lib, st, seed ^ 0xA5A5A5A5, k * st))
    assert ln.act_step() == k + 1


def test_noise_distribution_is_f_of_truncated_normal():
  """x = sign(e) e^2 for the generated e must be the truncated normal on [-2, 2]
  (networks.py:142-144: f(x) = sign(x) sqrt|x| of jax.random.truncated_normal(-2, 2)):
  Kolmogorov-Smirnov at n = 2e5, plus the exact support."""
  from dqn_zoo_amd import _lib
  lib = _lib.load()
  n = 200000
  e = _fill(lib, n, 99, 12345).astype(np.float64)
  x = np.sign(e) * e * e
  assert np.abs(x).max() <= 2.0 and np.abs(e).max() <= np.sqrt(2.0) + 1e-6
  d, p = scipy.stats.kstest(x, scipy.stats.truncnorm(-2.0, 2.0).cdf)
  assert p > 1e-3, (d, p)
  assert d < 0.005, d
  # consecutive stream positions are independent: lag-1 autocorrelation
  assert abs(np.corrcoef(x[:-1], x[1:])[0, 1]) < 0.01


def test_iqn_tau_sets_are_uniform_and_disjoint():
  """iqn/agent.py:182-187: three tau sets per step from three keys, U[0, 1).  Here: one
  counter-based stream, step c takes [c n, (c+1) n) with n = B (N0 + N1 + N2), the three
  sets are consecutive slices of it; KS against U(0, 1)."""
  from dqn_zoo_amd import _lib, learner as ll, networks
  lib = _lib.load()
  A, B, seed = 4, 8, 4242
  net = networks.IqnNetwork(A, 64)
  ln = ll.IqnLearner(net, ll.AdamConfig(learning_rate=5e-5, eps=0.01 / 32), B,
                     tau_samples=(16, 8, 8), seed=seed)
  ln.use_graphs = False
  n = ln.taus.numel()
  rs = np.random.RandomState(1)
  seen = []
  for c in range(3):
    b = _batch(rs, A, B)
    ln.step(*b[:5])
    torch.cuda.synchronize()
    got = ln.taus.cpu().numpy().copy()
    want = torch.empty(n, dtype=torch.float32, device='cuda')
    _lib.check(lib.dz_uniform_fill(want.data_ptr(), n, seed, c * n, None, None), 'dz_uniform_fill')
    torch.cuda.synchronize()
    np.testing.assert_array_equal(got, want.cpu().numpy())
    seen.append(got)
  assert (seen[0] != seen[1]).mean() > 0.99 and (seen[1] != seen[2]).mean() > 0.99
  t0, t1, t2 = (ln.tau_tm1.cpu().numpy().ravel(), ln.tau_sel.cpu().numpy().ravel(),
                ln.tau_t.cpu().numpy().ravel())
  assert len(t0) + len(t1) + len(t2) == n and (t1 != t2).mean() > 0.99
  big = torch.empty(200000, dtype=torch.float32, device='cuda')
  _lib.check(lib.dz_uniform_fill(big.data_ptr(), big.numel(), 5, 0, None, None), 'dz_uniform_fill')
  torch.cuda.synchronize()
  u = big.cpu().numpy().astype(np.float64)
  assert u.min() >= 0.0 and u.max() < 1.0
  d, p = scipy.stats.kstest(u, 'uniform')
  assert p > 1e-3 and d < 0.005, (d, p)
  assert abs(np.corrcoef(u[:-1], u[1:])[0, 1]) < 0.01
