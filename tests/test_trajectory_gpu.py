"""Multi-step END-TO-END parity: 10 consecutive learner steps on the device vs
10 steps of the CPU oracle, each side using ITS OWN gradients and its own
optimiser state throughout (nothing is re-synchronised between steps), fresh
batch / weights / noise every step, at the BASELINE shape (A = 6, B = 32) and
with the reference's optimiser settings.

Bounds (stated, and chosen to exclude sign flips of the normalised updates):
  * per-sample losses / TD errors of EVERY step: rtol 1e-5 (+ 2e-6 abs for TD
    errors that cancel to ~0) -- the north-star tolerance holds along the
    trajectory, not just on the first step;
  * after k steps, max |p_device - p_oracle| <= 0.05 * k * lr per tensor.
    Adam and centred RMSProp move a weight by ~lr * sign(g) per step when
    |g| >> eps: one flipped sign is a 2 * lr discrepancy at step 1, i.e. 40x the
    bound; a whole-step error (the round-1 bound of 1.01 * lr) is 20x over it.
"""

import numpy as np
import pytest
import torch

from oracle import qnet_oracle as qo

pytestmark = pytest.mark.gpu

A, B, STEPS = 6, 32, 10
SUPPORT = np.linspace(-10.0, 10.0, 51).astype(np.float32)


def _batch(rs, n_step):
  s_tm1 = rs.randint(0, 256, (B, 84, 84, 4)).astype(np.uint8)
  s_t = rs.randint(0, 256, (B, 84, 84, 4)).astype(np.uint8)
  a = rs.randint(A, size=B).astype(np.int64)
  r = rs.choice([-1.0, 0.0, 1.0], size=B) * rs.uniform(0.5, 2.5, size=B)
  d = rs.choice([0.0, 0.99 ** n_step], size=B)
  return s_tm1, a, r, d, s_t


def _dev(xs):
  return [None if x is None else torch.from_numpy(np.ascontiguousarray(x)).cuda()
          for x in xs]


def _check_params(p_dev, p_orc, k, lr, what):
  worst = 0.0
  for name in p_orc:
    diff = np.abs(p_dev[name] - p_orc[name]).max()
    worst = max(worst, diff / lr)
    assert diff <= 0.05 * k * lr, (what, name, 'step', k, diff / lr)
  return worst


def test_rainbow_trajectory_vs_oracle():
  from dqn_zoo_amd import learner as ll, networks
  rs = np.random.RandomState(77)
  online = qo.init_params('rainbow', A, rs)
  target = qo.init_params('rainbow', A, rs)
  for p in (online, target):
    for k in p:
      if 'sigma' in k:
        p[k] = (p[k] * 3).astype(np.float32)
  opt = ll.AdamConfig()   # rainbow/run_atari.py: lr 6.25e-5, eps 1.5625e-4, clip 10
  ln = ll.RainbowLearner(networks.RainbowNetwork(A, SUPPORT), opt, B, params=online)
  ln.set_params(target, 'target')
  p, st = dict(online), qo.adam_init(online)
  worst = 0.0
  for k in range(1, STEPS + 1):
    batch = _batch(rs, 3)
    w = rs.uniform(0.1, 1.0, size=B).astype(np.float32)
    noises = [qo.sample_noise(rs, A) for _ in range(3)]
    ln.set_noise(noises)
    ln.step(*_dev(batch), torch.from_numpy(w).cuda(), resample_noise=False)
    torch.cuda.synchronize()
    p, st, out = qo.rainbow_update(p, target, st, batch, w, noises, SUPPORT, A,
                                   lr=opt.learning_rate, eps=opt.eps,
                                   max_norm=opt.max_global_grad_norm)
    np.testing.assert_allclose(ln.losses.cpu().numpy(), out['losses'], rtol=1e-5,
                               err_msg='step %d' % k)
    np.testing.assert_allclose(ln.priorities.cpu().numpy(), out['priorities'],
                               rtol=1e-5)
    sc = ln.scalars()
    np.testing.assert_allclose(sc['loss'], out['loss'], rtol=1e-5)
    np.testing.assert_allclose(sc['gnorm'], out['gnorm'], rtol=1e-4)
    worst = max(worst, _check_params(ln.get_params(), p, k, opt.learning_rate,
                                     'rainbow'))
  assert int(ln.adam_count.item()) == STEPS
  m_dev = ln.layout.unpack(ln.adam_m.cpu().numpy())
  for name in st['mu']:
    scale = np.abs(st['mu'][name]).max()
    assert np.abs(m_dev[name] - st['mu'][name]).max() <= 2e-4 * scale, name
  print('rainbow: worst |dp| / lr over %d steps = %.4f' % (STEPS, worst))


@pytest.mark.parametrize('kind', ['dqn', 'prioritized'])
def test_dqn_rmsprop_trajectory_vs_oracle(kind):
  """BASELINE configs[1] (DQN) and configs[2] (double-Q + importance weights)
  with their run_atari.py optimiser settings."""
  from dqn_zoo_amd import learner as ll, networks
  rs = np.random.RandomState(78 + len(kind))
  net = 'dqn' if kind == 'dqn' else 'double_dqn'
  online = qo.init_params(net, A, rs)
  target = qo.init_params(net, A, rs)
  if kind == 'dqn':   # dqn/run_atari.py:78-83
    opt = ll.RmsPropConfig(learning_rate=0.00025, decay=0.95, eps=0.01 / 32 ** 2)
  else:               # prioritized/run_atari.py:84-88
    opt = ll.RmsPropConfig(learning_rate=0.00025 / 4, decay=0.95,
                           eps=(0.01 / 32 ** 2) * (1.0 / 4) ** 2)
  ln = ll.DenseLearner(networks.DenseNetwork(net, A),
                       'q' if kind == 'dqn' else 'double_q', opt, B,
                       grad_error_bound=1.0 / 32, params=online)
  ln.set_params(target, 'target')
  p, st = dict(online), qo.rmsprop_init(online)
  worst = 0.0
  for k in range(1, STEPS + 1):
    batch = _batch(rs, 1)
    w = rs.uniform(0.1, 1.0, size=B).astype(np.float32) if kind != 'dqn' else None
    ln.step(*_dev(batch), None if w is None else torch.from_numpy(w).cuda())
    torch.cuda.synchronize()
    _, td, grads, _ = qo.dqn_family_loss_and_grads(kind, p, target, batch, w, 1.0 / 32)
    p, st = qo.rmsprop_centered_update(p, grads, st, opt.learning_rate, opt.decay,
                                       opt.eps)
    np.testing.assert_allclose(ln.losses.cpu().numpy(), td, rtol=1e-5, atol=2e-6,
                               err_msg='step %d' % k)
    np.testing.assert_allclose(ln.priorities.cpu().numpy(), np.abs(td), rtol=1e-5,
                               atol=2e-6)
    assert (np.abs(td) > 1.0).any() and (np.abs(td) < 1.0).any()
    worst = max(worst, _check_params(ln.get_params(), p, k, opt.learning_rate, kind))
  print('%s: worst |dp| / lr over %d steps = %.4f' % (kind, STEPS, worst))
