"""HIP learners vs the FROZEN fixtures tests/golden/qnet_*.npz (float64 oracle
outputs, independently reproduced by a torch-autograd model in the CPU suite):
per-sample losses / TD errors within the north-star 1e-5, every gradient tensor
within 1e-4 of its scale, and one optimiser step with the reference's
hyper-parameters -- all seven agents, through the C ABI."""

import os

import numpy as np
import pytest
import torch

from tests.golden import qnet_cases as qc

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(__file__), 'golden')


def _f32(t):
  return {k: np.asarray(v, np.float32) for k, v in t.items()}


def _dev(xs):
  return [None if x is None else torch.from_numpy(np.ascontiguousarray(x)).cuda()
          for x in xs]


def _learner(name, inp):
  from dqn_zoo_amd import learner as ll, networks
  c = inp['case']
  on, tg = _f32(inp['online']), _f32(inp['target'])
  if c['opt'] == 'adam':
    opt = ll.AdamConfig(learning_rate=c['lr'], eps=c['eps'],
                        max_global_grad_norm=c['max_norm'])
  else:
    opt = ll.RmsPropConfig(learning_rate=c['lr'], decay=c['decay'], eps=c['eps'])
  if name == 'rainbow':
    ln = ll.RainbowLearner(networks.RainbowNetwork(qc.A, qc.SUPPORT.astype(np.float32)),
                           opt, qc.B, params=on)
    ln.set_noise([_f32(n) for n in inp['noises']])
  elif name == 'iqn':
    ln = ll.IqnLearner(networks.IqnNetwork(qc.A, 64), opt, qc.B,
                       tau_samples=qc.IQN_TAUS, huber_param=c['kappa'], params=on)
  else:
    loss = {'dqn': 'q', 'double_q': 'double_q', 'prioritized': 'double_q',
            'c51': 'categorical', 'qr': 'quantile'}[name]
    net = networks.DenseNetwork(c['net'], qc.A, support=qc.SUPPORT.astype(np.float32),
                                quantiles=qc.QUANTILES.astype(np.float32))
    ln = ll.DenseLearner(net, loss, opt, qc.B, grad_error_bound=c.get('bound', 1 / 32),
                         huber_param=c.get('kappa', 1.0), params=on)
  ln.set_params(tg, 'target')
  return ln


def _step(name, ln, inp, phases):
  s_tm1, a, r, d, s_t = inp['batch']
  dev = _dev((s_tm1, a, r, d, s_t))
  if name == 'rainbow':
    w = torch.from_numpy(inp['weights'].astype(np.float32)).cuda()
    ln.step(*dev, w, phases=phases, resample_noise=False)
  elif name == 'iqn':
    ln.step(*dev, taus=_dev([t.astype(np.float32) for t in inp['taus']]), phases=phases)
  else:
    w = None if inp['weights'] is None else torch.from_numpy(
        inp['weights'].astype(np.float32)).cuda()
    ln.step(*dev, w, phases=phases)
  torch.cuda.synchronize()


@pytest.mark.parametrize('name', sorted(qc.CASES))
def test_learner_vs_frozen_fixture(name):
  from dqn_zoo_amd import _lib
  g = np.load(os.path.join(GOLDEN, 'qnet_%s.npz' % name))
  inp = qc.make_inputs(name)
  c = inp['case']
  ln = _learner(name, inp)
  if hasattr(ln, 'keep_all_grads'):
    ln.keep_all_grads = True
  _step(name, ln, inp, _lib.PHASE_FORWARD | _lib.PHASE_BACKWARD)
  np.testing.assert_allclose(ln.losses.cpu().numpy(), g['losses'], rtol=1e-5, atol=2e-6)
  grads = ln.layout.unpack(ln.grad.cpu().numpy())
  assert set(grads) == {k[2:] for k in g.files if k.startswith('g/')}
  for k, v in grads.items():
    scale = g['gstat/' + k][2]
    err = np.abs(qc.sample_tensor(v) - g['g/' + k]).max() / scale
    assert err < 1e-4, (k, err)
    l2 = np.sqrt((v.astype(np.float64) ** 2).sum())
    np.testing.assert_allclose(l2, g['gstat/' + k][1], rtol=1e-4, err_msg=k)
  # one full step from the same state: parameters and optimiser moments
  ln2 = _learner(name, inp)
  _step(name, ln2, inp, _lib.PHASE_ALL)
  p = ln2.get_params()
  st = ln2.get_opt_state()
  assert st['count'] == (1 if c['opt'] == 'adam' else st['count'])
  for k in p:
    scale = g['gstat/' + k][2]
    # a sign flip of the (normalised) update would be ~2 lr: excluded by 50x
    dp = np.abs(qc.sample_tensor(p[k]) - g['p/' + k]).max()
    assert dp <= 0.04 * c['lr'], (k, dp / c['lr'])
    np.testing.assert_allclose(qc.sample_tensor(st['mu'][k]), g['m/' + k], rtol=2e-4,
                               atol=1e-5 * scale, err_msg=k)
    np.testing.assert_allclose(qc.sample_tensor(st['nu'][k]), g['v/' + k], rtol=4e-4,
                               atol=1e-5 * scale * scale, err_msg=k)
