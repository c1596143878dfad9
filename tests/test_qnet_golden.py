"""Frozen fixtures of the Q-loss / update half (tests/golden/qnet_*.npz).

Three independent things must agree on every case (seven agents + the Cramer
projection): the committed FILES, the CPU oracle as it is today, and the torch-
autograd float64 models of tests/torch_models.py.  The arithmetic of this half
lives in rlax/optax/haiku/jax (absent): this does not pin it to the reference
("parity unpinned") -- it pins the oracle to a file and to a second
implementation, so that the target the HIP kernels are compared with cannot
drift."""

import os

import numpy as np
import pytest
import torch

from oracle import qnet_oracle as qo
from tests import torch_models as tm
from tests.golden import qnet_cases as qc

GOLDEN = os.path.join(os.path.dirname(__file__), 'golden')


def _load(name):
  return np.load(os.path.join(GOLDEN, 'qnet_%s.npz' % name))


@pytest.mark.parametrize('name', sorted(qc.CASES))
def test_oracle_reproduces_fixture(name):
  g = _load(name)
  got = qc.pack(qc.oracle_step(name, qc.make_inputs(name)))
  assert set(got) == set(g.files)
  for k in g.files:
    np.testing.assert_allclose(got[k], g[k], rtol=1e-12, atol=1e-300, err_msg=k)


@pytest.mark.parametrize('name', sorted(qc.CASES))
def test_torch_autograd_model_reproduces_fixture(name):
  g = _load(name)
  inp = qc.make_inputs(name)
  report, loss, grads = tm.learner_step(name, inp, qc.SUPPORT, qc.QUANTILES, qc.A)
  np.testing.assert_allclose(report, g['losses'], rtol=1e-9, atol=1e-12)
  np.testing.assert_allclose(loss, g['loss'], rtol=1e-9)
  new_p, opt, gnorm = tm.optimizer_step(inp['case'], inp['online'], grads)
  np.testing.assert_allclose(gnorm, g['gnorm'], rtol=1e-9)
  assert set(grads) == set(inp['online'])
  for k in grads:
    scale = g['gstat/' + k][2]
    assert scale > 0, k
    assert np.abs(qc.sample_tensor(grads[k]) - g['g/' + k]).max() / scale < 1e-9, k
    st = np.array([grads[k].sum(), np.sqrt((grads[k] ** 2).sum()),
                   np.abs(grads[k]).max()])
    np.testing.assert_allclose(st[1:], g['gstat/' + k][1:], rtol=1e-9, err_msg=k)
    assert abs(st[0] - g['gstat/' + k][0]) <= 1e-9 * st[1] * np.sqrt(grads[k].size), k
    # one optimiser step: the update itself (p_new - p_old) to 1e-7 of the step
    lr = inp['case']['lr']
    dp = qc.sample_tensor(new_p[k]) - g['p/' + k]
    assert np.abs(dp).max() <= 1e-7 * lr, (k, np.abs(dp).max())
    np.testing.assert_allclose(qc.sample_tensor(opt['m'][k]), g['m/' + k],
                               rtol=1e-7, atol=1e-9 * scale, err_msg=k)
    np.testing.assert_allclose(qc.sample_tensor(opt['v'][k]), g['v/' + k],
                               rtol=1e-7, atol=1e-9 * scale * scale, err_msg=k)


@pytest.mark.parametrize('name', sorted(qc.CASES))
def test_float32_oracle_is_within_the_north_star_tolerance_of_the_fixture(name):
  """The float32 oracle (what the HIP kernels are compared with at the full
  batch size) against the float64 fixture: per-sample losses within 1e-5."""
  g = _load(name)
  inp = qc.make_inputs(name)
  f32 = lambda t: {k: v.astype(np.float32) for k, v in t.items()}
  inp32 = dict(inp, online=f32(inp['online']), target=f32(inp['target']))
  if 'noises' in inp:
    inp32['noises'] = [f32(n) for n in inp['noises']]
  if 'taus' in inp:
    inp32['taus'] = [t.astype(np.float32) for t in inp['taus']]
  out = qc.oracle_step(name, inp32, np.float32)
  np.testing.assert_allclose(out['losses'], g['losses'], rtol=1e-5, atol=2e-6)
  np.testing.assert_allclose(out['loss'], g['loss'], rtol=1e-5)


def test_projection_fixture():
  g = np.load(os.path.join(GOLDEN, 'qnet_projection.npz'))['m']
  zs = torch.from_numpy(qc.SUPPORT)
  for i, (zp, p) in enumerate(qc.projection_cases()):
    np.testing.assert_allclose(qo.categorical_l2_project(zp, p, qc.SUPPORT), g[i],
                               rtol=1e-12, atol=1e-300)
    np.testing.assert_allclose(
        tm.project(torch.from_numpy(zp), torch.from_numpy(p), zs).numpy(), g[i],
        rtol=1e-9, atol=1e-13)
    assert abs(g[i].sum() - 1.0) < 1e-12
