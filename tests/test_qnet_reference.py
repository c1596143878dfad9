"""The Q-loss / update half against the REAL reference, when somebody has produced the
files: tools/gen_reference_qnet_fixtures.py runs the unmodified dqn_zoo agents' own
`update` closures (jax 0.3.10 / haiku 0.0.6 / optax 0.1.2 / rlax 0.1.2,
docker_requirements.txt:6-16) on the seeded cases of tests/golden/qnet_cases.py and writes
tests/golden/ref_qnet_<agent>.npz.  None of those packages can be installed in the build
container, so until the files exist every test here SKIPS with the reason UNPINNED -- the
half's parity status -- and the command that pins it.

With the files present:
  * CPU: the float64 oracle on the same inputs (Rainbow: with the noise the reference
    network actually used, stored in the file) reproduces the reference's float32 numbers --
    per-sample losses to 2e-6, every gradient tensor to 1e-5 of its scale (the reference
    accumulates in float32: its own rounding is ~1e-6 of scale), one optimiser step;
  * GPU: the HIP learners against the same files at the north-star 1e-5 (the bounds of
    tests/test_qnet_golden_gpu.py).
"""

import os

import numpy as np
import pytest

from tests.golden import qnet_cases as qc

GOLDEN = os.path.join(os.path.dirname(__file__), 'golden')
HOW = ('UNPINNED: tests/golden/ref_qnet_%s.npz is absent.  In an environment with '
       'docker_requirements.txt:6-16 run  python tools/gen_reference_qnet_fixtures.py '
       '--dqn_zoo <checkout>  and commit the files.')


def _load(name):
  path = os.path.join(GOLDEN, 'ref_qnet_%s.npz' % name)
  if not os.path.exists(path):
    pytest.skip(HOW % name)
  g = np.load(path)
  inp = qc.make_inputs(name)          # float64 copies of the same seeded inputs
  inp32 = qc.make_inputs(name, np.float32)
  for k in ('online', 'target'):      # the reference saw the float32 values
    inp[k] = {n: v.astype(np.float64) for n, v in inp32[k].items()}
  if inp['weights'] is not None:
    inp['weights'] = inp32['weights'].astype(np.float64)
  if name == 'rainbow':               # ... and the noise its network formed
    inp['noises'] = [{n: g['noise/%d/%s' % (i, n)].astype(np.float64)
                      for n in inp['noises'][i]} for i in range(3)]
  if name == 'iqn':
    inp['taus'] = [t.astype(np.float64) for t in inp32['taus']]
  return g, inp


@pytest.mark.parametrize('name', sorted(qc.CASES))
def test_oracle_vs_reference(name):
  g, inp = _load(name)
  c = inp['case']
  res = qc.oracle_step(name, inp)
  if g['losses'].size:
    np.testing.assert_allclose(res['losses'], g['losses'], rtol=2e-6, atol=1e-6)
  np.testing.assert_allclose(res['gnorm'], g['gnorm'], rtol=1e-5)
  for k, v in res['grads'].items():
    scale = g['gstat/' + k][2]
    assert np.abs(qc.sample_tensor(v) - g['g/' + k]).max() <= 1e-5 * scale, k
    np.testing.assert_allclose(np.sqrt((v * v).sum()), g['gstat/' + k][1], rtol=1e-5, err_msg=k)
    assert np.abs(qc.sample_tensor(res['params'][k]) - g['p/' + k]).max() <= 0.01 * c['lr'], k
    np.testing.assert_allclose(qc.sample_tensor(res['opt']['m'][k]), g['m/' + k], rtol=1e-4,
                               atol=1e-6 * scale, err_msg=k)
    np.testing.assert_allclose(qc.sample_tensor(res['opt']['v'][k]), g['v/' + k], rtol=2e-4,
                               atol=1e-6 * scale * scale, err_msg=k)


@pytest.mark.gpu
@pytest.mark.parametrize('name', sorted(qc.CASES))
def test_hip_learner_vs_reference(name):
  from dqn_zoo_amd import _lib
  from tests import test_qnet_golden_gpu as gg
  g, inp = _load(name)
  c = inp['case']
  ln = gg._learner(name, inp)   # pylint: disable=protected-access
  if hasattr(ln, 'keep_all_grads'):
    ln.keep_all_grads = True
  gg._step(name, ln, inp, _lib.PHASE_FORWARD | _lib.PHASE_BACKWARD)   # pylint: disable=protected-access
  if g['losses'].size:
    np.testing.assert_allclose(ln.losses.cpu().numpy(), g['losses'], rtol=1e-5, atol=2e-6)
  grads = ln.layout.unpack(ln.grad.cpu().numpy())
  for k, v in grads.items():
    scale = g['gstat/' + k][2]
    assert np.abs(qc.sample_tensor(v) - g['g/' + k]).max() / scale < 1e-4, k
  ln2 = gg._learner(name, inp)   # pylint: disable=protected-access
  gg._step(name, ln2, inp, _lib.PHASE_ALL)   # pylint: disable=protected-access
  p = ln2.get_params()
  for k in p:
    assert np.abs(qc.sample_tensor(p[k]) - g['p/' + k]).max() <= 0.04 * c['lr'], k


def test_generator_selfcheck():
  """Everything of the generator that does not need JAX: haiku-name mapping for the seven
  networks, parameter-tree round trip, noise inversion, pack()."""
  import importlib.util
  spec = importlib.util.spec_from_file_location(
      'gen_ref', os.path.join(os.path.dirname(GOLDEN), '..', 'tools',
                              'gen_reference_qnet_fixtures.py'))
  mod = importlib.util.module_from_spec(spec)
  spec.loader.exec_module(mod)
  mod.selfcheck()
