"""Rainbow's head chain as ONE multi-role launch (csrc/dz_head_chain.h) is BIT-IDENTICAL to the
four launches it replaces (fc1 epilogue, noisy fc2, loss, fc2 backward), and fails loudly.

ref: networks.py:239-258 (the dueling head's two noisy layers), rainbow/agent.py:97-118 (loss and
its gradient).  The comparison treats -0.0 and +0.0 as equal (seams carry -0.0f for a zero)."""

import numpy as np
import pytest
import torch

from dqn_zoo_amd import _lib
from tests.test_rainbow_gpu import _dev_batch, _learner, _problem

pytestmark = pytest.mark.gpu


def _run(A, B, seed, separate, steps=3, sink=False):
  online, target, batch, w, noises = _problem(A, B, seed)
  ln = _learner(A, B, online, target, noises)
  ln.use_graphs = False
  ln.separate_launches = separate
  dev = _dev_batch(batch, w)
  out = []
  L = ln.layout
  ld2 = L.ld2
  for i in range(steps):
    # (the same explicit noise every step: resample_noise=False keeps set_noise's blocks)
    ln.step(*dev, resample_noise=(i > 0))
    torch.cuda.synchronize()
    snap = dict(
        losses=ln.losses.cpu().numpy().copy(), priorities=ln.priorities.cpu().numpy().copy(),
        h1=ln.ws_view('h1', 3 * B * 1024).cpu().numpy().copy(),
        fc2_part=ln.ws_view('fc2_part', 4 * 3 * B * ld2).cpu().numpy().copy(),
        fc2_out=ln.ws_view('fc2_out', 3 * B * ld2).cpu().numpy().copy(),
        dout2=ln.ws_view('dout2', B * ld2).cpu().numpy().copy(),
        dh1=ln.ws_view('dh1', B * 1024).cpu().numpy().copy(),
        q_sel=ln.ws_view('q_sel', B * A).cpu().numpy().copy(),
        target_probs=ln.ws_view('target_probs', B * 51).cpu().numpy().copy(),
        grad=ln.grad.cpu().numpy().copy(), online=ln.online.cpu().numpy().copy(),
        adam_m=ln.adam_m.cpu().numpy().copy(), adam_v=ln.adam_v.cpu().numpy().copy(),
        scalars=ln.scalars())
    out.append(snap)
  return out


@pytest.mark.parametrize('A,B,seed', [(6, 32, 0), (18, 32, 1), (3, 10, 2), (4, 1, 3), (9, 7, 4)])
def test_multi_role_head_launch_is_bit_identical_to_the_four_launches(A, B, seed):
  a = _run(A, B, seed, separate=True)
  b = _run(A, B, seed, separate=False)
  for step, (sa, sb) in enumerate(zip(a, b)):
    assert not sa['scalars']['chain_failed'] and not sb['scalars']['chain_failed']
    bad = []
    for k in ('h1', 'fc2_part', 'fc2_out', 'losses', 'priorities', 'q_sel', 'target_probs', 'dout2', 'dh1',
              'grad', 'online', 'adam_m', 'adam_v'):
      assert np.isfinite(sb[k]).all(), k
      if not np.array_equal(sa[k], sb[k]):                       # (-0.0 == +0.0)
        d = np.flatnonzero(sa[k] != sb[k])
        bad.append('%s: %d of %d differ, first at %d (%r vs %r)' % (
            k, d.size, sa[k].size, d[0], sa[k].flat[d[0]], sb[k].flat[d[0]]))
    assert not bad, 'step %d: %s' % (step, '; '.join(bad))
    assert sa['scalars']['gnorm'] == sb['scalars']['gnorm'] and sa['scalars']['loss'] == sb['scalars']['loss']
  assert np.abs(a[0]['dh1']).max() > 0 and np.abs(a[-1]['online'] - a[0]['online']).max() > 0


def test_step_with_stale_seam_buffers_is_still_correct():
  """The seam buffers are cleared by the step's own first launch: garbage left in them by
  anything else (an abandoned step, another shape of the call) cannot be taken for data."""
  A, B = 6, 32
  ref = _run(A, B, 5, separate=True, steps=1)[0]
  online, target, batch, w, noises = _problem(A, B, 5)
  ln = _learner(A, B, online, target, noises)
  ln.use_graphs = False
  for name, n in (('h1', 3 * B * 1024), ('fc2_part', 4 * 3 * B * ln.layout.ld2), ('dout2', B * ln.layout.ld2)):
    ln.ws_view(name, n).fill_(7.25)
  ln.step(*_dev_batch(batch, w), resample_noise=False)
  torch.cuda.synchronize()
  np.testing.assert_array_equal(ln.losses.cpu().numpy(), ref['losses'])
  np.testing.assert_array_equal(ln.online.cpu().numpy(), ref['online'])


def test_forced_seam_timeout_in_the_head_chain_is_loud():
  """dz_act_debug_spin_limit(0): every consumer role gives up at its first look -- the step's
  losses are NaN, the sticky word is set (RainbowLearner.check_status raises) and the next
  step, with the normal limit, is bit-identical to the four-launch form again."""
  A, B = 6, 32
  lib = _lib.load()
  online, target, batch, w, noises = _problem(A, B, 6)
  ln = _learner(A, B, online, target, noises)
  ln.use_graphs = False
  dev = _dev_batch(batch, w)
  p0, m0, v0 = ln.online.clone(), ln.adam_m.clone(), ln.adam_v.clone()
  old = lib.dz_act_debug_spin_limit(0)
  try:
    ln.step(*dev, resample_noise=False)
    torch.cuda.synchronize()
  finally:
    lib.dz_act_debug_spin_limit(old)
  assert np.isnan(ln.losses.cpu().numpy()).any()
  assert ln.scalars()['chain_failed']
  with pytest.raises(RuntimeError, match='DZ_SC_CHAIN_FAIL'):
    ln.check_status()
  ln.check_status()       # cleared
  # restore the state the failed step may have touched, then a normal step
  ln.online.copy_(p0); ln.adam_m.copy_(m0); ln.adam_v.copy_(v0); ln.adam_count.zero_()
  ln.step(*dev, resample_noise=False)
  torch.cuda.synchronize()
  ref = _run(A, B, 6, separate=True, steps=1)[0]
  np.testing.assert_array_equal(ln.losses.cpu().numpy(), ref['losses'])
  np.testing.assert_array_equal(ln.online.cpu().numpy(), ref['online'])
