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
  losses are NaN, the sticky word is set (RainbowLearner.check_status raises) and the step is VOID:
  its finalize / optimiser launches read the word and change neither parameters, moments nor the
  optax count -- also for a second step enqueued behind it; after the flag is cleared
  the next step is bit-identical to the four-launch form, which check_status() has switched to."""
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
  ln.step(*dev, resample_noise=False)   # enqueued behind the failure, before anyone looked: void too
  torch.cuda.synchronize()
  assert torch.equal(ln.online, p0) and torch.equal(ln.adam_m, m0) and torch.equal(ln.adam_v, v0)
  assert int(ln.adam_count.item()) == 0
  from dqn_zoo_amd import replay as replay_lib
  with pytest.raises(replay_lib.ChainTimeoutError, match='DZ_SC_CHAIN_FAIL'):
    ln.check_status()
  assert ln.separate_launches          # the fallback
  ln.check_status()       # cleared
  ln.step(*dev, resample_noise=False)
  torch.cuda.synchronize()
  ref = _run(A, B, 6, separate=True, steps=1)[0]
  np.testing.assert_array_equal(ln.losses.cpu().numpy(), ref['losses'])
  np.testing.assert_array_equal(ln.online.cpu().numpy(), ref['online'])


def test_forced_seam_timeout_leaves_the_sum_tree_and_the_agent_running():
  """The same forced give-up with a priority sink: the write-back block reads the sticky word and
  writes NOTHING (tree and running max unchanged), the replay's pinned status word raises
  `ChainTimeoutError` at the next poll, and `Rainbow._recover_from_chain_timeout` clears both words
  and switches the learner to the four-launch form instead of aborting the run."""
  from dqn_zoo_amd import learner as ll, networks, parts
  from dqn_zoo_amd import replay as rl
  A, B, cap = 6, 32, 400
  sup = np.linspace(-10, 10, 51).astype(np.float32)
  T = rl.Transition
  rep = rl.PrioritizedTransitionReplay(
      cap, T(None, None, None, None, None), 0.5,
      parts.LinearSchedule(begin_t=0, end_t=1000, begin_value=0.4, end_value=1.0),
      1e-3, True, np.random.RandomState(11))
  rs = np.random.RandomState(5)
  for i in range(cap):
    rep.add(T(rs.randint(0, 256, (84, 84, 4)).astype(np.uint8), int(rs.randint(A)),
              float(rs.randint(-1, 2)), 0.97, rs.randint(0, 256, (84, 84, 4)).astype(np.uint8)),
            1.0 + (i % 7))
  ln = ll.RainbowLearner(networks.RainbowNetwork(A, sup, 0.1), ll.AdamConfig(), B, seed=3)
  ln.use_graphs = False
  lib = _lib.load()
  s = rep.sample_device(B)
  t = s.transitions
  torch.cuda.synchronize()
  tree0 = rep.tree_storage.clone()
  max0 = float(rep.max_seen_priority_device.item())
  p0 = ln.online.clone()
  old = lib.dz_act_debug_spin_limit(0)
  try:
    ln.step(t.s_tm1, t.a_tm1, t.r_t, t.discount_t, t.s_t, s.weights32, priority_sink=rep.priority_sink(s.ids))
    torch.cuda.synchronize()
  finally:
    lib.dz_act_debug_spin_limit(old)
  assert torch.equal(rep.tree_storage, tree0) and float(rep.max_seen_priority_device.item()) == max0
  assert torch.equal(ln.online, p0) and int(ln.adam_count.item()) == 0
  with pytest.raises(rl.ChainTimeoutError):
    rep.poll_status()
  # what the agent does with it
  from dqn_zoo_amd.rainbow import agent as agent_lib
  shell = agent_lib.Rainbow.__new__(agent_lib.Rainbow)
  shell._learner = ln
  with pytest.warns(RuntimeWarning, match='separate launches'):
    shell._recover_from_chain_timeout(rl.ChainTimeoutError('forced'))
  assert ln.separate_launches and shell.chain_timeouts == 1 and not ln.scalars()['chain_failed']
  # the run goes on: a normal step writes priorities again
  s = rep.sample_device(B)
  t = s.transitions
  ln.step(t.s_tm1, t.a_tm1, t.r_t, t.discount_t, t.s_t, s.weights32, priority_sink=rep.priority_sink(s.ids))
  torch.cuda.synchronize()
  rep.check_status(); ln.check_status()
  assert np.isfinite(ln.losses.cpu().numpy()).all() and not torch.equal(rep.tree_storage, tree0)
  assert int(ln.adam_count.item()) == 1
