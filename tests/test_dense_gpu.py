"""GPU parity of the dense-head learner steps (dz_dense_learn) against the NumPy
oracle: DQN, double-Q, prioritized (double-Q + importance weights), C51 and
QR-DQN -- head outputs, td errors / per-sample losses (<= 1e-5 relative),
every gradient tensor against the float64 truth, and the optimiser step
(centred RMSProp or clip + Adam)."""

import numpy as np
import pytest
import torch

from oracle import qnet_oracle as qo

pytestmark = pytest.mark.gpu

A, B = 6, 32
K = 51
SUPPORT = np.linspace(-10.0, 10.0, K).astype(np.float32)
NQ = 201
QUANTILES = ((np.arange(NQ) + 0.5) / NQ).astype(np.float32)


def _batch(rs, scale_r=1.0):
  s_tm1 = rs.randint(0, 256, (B, 84, 84, 4)).astype(np.uint8)
  s_t = rs.randint(0, 256, (B, 84, 84, 4)).astype(np.uint8)
  a = rs.randint(A, size=B).astype(np.int64)
  r = rs.choice([-1.0, 0.0, 1.0], size=B) * scale_r
  d = rs.choice([0.0, 0.99], size=B)
  return s_tm1, a, r, d, s_t


def _dev(batch):
  return [torch.from_numpy(x).cuda() for x in batch]


def _f64(t):
  return {k: v.astype(np.float64) for k, v in t.items()}


def _check_grads(g_dev, g32, g64):
  assert set(g_dev) == set(g64)
  for k in sorted(g64):
    scale = max(np.abs(g64[k]).max(), 1e-30)
    e_dev = np.abs(g_dev[k] - g64[k]).max() / scale
    e_orc = np.abs(g32[k] - g64[k]).max() / scale
    assert e_dev < max(1e-4, 4 * e_orc), (k, e_dev, e_orc)


AMBIG = 2e-5   # |float64 pre-activation| below which a ReLU side may legitimately differ


def _relu_ties(ln, online, s_tm1):
  """ReLU units of the differentiated apply (group 0) whose side differs between the device's
  stored activations and the oracle's float64 pre-activations: (count, max |z64| among them).
  A float32 sum in another order may land a pre-activation of ~1e-7 on the other side of zero;
  that switches a whole gradient path (tests/test_trajectory_gpu.py has the full argument)."""
  _, c = qo.mlp_head_fwd(_f64(online), s_tm1, np.float64)
  tc = c['torso']
  pre = dict(act1=tc['conv1'][2], act2=tc['conv2'][2], feat=tc['conv3'][2], h1=c['z1'])
  n, worst = 0, 0.0
  for k, z in pre.items():
    dev = ln.ws_view(k, z.size).cpu().numpy().reshape(z.shape)
    bad = (dev > 0) != (z > 0)
    if bad.any():
      n += int(bad.sum())
      worst = max(worst, float(np.abs(z[bad]).max()))
  return n, worst


def _tie_free_batch(rs, ln, online, run, **kw):
  """Draws batches until the device's ReLU sides equal the float64 oracle's (a mismatch is
  accepted ONLY at |z64| <= AMBIG, and at most twice); returns the batch `run` last saw."""
  for _ in range(3):
    batch = _batch(rs, **kw)
    run(batch)
    torch.cuda.synchronize()
    n, worst = _relu_ties(ln, online, batch[0])
    if not n:
      return batch
    assert worst <= AMBIG, ('ReLU side differs at |z64| =', worst)
  raise AssertionError('three batches in a row with a ReLU tie')


def _make(kind_net, loss, opt, seed, **kw):
  from dqn_zoo_amd import learner as ll, networks
  rs = np.random.RandomState(seed)
  online = qo.init_params(kind_net, A, rs, num_atoms=K, num_quantiles=NQ)
  target = qo.init_params(kind_net, A, rs, num_atoms=K, num_quantiles=NQ)
  net = networks.DenseNetwork(kind_net, A, support=SUPPORT, quantiles=QUANTILES)
  ln = ll.DenseLearner(net, loss, opt, B, params=online, **kw)
  ln.set_params(target, 'target')
  return rs, online, target, ln


@pytest.mark.parametrize('kind', ['dqn', 'double_q', 'prioritized'])
def test_dqn_family_step(kind):
  from dqn_zoo_amd import learner as ll, _lib
  net = 'dqn' if kind == 'dqn' else 'double_dqn'
  loss = 'q' if kind == 'dqn' else 'double_q'
  opt = ll.RmsPropConfig(learning_rate=0.00025, decay=0.95, eps=0.01 / 32 ** 2)
  rs, online, target, ln = _make(net, loss, opt, 3 + len(kind),
                                 grad_error_bound=1.0 / 32)
  w = rs.uniform(0.2, 1.0, size=B).astype(np.float32) if kind == 'prioritized' \
      else None
  wd = None if w is None else torch.from_numpy(w).cuda()
  batch = _tie_free_batch(   # scale_r: some |td| > 1 so the gradient clip binds
      rs, ln, online,
      lambda b: ln.step(*_dev(b), wd, phases=_lib.PHASE_FORWARD | _lib.PHASE_BACKWARD), scale_r=2.5)
  l32, td, g32, aux = qo.dqn_family_loss_and_grads(kind, online, target, batch, w,
                                                   1.0 / 32)
  _, _, g64, _ = qo.dqn_family_loss_and_grads(kind, _f64(online), _f64(target),
                                              batch, w, 1.0 / 32, np.float64)
  L = ln.layout
  out = ln.ws_view('out', ln.groups * B * L.c.fc2_ld).cpu().numpy().reshape(
      ln.groups, B, L.c.fc2_ld)[:, :, :A]
  np.testing.assert_allclose(out[0], aux['q_tm1'], rtol=2e-5, atol=2e-6)
  np.testing.assert_allclose(out[1], aux['q_target'], rtol=2e-5, atol=2e-6)
  if kind != 'dqn':
    np.testing.assert_allclose(out[2], aux['q_sel'], rtol=2e-5, atol=2e-6)
  np.testing.assert_allclose(ln.losses.cpu().numpy(), td, rtol=1e-5, atol=2e-6)
  np.testing.assert_allclose(ln.priorities.cpu().numpy(), np.abs(td), rtol=1e-5,
                             atol=2e-6)
  assert (np.abs(td) > 1.0).any() and (np.abs(td) < 1.0).any()
  g_dev = L.unpack(ln.grad.cpu().numpy())
  _check_grads(g_dev, g32, g64)
  # centred RMSProp fed with the device gradients (the stored-gradient form of the step:
  # this loop reads the whole gradient vector back)
  ln.keep_all_grads = True
  p, st = dict(online), qo.rmsprop_init(online)
  for it in range(2):
    ln.step(*_dev(batch), wd)
    torch.cuda.synchronize()
    g_dev = L.unpack(ln.grad.cpu().numpy())
    p, st = qo.rmsprop_centered_update(p, g_dev, st, opt.learning_rate, opt.decay,
                                       opt.eps)
    p_dev = ln.get_params()
    mu_dev = L.unpack(ln.opt_m.cpu().numpy())
    for k in p:
      np.testing.assert_allclose(mu_dev[k], st['mu'][k], rtol=1e-5, atol=1e-12)
      assert np.abs(p_dev[k] - p[k]).max() <= 2e-3 * opt.learning_rate * 40, k
    p, st = p_dev, dict(mu=mu_dev, nu=L.unpack(ln.opt_v.cpu().numpy()))

@pytest.mark.parametrize('kind', ['dqn', 'prioritized'])
def test_rmsprop_inside_finalize_is_bit_identical_to_the_split_phases(kind):
  """A full step applies RMSProp where finalize produces the small gradients and in
  extra blocks of that launch for the GEMM-written ranges; forward+backward followed
  by the optimiser phase alone goes through finalize_grads_kernel + rmsprop_kernel.
  Same arithmetic, same operands: parameters and both moments must agree bitwise."""
  from dqn_zoo_amd import learner as ll, _lib
  net = 'dqn' if kind == 'dqn' else 'double_dqn'
  loss = 'q' if kind == 'dqn' else 'double_q'
  opt = ll.RmsPropConfig(learning_rate=0.00025, decay=0.95, eps=0.01 / 32 ** 2)
  rs, _, _, fused = _make(net, loss, opt, 11, grad_error_bound=1.0 / 32)
  _, _, _, split = _make(net, loss, opt, 11, grad_error_bound=1.0 / 32)
  fused.keep_all_grads = True   # (the default forms fc1's gradient in the optimiser: next test)
  w = rs.uniform(0.2, 1.0, size=B).astype(np.float32) if kind == 'prioritized' else None
  wd = None if w is None else torch.from_numpy(w).cuda()
  for it in range(3):
    batch = _dev(_batch(rs, scale_r=2.5))
    fused.step(*batch, wd)
    split.step(*batch, wd, phases=_lib.PHASE_FORWARD | _lib.PHASE_BACKWARD)
    split.step(*batch, wd, phases=_lib.PHASE_OPTIMIZER)
    torch.cuda.synchronize()
    for name in ('online', 'opt_m', 'opt_v', 'grad'):
      a, b = getattr(fused, name).cpu().numpy(), getattr(split, name).cpu().numpy()
      assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), (it, name)

@pytest.mark.parametrize('kind', ['dqn', 'prioritized'])
def test_on_the_fly_fc1_gradient_in_rmsprop(kind):
  """The DEFAULT full step of the RMSProp learners never stores fc1's weight gradient: the
  optimiser launch forms each entry from the layer's input and dh1 (RmsOnFly).  Read back
  directly: from a zero state the first step leaves mu = (1 - decay) g and nu = (1 - decay)
  g^2 element by element -- against the float64 oracle -- and three steps stay within
  float32 rounding of the stored-gradient form (same sums, another order)."""
  from dqn_zoo_amd import learner as ll
  net = 'dqn' if kind == 'dqn' else 'double_dqn'
  loss = 'q' if kind == 'dqn' else 'double_q'
  opt = ll.RmsPropConfig(learning_rate=0.00025, decay=0.95, eps=0.01 / 32 ** 2)
  rs, online, target, fly = _make(net, loss, opt, 17, grad_error_bound=1.0 / 32)
  _, _, _, stored = _make(net, loss, opt, 17, grad_error_bound=1.0 / 32)
  stored.keep_all_grads = True
  assert not fly.keep_all_grads
  w = rs.uniform(0.2, 1.0, size=B).astype(np.float32) if kind == 'prioritized' else None
  wd = None if w is None else torch.from_numpy(w).cuda()
  batch = _batch(rs, scale_r=2.5)
  fly.step(*_dev(batch), wd)
  stored.step(*_dev(batch), wd)
  torch.cuda.synchronize()
  _, _, g64, _ = qo.dqn_family_loss_and_grads(kind, _f64(online), _f64(target), batch, w,
                                              1.0 / 32, np.float64)
  L = fly.layout
  mu = L.unpack(fly.opt_m.cpu().numpy())
  nu = L.unpack(fly.opt_v.cpu().numpy())
  g = mu['fc1/w'].astype(np.float64) / (1.0 - opt.decay)
  scale = np.abs(g64['fc1/w']).max()
  assert np.abs(g - g64['fc1/w']).max() / scale < 2e-6
  np.testing.assert_allclose(nu['fc1/w'].astype(np.float64) / (1.0 - opt.decay),
                             g64['fc1/w'] ** 2, rtol=1e-5, atol=5e-6 * scale ** 2)
  # the stored-gradient buffer of the default form holds no fc1 block (never written)
  assert float(fly.grad[int(L.c.fc1_w):int(L.c.fc1_w) + 3136 * int(L.c.fc1_ld)].abs().max()) == 0.0
  for it in range(2):
    batch = _batch(rs, scale_r=2.5)
    fly.step(*_dev(batch), wd)
    stored.step(*_dev(batch), wd)
  torch.cuda.synchronize()
  for name in ('online', 'opt_m', 'opt_v'):
    a, b = getattr(fly, name).cpu().numpy(), getattr(stored, name).cpu().numpy()
    if name == 'online':   # an update is lr * g / sqrt(nu - mu^2 + eps): |.| <= lr / sqrt(eps) = 32 lr
      np.testing.assert_allclose(a, b, rtol=0, atol=0.02 * opt.learning_rate)
    else:
      np.testing.assert_allclose(a, b, rtol=2e-5, atol=1e-9)


def test_rmsprop_inside_finalize_two_flat_ranges():
  """33 actions: the second layer's weight gradient is written by the GEMM launch, so
  the optimiser blocks of the finalize launch walk TWO flat ranges (fc1 and fc2 weights).
  Must still be bit-identical to finalize + rmsprop_kernel."""
  from dqn_zoo_amd import learner as ll, networks, _lib
  actions, batch = 33, 16
  opt = ll.RmsPropConfig(learning_rate=0.00025, decay=0.95, eps=0.01 / 32 ** 2)
  lns = []
  for _ in range(2):
    rs = np.random.RandomState(5)
    online = qo.init_params('double_dqn', actions, rs)
    target = qo.init_params('double_dqn', actions, rs)
    ln = ll.DenseLearner(networks.DenseNetwork('double_dqn', actions), 'double_q', opt, batch,
                         params=online, grad_error_bound=1.0 / 32)
    ln.set_params(target, 'target')
    lns.append(ln)
  fused, split = lns
  fused.keep_all_grads = True   # (bit-identity is a property of the stored-gradient form)
  for it in range(2):
    b = [torch.from_numpy(x).cuda() for x in (
        rs.randint(0, 256, (batch, 84, 84, 4)).astype(np.uint8),
        rs.randint(actions, size=batch).astype(np.int64),
        rs.choice([-1.0, 0.0, 1.0], size=batch) * 2.5, rs.choice([0.0, 0.99], size=batch),
        rs.randint(0, 256, (batch, 84, 84, 4)).astype(np.uint8))]
    fused.step(*b, None)
    split.step(*b, None, phases=_lib.PHASE_FORWARD | _lib.PHASE_BACKWARD)
    split.step(*b, None, phases=_lib.PHASE_OPTIMIZER)
    torch.cuda.synchronize()
    for name in ('online', 'opt_m', 'opt_v', 'grad'):
      x, y = getattr(fused, name).cpu().numpy(), getattr(split, name).cpu().numpy()
      assert np.array_equal(x.view(np.uint32), y.view(np.uint32)), (it, name)

@pytest.mark.parametrize('kind_net,loss,actions', [('dqn', 'q', 6), ('c51', 'categorical', 6)])
def test_fused_global_norm_matches_the_norm_of_the_stored_gradient(kind_net, loss, actions):
  """With Adam the global norm of a full step is assembled from partials the gradient
  writers leave (weight-gradient kernels, finalize incl. the narrow head's weight-gradient
  job); forward+backward followed by the optimiser phase alone takes it from the stored
  gradient (sumsq_kernel).  With a clip that binds hard, a missing block of the norm would
  scale the whole update: the two paths must agree to float rounding."""
  from dqn_zoo_amd import learner as ll, networks, _lib
  opt = ll.AdamConfig(learning_rate=1e-3, max_global_grad_norm=1e-3)
  lns = []
  for _ in range(2):
    rs = np.random.RandomState(9)
    online = qo.init_params(kind_net, actions, rs, num_atoms=K, num_quantiles=NQ)
    target = qo.init_params(kind_net, actions, rs, num_atoms=K, num_quantiles=NQ)
    net = networks.DenseNetwork(kind_net, actions, support=SUPPORT, quantiles=QUANTILES)
    kw = dict(grad_error_bound=1.0 / 32) if loss == 'q' else {}
    ln = ll.DenseLearner(net, loss, opt, B, params=online, **kw)
    ln.set_params(target, 'target')
    lns.append(ln)
  fused, split = lns
  batch = _dev(_batch(rs, scale_r=2.5))
  wd = torch.ones(B, dtype=torch.float32, device='cuda') if loss == 'categorical' else None
  p0 = fused.online.clone()
  fused.step(*batch, wd)
  split.step(*batch, wd, phases=_lib.PHASE_FORWARD | _lib.PHASE_BACKWARD)
  split.step(*batch, wd, phases=_lib.PHASE_OPTIMIZER)
  torch.cuda.synchronize()
  du_f = (fused.online - p0).cpu().numpy().astype(np.float64)
  du_s = (split.online - p0).cpu().numpy().astype(np.float64)
  assert np.abs(du_s).max() > 0
  # the update is proportional to the clip scale: compare the two updates as vectors
  rel = np.linalg.norm(du_f - du_s) / np.linalg.norm(du_s)
  assert rel < 1e-5, rel


@pytest.mark.parametrize('actions,batch,kind', [(3, 10, 'dqn'), (18, 32, 'prioritized'),
                                                (32, 48, 'double_q'), (4, 7, 'prioritized'),
                                                (33, 16, 'double_q')])  # 33 actions: the unfused fallback
def test_fused_q_head_other_shapes(actions, batch, kind):
  """The one-launch Q head (fc1 epilogue + second layer + TD loss + dh1, the
  second layer's weight gradient in finalize) at other action counts and batch
  sizes than BASELINE's, against the oracle: outputs, TD errors, every gradient."""
  from dqn_zoo_amd import learner as ll, networks, _lib
  rs = np.random.RandomState(100 + actions + batch)
  net_kind = 'dqn' if kind == 'dqn' else 'double_dqn'
  loss = 'q' if kind == 'dqn' else 'double_q'
  online = qo.init_params(net_kind, actions, rs)
  target = qo.init_params(net_kind, actions, rs)
  net = networks.DenseNetwork(net_kind, actions)
  opt = ll.RmsPropConfig(learning_rate=0.00025, decay=0.95, eps=0.01 / 32 ** 2)
  ln = ll.DenseLearner(net, loss, opt, batch, params=online, grad_error_bound=1.0 / 32)
  ln.set_params(target, 'target')
  s_tm1 = rs.randint(0, 256, (batch, 84, 84, 4)).astype(np.uint8)
  s_t = rs.randint(0, 256, (batch, 84, 84, 4)).astype(np.uint8)
  a = rs.randint(actions, size=batch).astype(np.int64)
  r = rs.choice([-1.0, 0.0, 1.0], size=batch) * 2.5
  d = rs.choice([0.0, 0.99], size=batch)
  b = (s_tm1, a, r, d, s_t)
  w = rs.uniform(0.2, 1.0, size=batch).astype(np.float32) if kind == 'prioritized' else None
  wd = None if w is None else torch.from_numpy(w).cuda()
  ln.step(*_dev(b), wd, phases=_lib.PHASE_FORWARD | _lib.PHASE_BACKWARD)
  torch.cuda.synchronize()
  _, td, g32, aux = qo.dqn_family_loss_and_grads(kind, online, target, b, w, 1.0 / 32)
  _, _, g64, _ = qo.dqn_family_loss_and_grads(kind, _f64(online), _f64(target), b, w,
                                              1.0 / 32, np.float64)
  L = ln.layout
  out = ln.ws_view('out', ln.groups * batch * L.c.fc2_ld).cpu().numpy().reshape(
      ln.groups, batch, L.c.fc2_ld)[:, :, :actions]
  np.testing.assert_allclose(out[0], aux['q_tm1'], rtol=2e-5, atol=2e-6)
  np.testing.assert_allclose(out[1], aux['q_target'], rtol=2e-5, atol=2e-6)
  np.testing.assert_allclose(ln.losses.cpu().numpy(), td, rtol=1e-5, atol=2e-6)
  _check_grads(L.unpack(ln.grad.cpu().numpy()), g32, g64)


@pytest.mark.parametrize('kind', ['c51', 'qr'])
def test_distributional_dense_step(kind):
  from dqn_zoo_amd import learner as ll, _lib
  opt = ll.AdamConfig(learning_rate=0.00025, eps=0.01 / 32,
                      max_global_grad_norm=10.0)
  loss = 'categorical' if kind == 'c51' else 'quantile'
  rs, online, target, ln = _make(kind, loss, opt, 11 if kind == 'c51' else 12,
                                 huber_param=1.0)
  batch = _tie_free_batch(
      rs, ln, online, lambda b: ln.step(*_dev(b), phases=_lib.PHASE_FORWARD | _lib.PHASE_BACKWARD))
  if kind == 'c51':
    f = lambda o, t, dt: qo.c51_loss_and_grads(o, t, batch, SUPPORT.astype(dt), A, dt)
  else:
    f = lambda o, t, dt: qo.qr_loss_and_grads(o, t, batch, QUANTILES.astype(dt), A,
                                              1.0, dt)
  l32, losses, g32, aux = f(online, target, np.float32)
  _, _, g64, _ = f(_f64(online), _f64(target), np.float64)
  L = ln.layout
  n_out = ln.network.num_outputs
  out = ln.ws_view('out', 2 * B * L.c.fc2_ld).cpu().numpy().reshape(
      2, B, L.c.fc2_ld)[:, :, :n_out]
  np.testing.assert_allclose(out[0], aux['out_tm1'], rtol=3e-5, atol=3e-6)
  np.testing.assert_allclose(out[1], aux['out_target'], rtol=3e-5, atol=3e-6)
  np.testing.assert_allclose(ln.losses.cpu().numpy(), losses, rtol=1e-5)
  _check_grads(L.unpack(ln.grad.cpu().numpy()), g32, g64)
  # clip + Adam fed with the device gradients
  ln.step(*_dev(batch))
  torch.cuda.synchronize()
  g_dev = L.unpack(ln.grad.cpu().numpy())
  clipped, gn = qo.clip_by_global_norm(g_dev, 10.0)
  p, st = qo.adam_update(online, clipped, qo.adam_init(online), opt.learning_rate,
                         opt.eps)
  p_dev = ln.get_params()
  for k in p:
    assert np.abs(p_dev[k] - p[k]).max() <= 2e-3 * opt.learning_rate + 1e-9, k
  assert int(ln.opt_count.item()) == 1


@pytest.mark.parametrize('kind,actions', [('c51', 11), ('c51', 18), ('qr', 18), ('qr', 3)])
def test_distributional_dense_step_other_head_widths(kind, actions, monkeypatch):
  """The second layer's input gradient is a row-owning stream over 256-output chunks
  (csrc/dz_row_dgrad.h): 561 / 918 outputs = 3 / 4 chunks (compile-time job counts), 3618
  = 15 chunks (run-time job count, two rows per workgroup), 603 = 3."""
  import sys
  monkeypatch.setattr(sys.modules[__name__], 'A', actions)
  test_distributional_dense_step(kind)


@pytest.mark.parametrize('kind_net,actions', [('dqn', 6), ('double_dqn', 6), ('dqn', 18), ('double_dqn', 3),
                                               ('c51', 6), ('c51', 18), ('qr', 6), ('qr', 18)])
def test_one_launch_q_decision(kind_net, actions, monkeypatch):
  """`head_async` is ONE launch (dz_dense_act, csrc/dz_act_one.h) whose head outputs -- A
  q-values, or the 51 A / 201 A distribution outputs of C51 / QR-DQN (up to 114 tail
  workgroups) -- land in a pinned slot as 8-byte {value, marker} words the host polls.  Nine decisions
  (more than the ring of slots, both sets of intermediates many times) against the oracle and
  the multi-launch apply; an all-zero observation; the shared fc2 bias of double_dqn."""
  import sys
  from dqn_zoo_amd import learner as ll
  monkeypatch.setattr(sys.modules[__name__], 'A', actions)
  loss = {'dqn': 'q', 'double_dqn': 'double_q', 'c51': 'categorical', 'qr': 'quantile'}[kind_net]
  opt = ll.RmsPropConfig() if loss in ('q', 'double_q') else ll.AdamConfig()
  rs, online, _, ln = _make(kind_net, loss, opt, 41 + actions)
  assert ln.act_one_launch
  for i in range(9):
    x = rs.randint(0, 256, (1, 84, 84, 4)).astype(np.uint8)
    if i == 4:
      x[:] = 0
    xd = torch.from_numpy(x).cuda()
    q = ln.head_async(xd)()
    ref, _ = qo.mlp_head_fwd(online, x)
    np.testing.assert_allclose(q, ref[0], rtol=2e-5, atol=2e-6)
    out = ln.apply(xd)[0]
    np.testing.assert_allclose(q, out.cpu().numpy()[0], rtol=0, atol=2e-6)
  seams = int(ln.network.layout(1, 1).c.ws_act_seams)
  words = ln._act_ws[seams:seams + 64 * 8:64].view(torch.int32).tolist()  # pylint: disable=protected-access
  assert words[3] == 9 and words[5] == 0, words


def test_dense_act_into_device_memory():
  """dz_dense_act's {value, marker} words may also go to device memory (include/dqnzoo_hip.h);
  called through the C ABI directly, twice on one workspace (both sets of intermediates)."""
  from dqn_zoo_amd import _lib, learner as ll
  rs, online, _, ln = _make('dqn', 'q', ll.RmsPropConfig(), 77)
  lib = _lib.load()
  ws = torch.zeros(ln.network.layout(1, 1).ws_count, dtype=torch.float32, device='cuda')
  for _ in range(2):
    x = rs.randint(0, 256, (1, 84, 84, 4)).astype(np.uint8)
    xd = torch.from_numpy(x).cuda()
    pairs = torch.zeros((A, 2), dtype=torch.float32, device='cuda')
    _lib.check(lib.dz_dense_act(A, 0, ln.online.data_ptr(), xd.data_ptr(), ws.data_ptr(),
                                pairs.data_ptr(), _lib.stream_ptr(ln.device)), 'dz_dense_act')
    torch.cuda.synchronize()
    got = pairs.cpu().numpy()
    ref, _ = qo.mlp_head_fwd(online, x)
    np.testing.assert_array_equal(got[:, 1], np.ones(A, np.float32))
    np.testing.assert_allclose(got[:, 0], ref[0], rtol=2e-5, atol=2e-6)
  # a misaligned slot is refused before anything is enqueued
  bad = torch.zeros(2 * A + 1, dtype=torch.float32, device='cuda')[1:]
  assert lib.dz_dense_act(A, 0, ln.online.data_ptr(), xd.data_ptr(), ws.data_ptr(),
                          bad.data_ptr(), _lib.stream_ptr(ln.device)) != 0


def test_dense_apply_and_shared_bias():
  from dqn_zoo_amd import learner as ll
  rs, online, target, ln = _make('double_dqn', 'double_q', ll.RmsPropConfig(), 30)
  assert online['fc2/b'].shape == (1,)
  x = rs.randint(0, 256, (3, 84, 84, 4)).astype(np.uint8)
  out, q, greedy, vmax = ln.apply(torch.from_numpy(x).cuda())
  ref, _ = qo.mlp_head_fwd(online, x)
  np.testing.assert_allclose(out.cpu().numpy(), ref, rtol=2e-5, atol=2e-6)
  np.testing.assert_array_equal(q.cpu().numpy(), out.cpu().numpy())
  np.testing.assert_array_equal(greedy.cpu().numpy(), ref.argmax(axis=1))
  np.testing.assert_allclose(vmax.cpu().numpy(), ref.max(axis=1), rtol=2e-5)
  ln.sync_target()
  for k, v in ln.get_params('target').items():
    np.testing.assert_array_equal(v, online[k])
