"""Self-validation of the Q-loss/update oracle (oracle/qnet_oracle.py).

The arithmetic of this half lives in third-party packages that are absent
(rlax/optax/haiku/jax): PARITY UNPINNED against the reference itself.  What CAN
be checked is checked here: float64 central finite differences, an independent
torch-autograd float64 model written with different primitives (NCHW conv2d,
vectorised projection), projection invariants, optimiser closed forms and the
pins of the reference's networks_test.py.
"""

import numpy as np
import pytest
import torch

from oracle import qnet_oracle as qo

A = 3
K = 51
SUPPORT = np.linspace(-10.0, 10.0, K)


def _batch(rs, b):
  s_tm1 = rs.randint(0, 256, (b, 84, 84, 4)).astype(np.uint8)
  s_t = rs.randint(0, 256, (b, 84, 84, 4)).astype(np.uint8)
  a = rs.randint(A, size=b)
  r = rs.choice([-1.0, 0.0, 1.0], size=b)
  d = rs.choice([0.0, 0.99 ** 3], size=b)
  return s_tm1, a, r, d, s_t


def _setup(dt, b=3, seed=0):
  rs = np.random.RandomState(seed)
  online = qo.init_params('rainbow', A, rs, dt)
  target = qo.init_params('rainbow', A, rs, dt)
  # make sigma matter and push activations off zero
  for k in online:
    if 'sigma' in k:
      online[k] = (online[k] * 5).astype(dt)
  batch = _batch(rs, b)
  w = rs.uniform(0.2, 1.0, size=b)
  noises = [qo.sample_noise(rs, A, dt) for _ in range(3)]
  return online, target, batch, w, noises


def test_param_counts_and_pins():
  rs = np.random.RandomState(0)
  p = qo.init_params('rainbow', 6, rs)
  assert sum(v.size for v in p.values()) == 6868485
  p = qo.init_params('dqn', 6, rs)
  assert sum(v.size for v in p.values()) == 1687206
  assert p['conv1/w'].shape == (8, 8, 4, 32)          # HWIO, networks_test.py:53
  assert p['fc1/w'].shape == (3136, 512)              # (in,out), networks_test.py:44
  d = qo.init_params('double_dqn', 6, rs)
  assert d['fc2/b'].shape == (1,)                     # networks_test.py:57-103
  r = qo.init_params('rainbow', 6, rs)
  np.testing.assert_allclose(r['adv1/sigma/w'], 0.1 / np.sqrt(3136), rtol=1e-6)
  np.testing.assert_allclose(r['val2/sigma/b'], 0.1 / np.sqrt(512), rtol=1e-6)
  assert 'adv2/mu/b' not in r and 'adv1/mu/b' in r    # networks.py:240-250
  assert np.abs(r['conv1/w']).max() <= 1 / np.sqrt(256)
  n = qo.sample_noise(rs, 6)
  assert np.abs(n['adv1/in']).max() <= np.sqrt(2.0) + 1e-6
  assert [k for k, _ in qo.noise_shapes(6)][:2] == ['adv1/in', 'adv1/out']


def test_shared_bias_broadcast():
  rs = np.random.RandomState(1)
  p = qo.init_params('double_dqn', 4, rs, np.float64)
  x = rs.randint(0, 256, (2, 84, 84, 4)).astype(np.uint8)
  out, cache = qo.mlp_head_fwd(p, x, np.float64)
  p2 = dict(p)
  p2['fc2/b'] = p['fc2/b'] + 1.0
  out2, _ = qo.mlp_head_fwd(p2, x, np.float64)
  np.testing.assert_allclose(out2 - out, 1.0)
  g = qo.mlp_head_bwd(p, cache, np.ones_like(out))
  assert g['fc2/b'].shape == (1,) and g['fc2/b'][0] == out.size


def test_projection_invariants():
  rs = np.random.RandomState(2)
  z = SUPPORT
  for _ in range(20):
    p = rs.dirichlet(np.ones(K))
    r, g = rs.uniform(-3, 3), rs.uniform(0, 1)
    m = qo.categorical_l2_project(r + g * z, p, z)
    assert abs(m.sum() - 1) < 1e-12 and (m >= 0).all()
    # mean is preserved when nothing is clipped
    if (r + g * z).min() >= z[0] and (r + g * z).max() <= z[-1]:
      assert abs((m * z).sum() - (p * (r + g * z)).sum()) < 1e-10
  p = rs.dirichlet(np.ones(K))
  np.testing.assert_allclose(qo.categorical_l2_project(z, p, z), p, atol=1e-14)
  # everything clipped to vmax -> all mass on the last atom
  m = qo.categorical_l2_project(100 + z, p, z)
  assert abs(m[-1] - 1) < 1e-12 and abs(m[:-1]).max() < 1e-12
  # hand-computed: one atom at 0.1 between z=0.0 and z=0.4 (delta 0.4)
  m = qo.categorical_l2_project(np.full(K, 0.1), p, z)
  i0 = K // 2
  assert abs(m[i0] - 0.75) < 1e-12 and abs(m[i0 + 1] - 0.25) < 1e-12


def _torch_rainbow(params, x_u8, noise, support):
  """Independent float64 forward: NCHW conv2d, einsum dueling."""
  x = torch.from_numpy(x_u8.astype(np.float64) / 255.0).permute(0, 3, 1, 2)
  for name, stride in (('conv1', 4), ('conv2', 2), ('conv3', 1)):
    w = params[name + '/w'].permute(3, 2, 0, 1)  # HWIO -> OIHW
    x = torch.relu(torch.nn.functional.conv2d(x, w, params[name + '/b'],
                                              stride=stride))
  feat = x.permute(0, 2, 3, 1).reshape(x.shape[0], -1)

  def noisy(name, h):
    ein = torch.from_numpy(noise[name + '/in'])
    eout = torch.from_numpy(noise[name + '/out'])
    w_eff = params[name + '/mu/w'] + params[name + '/sigma/w'] * torch.outer(
        ein, eout)
    y = h @ w_eff + params[name + '/sigma/b'] * eout
    if name + '/mu/b' in params:
      y = y + params[name + '/mu/b']
    return y

  adv = noisy('adv2', torch.relu(noisy('adv1', feat))).reshape(-1, A, K)
  val = noisy('val2', torch.relu(noisy('val1', feat))).reshape(-1, 1, K)
  logits = val + adv - adv.mean(dim=1, keepdim=True)
  q = (torch.softmax(logits, -1) * torch.from_numpy(support)).sum(-1)
  return logits, q


def _torch_project(z_p, probs, z_q):
  """Vectorised triangular-kernel form of the Cramer projection."""
  dz = z_q[1] - z_q[0]
  zc = torch.clamp(z_p, z_q[0], z_q[-1])
  tri = torch.clamp(1 - (zc[None, :] - z_q[:, None]).abs() / dz, 0, 1)
  return (tri * probs[None, :]).sum(-1)


def test_rainbow_grads_vs_torch_autograd_f64():
  dt = np.float64
  online, target, batch, w, noises = _setup(dt, b=4, seed=3)
  loss, losses, grads, aux = qo.rainbow_loss_and_grads(
      online, target, batch, w, noises, SUPPORT, A, dt)
  tp = {k: torch.tensor(v, requires_grad=True) for k, v in online.items()}
  tt = {k: torch.tensor(v) for k, v in target.items()}
  s_tm1, a, r, d, s_t = batch
  logits_tm1, _ = _torch_rainbow(tp, s_tm1, noises[0], SUPPORT)
  with torch.no_grad():
    _, q_t = _torch_rainbow(tp, s_t, noises[1], SUPPORT)
    logits_tgt, _ = _torch_rainbow(tt, s_t, noises[2], SUPPORT)
  zs = torch.from_numpy(SUPPORT)
  tl = []
  for i in range(len(a)):
    astar = int(torch.argmax(q_t[i]))
    m = _torch_project(r[i] + d[i] * zs, torch.softmax(logits_tgt[i, astar], -1),
                       zs)
    tl.append(-(m * torch.log_softmax(logits_tm1[i, a[i]], -1)).sum())
  tl = torch.stack(tl)
  tloss = (tl * torch.from_numpy(w)).mean()
  tloss.backward()
  np.testing.assert_allclose(losses, tl.detach().numpy(), rtol=1e-10)
  np.testing.assert_allclose(loss, tloss.item(), rtol=1e-10)
  np.testing.assert_allclose(aux['q_t'], q_t.numpy(), rtol=1e-10)
  assert set(grads) == set(online)
  for k in online:
    g = tp[k].grad.numpy()
    scale = max(np.abs(g).max(), 1e-12)
    assert np.abs(grads[k] - g).max() / scale < 1e-9, k


def test_rainbow_loss_finite_differences_f64():
  dt = np.float64
  online, target, batch, w, noises = _setup(dt, b=2, seed=4)
  loss, _, grads, _ = qo.rainbow_loss_and_grads(online, target, batch, w,
                                                noises, SUPPORT, A, dt)
  rs = np.random.RandomState(0)
  for k in ['conv1/w', 'conv3/b', 'adv1/sigma/w', 'adv2/mu/w', 'val1/mu/b',
            'val2/sigma/b', 'adv2/sigma/b']:
    g = grads[k].reshape(-1)
    idx = np.argsort(-np.abs(g))[:2].tolist() + [int(rs.randint(g.size))]
    for j in idx:
      h = 1e-5
      vals = []
      for sgn in (+1, -1):
        p2 = dict(online)
        arr = online[k].copy().reshape(-1)
        arr[j] += sgn * h
        p2[k] = arr.reshape(online[k].shape)
        vals.append(qo.rainbow_loss_and_grads(p2, target, batch, w, noises,
                                              SUPPORT, A, dt)[0])
      fd = (vals[0] - vals[1]) / (2 * h)
      assert abs(fd - g[j]) <= 1e-6 * max(1.0, abs(g[j])) + 1e-9, (k, j, fd, g[j])


@pytest.mark.parametrize('kind', ['dqn', 'double_q', 'prioritized'])
def test_dqn_family_vs_torch_autograd_f64(kind):
  dt = np.float64
  rs = np.random.RandomState(5)
  net = 'dqn' if kind == 'dqn' else 'double_dqn'
  online = qo.init_params(net, A, rs, dt)
  target = qo.init_params(net, A, rs, dt)
  batch = _batch(rs, 6)
  batch = (batch[0], batch[1], batch[2] * 3.0, batch[3], batch[4])
  w = rs.uniform(0.2, 1.0, size=6) if kind == 'prioritized' else None
  bound = 1.0 / 32
  loss, td, grads, _ = qo.dqn_family_loss_and_grads(kind, online, target, batch,
                                                    w, bound, dt)
  tp = {k: torch.tensor(v, requires_grad=True) for k, v in online.items()}
  tt = {k: torch.tensor(v) for k, v in target.items()}

  def fwd(p, x_u8):
    x = torch.from_numpy(x_u8.astype(np.float64) / 255.0).permute(0, 3, 1, 2)
    for name, stride in (('conv1', 4), ('conv2', 2), ('conv3', 1)):
      x = torch.relu(torch.nn.functional.conv2d(
          x, p[name + '/w'].permute(3, 2, 0, 1), p[name + '/b'], stride=stride))
    f = x.permute(0, 2, 3, 1).reshape(x.shape[0], -1)
    return torch.relu(f @ p['fc1/w'] + p['fc1/b']) @ p['fc2/w'] + p['fc2/b']

  class ClipGrad(torch.autograd.Function):  # rlax.clip_gradient

    @staticmethod
    def forward(ctx, x):
      return x.clone()

    @staticmethod
    def backward(ctx, g):
      return torch.clamp(g, -bound, bound)

  s_tm1, a, r, d, s_t = batch
  q_tm1 = fwd(tp, s_tm1)
  with torch.no_grad():
    q_tgt = fwd(tt, s_t)
    sel = fwd(tp, s_t) if kind != 'dqn' else q_tgt
  idx = torch.arange(6)
  boot = q_tgt[idx, torch.argmax(sel, 1)]
  ttd = torch.from_numpy(r) + torch.from_numpy(d) * boot - q_tm1[idx, a]
  tl = 0.5 * ClipGrad.apply(ttd) ** 2
  if w is not None:
    tl = tl * torch.from_numpy(w)
  tl.mean().backward()
  np.testing.assert_allclose(td, ttd.detach().numpy(), rtol=1e-10)
  assert (np.abs(td) > bound * 32).any() and (np.abs(td) < bound * 32).any()
  for k in online:
    g = tp[k].grad.numpy()
    assert np.abs(grads[k] - g).max() / max(np.abs(g).max(), 1e-12) < 1e-9, k


def test_quantile_loss_finite_differences():
  rs = np.random.RandomState(6)
  b, n, a = 3, 7, 4
  d1 = rs.standard_normal((b, n, a))
  d2 = rs.standard_normal((b, n, a))
  tau = np.tile((np.arange(n) + 0.5) / n, (b, 1))
  act = rs.randint(a, size=b)
  r = rs.standard_normal(b)
  g = rs.uniform(0, 1, b)
  losses, dd = qo.quantile_q_losses(d1, tau, act, r, g, d2, d2, 1.0)
  for _ in range(10):
    i, j, c = rs.randint(b), rs.randint(n), rs.randint(a)
    h = 1e-6
    e = np.zeros_like(d1)
    e[i, j, c] = h
    fd = (qo.quantile_q_losses(d1 + e, tau, act, r, g, d2, d2, 1.0)[0].sum() -
          qo.quantile_q_losses(d1 - e, tau, act, r, g, d2, d2, 1.0)[0].sum()
          ) / (2 * h)
    assert abs(fd - dd[i, j, c]) < 1e-6


def test_optimizer_closed_forms():
  dt = np.float64
  p = {'a': np.array([1.0, -2.0, 3.0])}
  g = {'a': np.array([0.5, -0.25, 2.0])}
  lr, eps = 0.1, 1e-3
  newp, st = qo.adam_update(p, g, qo.adam_init(p), lr, eps)
  # first Adam step: m_hat = g, v_hat = g^2 -> update = g / (|g| + eps)
  np.testing.assert_allclose(newp['a'], p['a'] - lr * g['a'] /
                             (np.abs(g['a']) + eps), rtol=1e-12)
  assert st['count'] == 1
  newp2, st2 = qo.adam_update(newp, g, st, lr, eps)
  m = (0.1 * g['a'] + 0.9 * 0.1 * g['a']) / (1 - 0.9 ** 2)
  v = (0.001 * g['a'] ** 2 * (1 + 0.999)) / (1 - 0.999 ** 2)
  np.testing.assert_allclose(newp2['a'], newp['a'] - lr * m / (np.sqrt(v) + eps),
                             rtol=1e-12)
  c, n = qo.clip_by_global_norm({'a': np.array([3.0]), 'b': np.array([4.0])}, 10)
  assert n == 5.0 and c['a'][0] == 3.0
  c, n = qo.clip_by_global_norm({'a': np.array([30.0]), 'b': np.array([40.0])},
                                10)
  np.testing.assert_allclose([c['a'][0], c['b'][0]], [6.0, 8.0])
  newp, st = qo.rmsprop_centered_update(p, g, qo.rmsprop_init(p), lr, 0.95, eps)
  mu, nu = 0.05 * g['a'], 0.05 * g['a'] ** 2
  np.testing.assert_allclose(newp['a'], p['a'] - lr * g['a'] /
                             np.sqrt(nu - mu ** 2 + eps), rtol=1e-12)


def test_rainbow_update_f32_runs_and_priorities():
  online, target, batch, w, noises = _setup(np.float32, b=4, seed=7)
  newp, st, out = qo.rainbow_update(online, target, qo.adam_init(online), batch,
                                    w, noises, SUPPORT, A)
  assert out['losses'].dtype == np.float32 and out['priorities'].max() <= 100
  assert all(newp[k].dtype == np.float32 for k in newp)
  assert st['count'] == 1 and np.isfinite(out['loss'])
  # f32 losses agree with the f64 evaluation to 1e-5 relative
  o64 = {k: v.astype(np.float64) for k, v in online.items()}
  t64 = {k: v.astype(np.float64) for k, v in target.items()}
  n64 = [{k: v.astype(np.float64) for k, v in n.items()} for n in noises]
  _, l64, _, _ = qo.rainbow_loss_and_grads(o64, t64, batch, w, n64, SUPPORT, A,
                                           np.float64)
  np.testing.assert_allclose(out['losses'], l64, rtol=1e-5)


def test_torch_cpu_port_matches_oracle():
  """The cpu_baseline port (oracle/qnet_torch_cpu.py) evaluates the oracle's
  arithmetic: same losses (1e-5), same gradient norm, same Adam step."""
  from oracle import qnet_torch_cpu
  online, target, batch, w, noises = _setup(np.float32, b=8, seed=11)
  sup = SUPPORT.astype(np.float32)
  port = qnet_torch_cpu.RainbowTorchCpu(online, target, sup, A)
  out_t = port.update(batch, w, noises)
  newp, st, out = qo.rainbow_update(online, target, qo.adam_init(online), batch,
                                    w, noises, sup, A)
  np.testing.assert_allclose(out_t['losses'], out['losses'], rtol=1e-5)
  np.testing.assert_allclose(out_t['gnorm'], out['gnorm'], rtol=1e-4)
  np.testing.assert_allclose(out_t['priorities'], out['priorities'], rtol=1e-5)
  lr = 0.00025 / 4
  for k in newp:
    diff = np.abs(port.p[k].detach().numpy() - newp[k])
    assert np.percentile(diff, 99) <= 0.02 * lr and diff.max() <= 1.01 * lr, k


@pytest.mark.parametrize('kind', ['c51', 'qr'])
def test_distributional_dense_losses_finite_differences_f64(kind):
  dt = np.float64
  rs = np.random.RandomState(21)
  nq = 9
  quant = (np.arange(nq) + 0.5) / nq
  online = qo.init_params(kind, A, rs, dt, num_atoms=K, num_quantiles=nq)
  target = qo.init_params(kind, A, rs, dt, num_atoms=K, num_quantiles=nq)
  batch = _batch(rs, 3)

  def f(p):
    if kind == 'c51':
      return qo.c51_loss_and_grads(p, target, batch, SUPPORT, A, dt)
    return qo.qr_loss_and_grads(p, target, batch, quant, A, 1.0, dt)

  loss, losses, grads, _ = f(online)
  assert losses.shape == (3,) and np.isfinite(loss)
  for k in ['conv2/w', 'fc1/b', 'fc2/w', 'fc2/b']:
    g = grads[k].reshape(-1)
    for j in np.argsort(-np.abs(g))[:3]:
      h = 1e-5
      vals = []
      for sgn in (+1, -1):
        p2 = dict(online)
        arr = online[k].copy().reshape(-1)
        arr[j] += sgn * h
        p2[k] = arr.reshape(online[k].shape)
        vals.append(f(p2)[0])
      fd = (vals[0] - vals[1]) / (2 * h)
      assert abs(fd - g[j]) <= 2e-6 * max(1.0, abs(g[j])) + 1e-9, (k, j, fd, g[j])


def test_iqn_vs_torch_autograd_f64():
  """IQN oracle (networks.py:264-292, iqn/agent.py:176-216) against an
  independent torch-autograd float64 model (broadcast [B,N,F] formulation,
  NCHW convs, vectorised quantile-regression loss)."""
  dt = np.float64
  rs = np.random.RandomState(31)
  b, n0, n1, n2, kappa = 4, 5, 6, 7, 1.0
  online = qo.init_params('iqn', A, rs, dt)
  target = qo.init_params('iqn', A, rs, dt)
  assert sum(v.size for v in qo.init_params('iqn', 6, rs).values()) == (
      77984 + 64 * 3136 + 3136 + 3136 * 512 + 512 + 512 * 6 + 6)
  batch = _batch(rs, b)
  batch = (batch[0], batch[1], batch[2] * 2.0, batch[3], batch[4])
  taus = [rs.uniform(size=(b, n)) for n in (n0, n1, n2)]
  loss, losses, grads, aux = qo.iqn_loss_and_grads(online, target, batch, taus,
                                                   kappa, dt)
  tp = {k: torch.tensor(v, requires_grad=True) for k, v in online.items()}
  tt = {k: torch.tensor(v) for k, v in target.items()}

  def fwd(p, x_u8, tau):
    x = torch.from_numpy(x_u8.astype(np.float64) / 255.0).permute(0, 3, 1, 2)
    for name, stride in (('conv1', 4), ('conv2', 2), ('conv3', 1)):
      x = torch.relu(torch.nn.functional.conv2d(
          x, p[name + '/w'].permute(3, 2, 0, 1), p[name + '/b'], stride=stride))
    f = x.permute(0, 2, 3, 1).reshape(x.shape[0], -1)
    t = torch.from_numpy(tau)
    # float32 multiples of pi, as jnp.arange(..., float32) * jnp.pi (networks.py:277)
    i = torch.from_numpy((np.arange(1, 65, dtype=np.float32) *
                          np.float32(np.pi)).astype(np.float64))
    emb = torch.relu(torch.cos(t[:, :, None] * i) @ p['emb/w'] + p['emb/b'])
    hin = emb * f[:, None, :]
    return torch.relu(hin @ p['fc1/w'] + p['fc1/b']) @ p['fc2/w'] + p['fc2/b']

  s_tm1, a, r, d, s_t = batch
  q0 = fwd(tp, s_tm1, taus[0])                       # [B,N0,A]
  with torch.no_grad():
    qs = fwd(tt, s_t, taus[1])
    qt = fwd(tt, s_t, taus[2])
  np.testing.assert_allclose(aux['dist_tm1'], q0.detach().numpy(), rtol=1e-9, atol=1e-12)
  np.testing.assert_allclose(aux['dist_t'], qt.numpy(), rtol=1e-9, atol=1e-12)
  idx = torch.arange(b)
  a_star = qs.mean(1).argmax(1)
  tgt = torch.from_numpy(r)[:, None] + torch.from_numpy(d)[:, None] * qt[idx, :, a_star]
  theta = q0[idx, :, torch.from_numpy(a)]            # [B,N0]
  delta = tgt[:, None, :] - theta[:, :, None]        # [B,N0,N2]
  wgt = (torch.from_numpy(taus[0])[:, :, None] - (delta < 0).double()).abs()
  hub = torch.nn.functional.huber_loss(delta, torch.zeros_like(delta),
                                       reduction='none', delta=kappa)
  tl = (wgt * hub).mean(2).sum(1)
  np.testing.assert_allclose(losses, tl.detach().numpy(), rtol=1e-10)
  tl.mean().backward()
  for k in online:
    g = tp[k].grad.numpy()
    assert np.abs(grads[k] - g).max() / max(np.abs(g).max(), 1e-12) < 1e-9, k
  assert np.abs(grads['emb/w']).max() > 0 and np.abs(grads['conv1/w']).max() > 0
