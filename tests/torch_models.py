"""Independent float64 torch-autograd models of the seven learner steps.

TEST INFRASTRUCTURE.  Written with different primitives from oracle/qnet_oracle.py
(NCHW `conv2d`, autograd instead of hand-derived backward passes, a
triangular-kernel form of the Cramer projection, `huber_loss`, broadcasting
instead of loops), following the reference's call sites and SURVEY.md
Appendix A.  tests/test_qnet_golden.py checks that these models reproduce the
frozen fixtures tests/golden/qnet_*.npz.
"""

import numpy as np
import torch

F = torch.nn.functional


def _t(x):
  return torch.from_numpy(np.asarray(x, dtype=np.float64))


def torso(p, x_u8):
  """networks.py:181-204."""
  x = _t(x_u8.astype(np.float64) / 255.0).permute(0, 3, 1, 2)
  for name, stride in (('conv1', 4), ('conv2', 2), ('conv3', 1)):
    x = torch.relu(F.conv2d(x, p[name + '/w'].permute(3, 2, 0, 1),
                            p[name + '/b'], stride=stride))
  return x.permute(0, 2, 3, 1).reshape(x.shape[0], -1)


def mlp(p, x_u8):
  """networks.py:207-221 (vector or shared scalar bias broadcast)."""
  f = torso(p, x_u8)
  return torch.relu(f @ p['fc1/w'] + p['fc1/b']) @ p['fc2/w'] + p['fc2/b']


def rainbow(p, x_u8, noise, support, num_actions):
  """networks.py:224-261."""
  feat = torso(p, x_u8)
  k = len(support)

  def noisy(name, h):
    w = p[name + '/mu/w'] + p[name + '/sigma/w'] * torch.outer(
        _t(noise[name + '/in']), _t(noise[name + '/out']))
    y = h @ w + p[name + '/sigma/b'] * _t(noise[name + '/out'])
    if name + '/mu/b' in p:
      y = y + p[name + '/mu/b']
    return y

  adv = noisy('adv2', torch.relu(noisy('adv1', feat))).reshape(-1, num_actions, k)
  val = noisy('val2', torch.relu(noisy('val1', feat))).reshape(-1, 1, k)
  logits = val + adv - adv.mean(dim=1, keepdim=True)
  q = (torch.softmax(logits, -1) * _t(support)).sum(-1)
  return logits, q


def iqn(p, x_u8, tau):
  """networks.py:264-292."""
  f = torso(p, x_u8)
  i = _t((np.arange(1, 65, dtype=np.float32) * np.float32(np.pi)).astype(np.float64))
  emb = torch.relu(torch.cos(_t(tau)[:, :, None] * i) @ p['emb/w'] + p['emb/b'])
  h = emb * f[:, None, :]
  return torch.relu(h @ p['fc1/w'] + p['fc1/b']) @ p['fc2/w'] + p['fc2/b']


def project(z_p, probs, z_q):
  """rlax.categorical_l2_project on an evenly spaced support, as a triangular
  kernel: m_i = sum_j max(0, 1 - |clip(z_p_j) - z_q_i| / dz) p_j."""
  dz = z_q[1] - z_q[0]
  zc = torch.clamp(z_p, z_q[0], z_q[-1])
  tri = torch.clamp(1 - (zc[None, :] - z_q[:, None]).abs() / dz, 0, 1)
  return (tri * probs[None, :]).sum(-1)


class _ClipGrad(torch.autograd.Function):
  """rlax.clip_gradient: identity forward, clips the incoming gradient."""

  @staticmethod
  def forward(ctx, x, bound):
    ctx.bound = bound
    return x.clone()

  @staticmethod
  def backward(ctx, g):
    return torch.clamp(g, -ctx.bound, ctx.bound), None


def quantile_losses(theta, tau, target, kappa):
  """rlax.quantile_q_learning's regression part: theta [B,N], tau [B,N],
  target [B,M] -> [B]."""
  delta = target[:, None, :] - theta[:, :, None]
  wgt = (tau[:, :, None] - (delta < 0).double()).abs()
  hub = F.huber_loss(delta, torch.zeros_like(delta), reduction='none', delta=kappa)
  return (wgt * hub).mean(2).sum(1)


def learner_step(name, inp, support, quantiles, num_actions):
  """(losses [B], loss, grads dict) of one case of tests/golden/qnet_cases.py."""
  c = inp['case']
  tp = {k: torch.tensor(np.asarray(v, np.float64), requires_grad=True)
        for k, v in inp['online'].items()}
  tt = {k: _t(v) for k, v in inp['target'].items()}
  s_tm1, a, r, d, s_t = inp['batch']
  b = len(a)
  idx = torch.arange(b)
  a_t, r_t, d_t = torch.from_numpy(np.asarray(a)), _t(r), _t(d)
  w = None if inp['weights'] is None else _t(inp['weights'])
  zs = _t(support)

  if name == 'rainbow':   # rainbow/agent.py:85-109
    nz = inp['noises']
    logits_tm1, _ = rainbow(tp, s_tm1, nz[0], support, num_actions)
    with torch.no_grad():
      _, q_sel = rainbow(tp, s_t, nz[1], support, num_actions)
      logits_tgt, _ = rainbow(tt, s_t, nz[2], support, num_actions)
    per = []
    for i in range(b):
      m = project(r_t[i] + d_t[i] * zs,
                  torch.softmax(logits_tgt[i, int(q_sel[i].argmax())], -1), zs)
      per.append(-(m * torch.log_softmax(logits_tm1[i, a[i]], -1)).sum())
    per = torch.stack(per)
    loss = (per * w).mean()
    report = per
  elif name in ('dqn', 'double_q', 'prioritized'):   # dqn/agent.py:85-107 etc.
    q_tm1 = mlp(tp, s_tm1)
    with torch.no_grad():
      q_tgt = mlp(tt, s_t)
      sel = q_tgt if name == 'dqn' else mlp(tp, s_t)
    td = r_t + d_t * q_tgt[idx, sel.argmax(1)] - q_tm1[idx, a_t]
    per = 0.5 * _ClipGrad.apply(td, c['bound']) ** 2
    if name == 'prioritized':
      per = per * w
    loss = per.mean()
    report = td
  elif name == 'c51':   # c51/agent.py:87-107, rlax.categorical_q_learning
    k = len(support)
    lg = mlp(tp, s_tm1).reshape(b, num_actions, k)
    with torch.no_grad():
      lt = mlp(tt, s_t).reshape(b, num_actions, k)
      q_t = (torch.softmax(lt, -1) * zs).sum(-1)
    per = []
    for i in range(b):
      m = project(r_t[i] + d_t[i] * zs, torch.softmax(lt[i, int(q_t[i].argmax())], -1), zs)
      per.append(-(m * torch.log_softmax(lg[i, a[i]], -1)).sum())
    per = torch.stack(per)
    loss = per.mean()
    report = per
  elif name == 'qr':    # qrdqn/agent.py:88-110, quantile-major head (networks.py:308)
    n = len(quantiles)
    dist = mlp(tp, s_tm1).reshape(b, n, num_actions)
    with torch.no_grad():
      dist_t = mlp(tt, s_t).reshape(b, n, num_actions)
    a_star = dist_t.mean(1).argmax(1)
    target = r_t[:, None] + d_t[:, None] * dist_t[idx, :, a_star]
    tau = _t(quantiles)[None, :].expand(b, n)
    per = quantile_losses(dist[idx, :, a_t], tau, target, c['kappa'])
    loss = per.mean()
    report = per
  elif name == 'iqn':   # iqn/agent.py:176-216
    taus = inp['taus']
    q0 = iqn(tp, s_tm1, taus[0])
    with torch.no_grad():
      qs = iqn(tt, s_t, taus[1])
      qt = iqn(tt, s_t, taus[2])
    a_star = qs.mean(1).argmax(1)
    target = r_t[:, None] + d_t[:, None] * qt[idx, :, a_star]
    per = quantile_losses(q0[idx, :, a_t], _t(taus[0]), target, c['kappa'])
    loss = per.mean()
    report = per
  else:
    raise KeyError(name)
  loss.backward()
  grads = {k: v.grad.numpy() if v.grad is not None else np.zeros(v.shape)
           for k, v in tp.items()}
  return report.detach().numpy(), float(loss), grads


def optimizer_step(c, params, grads):
  """optax 0.1.2: chain(clip_by_global_norm, adam) or
  centred rmsprop, one step from a zero state, float64 tensors."""
  p = {k: _t(v) for k, v in params.items()}
  g = {k: _t(v) for k, v in grads.items()}
  gnorm = torch.sqrt(sum((x * x).sum() for x in g.values()))
  if c['opt'] == 'adam':
    if c['max_norm'] > 0 and float(gnorm) >= c['max_norm']:
      g = {k: x * (c['max_norm'] / gnorm) for k, x in g.items()}
    b1, b2 = 0.9, 0.999
    m = {k: (1 - b1) * x for k, x in g.items()}
    v = {k: (1 - b2) * x * x for k, x in g.items()}
    new = {k: p[k] - c['lr'] * (m[k] / (1 - b1)) /
           (torch.sqrt(v[k] / (1 - b2)) + c['eps']) for k in p}
  else:
    dcy = c['decay']
    m = {k: (1 - dcy) * x for k, x in g.items()}
    v = {k: (1 - dcy) * x * x for k, x in g.items()}
    new = {k: p[k] - c['lr'] * g[k] / torch.sqrt(v[k] - m[k] ** 2 + c['eps'])
           for k in p}
  num = lambda d: {k: x.numpy() for k, x in d.items()}
  return num(new), dict(m=num(m), v=num(v)), float(gnorm)
