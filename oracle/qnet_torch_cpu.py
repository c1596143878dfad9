"""Multi-threaded CPU port of the Rainbow update (TEST INFRASTRUCTURE ONLY).

Used for `bench.py`'s `cpu_baseline` leg as the stand-in for the reference's
`--jax_platform_name=cpu` path: JAX/XLA is not installable here, and a plain
NumPy oracle would be unfairly slow next to XLA-CPU's threaded Eigen kernels.
This file evaluates exactly the arithmetic of oracle/qnet_oracle.py
(rainbow_update) with torch CPU ops (oneDNN conv, threaded GEMM, autograd) and
is pinned to it by tests/test_oracle_qnet.py::test_torch_cpu_port_matches_oracle.
"""

import numpy as np
import torch
import torch.nn.functional as F


class RainbowTorchCpu:

  def __init__(self, params, target, support, num_actions, lr=0.00025 / 4,
               eps=0.005 / 32, max_norm=10.0):
    self.p = {k: torch.tensor(v, dtype=torch.float32, requires_grad=True)
              for k, v in params.items()}
    self.t = {k: torch.tensor(v, dtype=torch.float32) for k, v in target.items()}
    self.support = torch.tensor(np.asarray(support), dtype=torch.float32)
    self.a = num_actions
    self.k = len(support)
    self.lr, self.eps, self.max_norm = lr, eps, max_norm
    self.m = {k: torch.zeros_like(v) for k, v in self.p.items()}
    self.v = {k: torch.zeros_like(v) for k, v in self.p.items()}
    self.count = 0

  def _net(self, p, x_u8, nz):
    x = torch.from_numpy(x_u8).to(torch.float32).div(255.0).permute(0, 3, 1, 2)
    for name, stride in (('conv1', 4), ('conv2', 2), ('conv3', 1)):
      x = F.relu(F.conv2d(x, p[name + '/w'].permute(3, 2, 0, 1), p[name + '/b'],
                          stride=stride))
    feat = x.permute(0, 2, 3, 1).reshape(x.shape[0], -1)

    def noisy(name, h):
      ein = torch.from_numpy(nz[name + '/in'])
      eout = torch.from_numpy(nz[name + '/out'])
      mu = h @ p[name + '/mu/w']
      if name + '/mu/b' in p:
        mu = mu + p[name + '/mu/b']
      sig = ((h * ein) @ p[name + '/sigma/w'] + p[name + '/sigma/b']) * eout
      return mu + sig

    adv = noisy('adv2', F.relu(noisy('adv1', feat))).reshape(-1, self.a, self.k)
    val = noisy('val2', F.relu(noisy('val1', feat))).reshape(-1, 1, self.k)
    logits = val + adv - adv.mean(dim=1, keepdim=True)
    q = (torch.softmax(logits, -1) * self.support).sum(-1)
    return logits, q

  def _project(self, z_p, probs):
    z = self.support
    dz_pos = torch.roll(z, -1) - z
    dz_neg = z - torch.roll(z, 1)
    r_pos = torch.where(dz_pos > 0, 1.0 / dz_pos, torch.zeros_like(z))[None, :, None]
    r_neg = torch.where(dz_neg > 0, 1.0 / dz_neg, torch.zeros_like(z))[None, :, None]
    zc = torch.clamp(z_p, z[0], z[-1])[:, None, :]
    delta = zc - z[None, :, None]
    dh = torch.where(delta >= 0, delta * r_pos, -(delta * r_neg))
    return (torch.clamp(1.0 - dh, 0, 1) * probs[:, None, :]).sum(-1)

  def update(self, batch, weights, noises):
    s_tm1, a_tm1, r_t, d_t, s_t = batch
    b = len(a_tm1)
    idx = torch.arange(b)
    a_tm1 = torch.from_numpy(np.asarray(a_tm1))
    r = torch.from_numpy(np.asarray(r_t)).to(torch.float32)
    d = torch.from_numpy(np.asarray(d_t)).to(torch.float32)
    w = torch.from_numpy(np.asarray(weights)).to(torch.float32)
    logits_tm1, _ = self._net(self.p, s_tm1, noises[0])
    with torch.no_grad():
      _, q_t = self._net(self.p, s_t, noises[1])
      logits_tgt, _ = self._net(self.t, s_t, noises[2])
      a_star = torch.argmax(q_t, dim=1)
      p_t = torch.softmax(logits_tgt[idx, a_star], -1)
      m = self._project(r[:, None] + d[:, None] * self.support[None, :], p_t)
    losses = -(m * torch.log_softmax(logits_tm1[idx, a_tm1], -1)).sum(-1)
    loss = (losses * w).mean()
    grads = torch.autograd.grad(loss, list(self.p.values()))
    gnorm = torch.sqrt(sum((g * g).sum() for g in grads))
    if self.max_norm > 0 and not bool(gnorm < self.max_norm):
      grads = [(g / gnorm) * self.max_norm for g in grads]
    self.count += 1
    bc1 = 1.0 - 0.9 ** self.count
    bc2 = 1.0 - 0.999 ** self.count
    with torch.no_grad():
      for (k, prm), g in zip(self.p.items(), grads):
        self.m[k].mul_(0.9).add_(g, alpha=0.1)
        self.v[k].mul_(0.999).addcmul_(g, g, value=0.001)
        upd = (self.m[k] / bc1) / (torch.sqrt(self.v[k] / bc2) + self.eps)
        prm.add_(upd, alpha=-self.lr)
    ld = losses.detach()
    return dict(loss=float(loss.detach()), losses=ld.numpy(), gnorm=float(gnorm),
                priorities=torch.clamp(ld.abs(), 0, 100).numpy())
