"""CPU oracle for the Q-loss / update half of the hot path (TEST INFRASTRUCTURE ONLY).

NumPy restatement of what the reference's jitted `update` computes:
networks (dqn_zoo/networks.py:58-363), the agents' loss_fn / update
(dqn/agent.py:85-117, double_q/agent.py:85-111, prioritized/agent.py:86-113,
c51/agent.py:87-107, qrdqn/agent.py:88-110, rainbow/agent.py:85-121) and the
third-party arithmetic they call.  Only tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg may import this file.

PARITY UNPINNED.  The arithmetic of this half lives in un-vendored third-party
packages that are absent from /root/reference and not installable here:
rlax==0.1.2, optax==0.1.2, dm-haiku==0.0.6, jax==0.3.10
(docker_requirements.txt:6-16).  Their published semantics are restated below
(SURVEY.md Appendix A); the reference's own tests pin no numeric values for
this half (networks_test.py checks shapes/names/sigma-init only).  The oracle is
validated instead by tests/test_oracle_qnet.py: float64 central finite
differences of every loss, an independent torch-autograd float64 model,
projection invariants, optimiser closed forms and the networks_test.py pins.

Conventions: activations NHWC, conv weights HWIO (networks_test.py:53), linear
weights (in, out) (networks_test.py:44), flatten order (h, w, c).  All functions
take `dt` (np.float32 for parity runs, np.float64 for derivative checks) and
explicit noise arrays -- the JAX threefry key schedule cannot be reproduced, so
noise is an INPUT to both the oracle and the HIP kernels.
"""

import numpy as np

# --------------------------------------------------------------------------- #
#  Layers
# --------------------------------------------------------------------------- #


def _patches(x, kh, kw, stride):
  """x[B,H,W,C] -> [B,OH,OW,kh,kw,C] view (VALID padding)."""
  b, h, w, c = x.shape
  oh = (h - kh) // stride + 1
  ow = (w - kw) // stride + 1
  s = x.strides
  return np.lib.stride_tricks.as_strided(
      x, (b, oh, ow, kh, kw, c),
      (s[0], s[1] * stride, s[2] * stride, s[1], s[2], s[3]), writeable=False)


def conv_fwd(x, w, b, stride):
  """hk.Conv2D, VALID, NHWC/HWIO cross-correlation (networks.py:82-103)."""
  kh, kw, ci, co = w.shape
  p = _patches(x, kh, kw, stride)
  bsz, oh, ow = p.shape[:3]
  cols = p.reshape(bsz * oh * ow, kh * kw * ci)
  y = cols @ w.reshape(kh * kw * ci, co) + b
  return y.reshape(bsz, oh, ow, co), cols


def conv_bwd(dy, cols, x_shape, w, stride, need_dx):
  kh, kw, ci, co = w.shape
  bsz, oh, ow, _ = dy.shape
  dy2 = dy.reshape(bsz * oh * ow, co)
  dw = (cols.T @ dy2).reshape(w.shape)
  db = dy2.sum(axis=0)
  dx = None
  if need_dx:
    dcols = (dy2 @ w.reshape(kh * kw * ci, co).T).reshape(bsz, oh, ow, kh, kw, ci)
    dx = np.zeros(x_shape, dtype=dy.dtype)
    for i in range(kh):
      for j in range(kw):
        dx[:, i:i + stride * oh:stride, j:j + stride * ow:stride, :] += \
            dcols[:, :, :, i, j, :]
  return dx, dw, db


def relu(x):
  return np.maximum(x, 0)


# --------------------------------------------------------------------------- #
#  Parameter initialisation (networks.py:58-79, 137-178)
# --------------------------------------------------------------------------- #
TORSO_SPEC = [('conv1', 8, 4, 4, 32), ('conv2', 4, 2, 32, 64),
              ('conv3', 3, 1, 64, 64)]
FLAT = 3136  # 7*7*64


def _uniform(rs, shape, fan_in, dt):
  c = np.sqrt(1.0 / fan_in)
  return rs.uniform(-c, c, size=shape).astype(dt)


def init_torso(rs, dt=np.float32):
  p = {}
  for name, k, _, ci, co in TORSO_SPEC:
    fan = ci * k * k
    p[name + '/w'] = _uniform(rs, (k, k, ci, co), fan, dt)
    p[name + '/b'] = _uniform(rs, (co,), fan, dt)
  return p


def init_linear(rs, p, name, nin, nout, dt, bias='vector'):
  p[name + '/w'] = _uniform(rs, (nin, nout), nin, dt)
  if bias == 'vector':
    p[name + '/b'] = _uniform(rs, (nout,), nin, dt)
  elif bias == 'shared':  # networks.py:120-134, shape (1,)
    p[name + '/b'] = _uniform(rs, (1,), nin, dt)


def init_noisy(rs, p, name, nin, nout, dt, with_bias, sigma0=0.1):
  p[name + '/mu/w'] = _uniform(rs, (nin, nout), nin, dt)
  if with_bias:
    p[name + '/mu/b'] = _uniform(rs, (nout,), nin, dt)
  s = dt(sigma0 / np.sqrt(nin))  # networks.py:157-159
  p[name + '/sigma/w'] = np.full((nin, nout), s, dt)
  p[name + '/sigma/b'] = np.full((nout,), s, dt)


def init_params(kind, num_actions, rs, dt=np.float32, num_atoms=51,
                num_quantiles=201):
  """kind in {dqn, double_dqn, c51, qr, iqn, rainbow}."""
  p = init_torso(rs, dt)
  a = num_actions
  if kind == 'iqn':  # creation order: torso, tau embedding, value head (networks.py:272-287)
    init_linear(rs, p, 'emb', IQN_LATENT, FLAT, dt)
    init_linear(rs, p, 'fc1', FLAT, 512, dt)
    init_linear(rs, p, 'fc2', 512, a, dt)
    return p
  if kind == 'rainbow':  # creation order adv1, adv2, val1, val2 (networks.py:239-251)
    init_noisy(rs, p, 'adv1', FLAT, 512, dt, True)
    init_noisy(rs, p, 'adv2', 512, a * num_atoms, dt, False)
    init_noisy(rs, p, 'val1', FLAT, 512, dt, True)
    init_noisy(rs, p, 'val2', 512, num_atoms, dt, False)
    return p
  init_linear(rs, p, 'fc1', FLAT, 512, dt)
  nout = {'dqn': a, 'double_dqn': a, 'c51': a * num_atoms,
          'qr': a * num_quantiles}[kind]
  init_linear(rs, p, 'fc2', 512, nout, dt,
              bias='shared' if kind == 'double_dqn' else 'vector')
  return p


def noise_shapes(num_actions, num_atoms=51):
  """Per-apply noise arrays of the Rainbow net, in layer creation order; each
  noisy layer draws input noise then output noise (networks.py:169-170)."""
  return [('adv1/in', FLAT), ('adv1/out', 512),
          ('adv2/in', 512), ('adv2/out', num_actions * num_atoms),
          ('val1/in', FLAT), ('val1/out', 512),
          ('val2/in', 512), ('val2/out', num_atoms)]


def sample_noise(rs, num_actions, dt=np.float32, num_atoms=51):
  """f(n) = sign(n) sqrt|n|, n ~ truncated normal on [-2, 2] (networks.py:142-144)."""
  out = {}
  for name, n in noise_shapes(num_actions, num_atoms):
    x = rs.standard_normal(4 * n + 64)
    x = x[np.abs(x) <= 2.0][:n]
    out[name] = (np.sign(x) * np.sqrt(np.abs(x))).astype(dt)
  return out


# --------------------------------------------------------------------------- #
#  Networks: forward with caches, and backward
# --------------------------------------------------------------------------- #
def torso_fwd(p, x_u8, dt):
  x = x_u8.astype(dt) / dt(255.0)  # networks.py:193
  cache = {'x': x}
  h = x
  for name, _, stride, _, _ in TORSO_SPEC:
    y, cols = conv_fwd(h, p[name + '/w'], p[name + '/b'], stride)
    cache[name] = (cols, h.shape, y)
    h = relu(y)
  return h.reshape(h.shape[0], -1), cache  # hk.Flatten: (h, w, c) order


def torso_bwd(p, cache, dflat, grads):
  d = dflat.reshape(cache['conv3'][2].shape)
  for idx in (2, 1, 0):
    name, _, stride, _, _ = TORSO_SPEC[idx]
    cols, xshape, y = cache[name]
    d = d * (y > 0)
    d, dw, db = conv_bwd(d, cols, xshape, p[name + '/w'], stride, idx > 0)
    grads[name + '/w'] = dw
    grads[name + '/b'] = db


def noisy_fwd(p, name, x, eps_in, eps_out):
  """networks.py:168-176, computed in the reference's two-GEMM form."""
  mu = x @ p[name + '/mu/w']
  if name + '/mu/b' in p:
    mu = mu + p[name + '/mu/b']
  xn = eps_in[None, :] * x
  sig = (xn @ p[name + '/sigma/w'] + p[name + '/sigma/b']) * eps_out[None, :]
  return mu + sig


def noisy_bwd(p, name, x, eps_in, eps_out, dy, grads):
  grads[name + '/mu/w'] = x.T @ dy
  if name + '/mu/b' in p:
    grads[name + '/mu/b'] = dy.sum(axis=0)
  dsig = dy * eps_out[None, :]
  xn = eps_in[None, :] * x
  grads[name + '/sigma/w'] = xn.T @ dsig
  grads[name + '/sigma/b'] = dsig.sum(axis=0)
  return dy @ p[name + '/mu/w'].T + (dsig @ p[name + '/sigma/w'].T) * eps_in[None, :]


def softmax(z, axis=-1):
  z = z - z.max(axis=axis, keepdims=True)
  e = np.exp(z)
  return e / e.sum(axis=axis, keepdims=True)


def log_softmax(z, axis=-1):
  z = z - z.max(axis=axis, keepdims=True)
  return z - np.log(np.exp(z).sum(axis=axis, keepdims=True))


def rainbow_fwd(p, x_u8, noise, support, num_actions, dt=np.float32):
  """networks.py:224-261.  Returns (q_logits [B,A,K], q_values [B,A], cache)."""
  k = len(support)
  feat, tc = torso_fwd(p, x_u8, dt)
  a1 = noisy_fwd(p, 'adv1', feat, noise['adv1/in'], noise['adv1/out'])
  ha = relu(a1)
  a2 = noisy_fwd(p, 'adv2', ha, noise['adv2/in'], noise['adv2/out'])
  adv = a2.reshape(-1, num_actions, k)
  v1 = noisy_fwd(p, 'val1', feat, noise['val1/in'], noise['val1/out'])
  hv = relu(v1)
  v2 = noisy_fwd(p, 'val2', hv, noise['val2/in'], noise['val2/out'])
  val = v2.reshape(-1, 1, k)
  logits = val + adv - adv.mean(axis=-2, keepdims=True)
  q = (softmax(logits) * support[None, None, :]).sum(axis=2)
  cache = dict(torso=tc, feat=feat, a1=a1, ha=ha, v1=v1, hv=hv)
  return logits, q, cache


def rainbow_bwd(p, cache, noise, dlogits, num_actions):
  """Gradient of sum(dlogits * q_logits) w.r.t. params (q_values carry
  stop_gradient, networks.py:258)."""
  grads = {}
  bsz, a, k = dlogits.shape
  dval = dlogits.sum(axis=1)                                  # [B,K]
  dadv = dlogits - dlogits.mean(axis=1, keepdims=True)        # [B,A,K]
  dhv = noisy_bwd(p, 'val2', cache['hv'], noise['val2/in'], noise['val2/out'],
                  dval, grads)
  dv1 = dhv * (cache['v1'] > 0)
  dfeat = noisy_bwd(p, 'val1', cache['feat'], noise['val1/in'],
                    noise['val1/out'], dv1, grads)
  dha = noisy_bwd(p, 'adv2', cache['ha'], noise['adv2/in'], noise['adv2/out'],
                  dadv.reshape(bsz, a * k), grads)
  da1 = dha * (cache['a1'] > 0)
  dfeat = dfeat + noisy_bwd(p, 'adv1', cache['feat'], noise['adv1/in'],
                            noise['adv1/out'], da1, grads)
  torso_bwd(p, cache['torso'], dfeat, grads)
  return grads


IQN_LATENT = 64  # iqn/run_atari.py:97 tau_latent_dim


def iqn_fwd(p, x_u8, taus, dt=np.float32):
  """iqn_atari_network (networks.py:264-292).  taus [B,N] -> q_dist [B,N,A],
  q_values [B,A] (mean over samples)."""
  feat, tc = torso_fwd(p, x_u8, dt)
  taus = np.asarray(taus).astype(dt)
  b, n = taus.shape
  latent = p['emb/w'].shape[0]
  pi_mult = (np.arange(1, latent + 1, dtype=np.float32) * np.float32(np.pi)).astype(dt)
  cosemb = np.cos(pi_mult[None, None, :] * taus[:, :, None]).astype(dt)   # [B,N,L]
  e2 = cosemb.reshape(b * n, latent)
  zt = e2 @ p['emb/w'] + p['emb/b']
  temb = relu(zt)                                                         # [B*N,F]
  hin = temb * np.repeat(feat, n, axis=0)
  z1 = hin @ p['fc1/w'] + p['fc1/b']
  h = relu(z1)
  out = h @ p['fc2/w'] + p['fc2/b']
  q_dist = out.reshape(b, n, -1)
  cache = dict(torso=tc, feat=feat, cosemb=e2, zt=zt, temb=temb, hin=hin, z1=z1, h=h,
               n=n)
  return q_dist, q_dist.mean(axis=1), cache


def iqn_bwd(p, cache, dq_dist):
  """Gradients of sum(q_dist * dq_dist) wrt the parameters (taus are data)."""
  grads = {}
  n = cache['n']
  dout = dq_dist.reshape(-1, dq_dist.shape[-1])
  grads['fc2/w'] = cache['h'].T @ dout
  grads['fc2/b'] = dout.sum(axis=0)
  dz1 = (dout @ p['fc2/w'].T) * (cache['z1'] > 0)
  grads['fc1/w'] = cache['hin'].T @ dz1
  grads['fc1/b'] = dz1.sum(axis=0)
  dhin = dz1 @ p['fc1/w'].T
  feat = cache['feat']
  b = feat.shape[0]
  dzt = dhin * np.repeat(feat, n, axis=0) * (cache['zt'] > 0)
  grads['emb/w'] = cache['cosemb'].T @ dzt
  grads['emb/b'] = dzt.sum(axis=0)
  dfeat = (dhin * cache['temb']).reshape(b, n, -1).sum(axis=1)
  torso_bwd(p, cache['torso'], dfeat, grads)
  return grads


def mlp_head_fwd(p, x_u8, dt=np.float32):
  """dqn_value_head on the torso (networks.py:207-221, 338-363).  fc2 bias may
  be a shared scalar of shape (1,) (networks.py:120-134)."""
  feat, tc = torso_fwd(p, x_u8, dt)
  z1 = feat @ p['fc1/w'] + p['fc1/b']
  h = relu(z1)
  out = h @ p['fc2/w'] + p['fc2/b']
  return out, dict(torso=tc, feat=feat, z1=z1, h=h)


def mlp_head_bwd(p, cache, dout):
  grads = {}
  grads['fc2/w'] = cache['h'].T @ dout
  shared = p['fc2/b'].shape == (1,) and p['fc2/w'].shape[1] != 1
  grads['fc2/b'] = dout.sum().reshape(1) if shared else dout.sum(axis=0)
  dh = dout @ p['fc2/w'].T
  dz1 = dh * (cache['z1'] > 0)
  grads['fc1/w'] = cache['feat'].T @ dz1
  grads['fc1/b'] = dz1.sum(axis=0)
  torso_bwd(p, cache['torso'], dz1 @ p['fc1/w'].T, grads)
  return grads


# --------------------------------------------------------------------------- #
#  rlax 0.1.2 losses (per sample)
# --------------------------------------------------------------------------- #
def categorical_l2_project(z_p, probs, z_q):
  """Cramer/L2 projection of (z_p, probs) onto support z_q."""
  dt = probs.dtype
  d_pos = np.roll(z_q, -1) - z_q
  d_neg = z_q - np.roll(z_q, 1)
  z_p = np.clip(z_p, z_q[0], z_q[-1])[None, :]
  with np.errstate(divide='ignore'):
    r_pos = np.where(d_pos > 0, dt.type(1) / d_pos, dt.type(0))[:, None]
    r_neg = np.where(d_neg > 0, dt.type(1) / d_neg, dt.type(0))[:, None]
  delta = z_p - z_q[:, None]              # [Kq, Kp]
  sign = (delta >= 0).astype(dt)
  delta_hat = sign * delta * r_pos - (dt.type(1) - sign) * delta * r_neg
  return (np.clip(dt.type(1) - delta_hat, 0, 1) * probs[None, :]).sum(axis=-1)


def categorical_double_q_losses(support, logits_tm1, a_tm1, r_t, d_t,
                                logits_target_t, q_selector_t):
  """vmap(rlax.categorical_double_q_learning) (rainbow/agent.py:97-106).
  Returns (losses [B], dlosses/dlogits_tm1 [B,A,K], targets [B,K])."""
  bsz, a, k = logits_tm1.shape
  dt = logits_tm1.dtype
  losses = np.zeros(bsz, dt)
  dlog = np.zeros_like(logits_tm1)
  targets = np.zeros((bsz, k), dt)
  for i in range(bsz):
    target_z = r_t[i] + d_t[i] * support
    a_star = int(np.argmax(q_selector_t[i]))
    p_target = softmax(logits_target_t[i, a_star])
    m = categorical_l2_project(target_z, p_target, support)
    lq = logits_tm1[i, a_tm1[i]]
    losses[i] = -(m * log_softmax(lq)).sum()
    dlog[i, a_tm1[i]] = softmax(lq) * m.sum() - m
    targets[i] = m
  return losses, dlog, targets


def categorical_q_losses(support, logits_tm1, a_tm1, r_t, d_t, logits_target_t):
  """vmap(rlax.categorical_q_learning) (c51/agent.py:96-104): the selector is
  the target network's own expectation."""
  q_t = (softmax(logits_target_t) * support[None, None, :]).sum(axis=2)
  return categorical_double_q_losses(support, logits_tm1, a_tm1, r_t, d_t,
                                     logits_target_t, q_t)


def td_errors_q(q_tm1, a_tm1, r_t, d_t, q_target_t, q_selector_t=None):
  """rlax.q_learning / double_q_learning td errors."""
  idx = np.arange(q_tm1.shape[0])
  if q_selector_t is None:
    boot = q_target_t.max(axis=1)
  else:
    boot = q_target_t[idx, np.argmax(q_selector_t, axis=1)]
  return r_t + d_t * boot - q_tm1[idx, a_tm1]


def clipped_l2_loss_and_grad(td, weights, bound):
  """loss = mean(0.5 td^2 * w) with rlax.clip_gradient(td, -bound, bound)
  (dqn/agent.py:94-106, prioritized/agent.py:105-112).  The clip acts on the
  gradient ARRIVING at td, i.e. on w*td/B.  Returns (loss, dloss/dq_tm1[a])."""
  bsz = td.shape[0]
  dt = td.dtype
  w = np.ones(bsz, dt) if weights is None else weights
  loss = (dt.type(0.5) * td * td * w).mean()
  g_td = np.clip(td * w / dt.type(bsz), -bound, bound)
  return loss, -g_td  # d td / d q_tm1[a] = -1


def quantile_q_losses(dist_tm1, tau, a_tm1, r_t, d_t, dist_sel_t, dist_t, kappa):
  """vmap(rlax.quantile_q_learning) (qrdqn/agent.py:98-107).  dist_*: [B,N,A].
  Returns (losses [B], dlosses/ddist_tm1 [B,N,A])."""
  bsz, n, a = dist_tm1.shape
  dt = dist_tm1.dtype
  losses = np.zeros(bsz, dt)
  dd = np.zeros_like(dist_tm1)
  for i in range(bsz):
    a_star = int(np.argmax(dist_sel_t[i].mean(axis=0)))
    tgt = r_t[i] + d_t[i] * dist_t[i][:, a_star]          # [Nt]
    theta = dist_tm1[i][:, a_tm1[i]]                      # [N]
    delta = tgt[None, :] - theta[:, None]                 # [N, Nt]
    wgt = np.abs(tau[i][:, None] - (delta < 0).astype(dt))
    absd = np.abs(delta)
    if kappa > 0:
      q = np.minimum(absd, kappa)
      hub = dt.type(0.5) * q * q + kappa * (absd - q)
      dh = np.where(absd <= kappa, delta, kappa * np.sign(delta))
    else:
      hub = absd
      dh = np.sign(delta)
    losses[i] = (wgt * hub).mean(axis=1).sum()
    dd[i][:, a_tm1[i]] = -(wgt * dh).mean(axis=1)
  return losses, dd


# --------------------------------------------------------------------------- #
#  optax 0.1.2 transforms
# --------------------------------------------------------------------------- #
def global_norm(grads):
  return np.sqrt(sum((g.astype(g.dtype) ** 2).sum() for g in grads.values()))


def clip_by_global_norm(grads, max_norm):
  dt = next(iter(grads.values())).dtype
  n = dt.type(global_norm(grads))
  if n < max_norm:
    return dict(grads), n
  return {k: (g / n) * dt.type(max_norm) for k, g in grads.items()}, n


def adam_init(params):
  return dict(count=0, mu={k: np.zeros_like(v) for k, v in params.items()},
              nu={k: np.zeros_like(v) for k, v in params.items()})


def adam_update(params, grads, state, lr, eps, b1=0.9, b2=0.999):
  dt = next(iter(params.values())).dtype
  c = state['count'] + 1
  b1, b2, lr, eps = dt.type(b1), dt.type(b2), dt.type(lr), dt.type(eps)
  bc1 = dt.type(1) - b1 ** dt.type(c)
  bc2 = dt.type(1) - b2 ** dt.type(c)
  new_p, mu, nu = {}, {}, {}
  for k, g in grads.items():
    mu[k] = (dt.type(1) - b1) * g + b1 * state['mu'][k]
    nu[k] = (dt.type(1) - b2) * (g * g) + b2 * state['nu'][k]
    upd = (mu[k] / bc1) / (np.sqrt(nu[k] / bc2) + eps)
    new_p[k] = params[k] + (-lr) * upd
  return new_p, dict(count=c, mu=mu, nu=nu)


def rmsprop_init(params):
  return dict(mu={k: np.zeros_like(v) for k, v in params.items()},
              nu={k: np.zeros_like(v) for k, v in params.items()})


def rmsprop_centered_update(params, grads, state, lr, decay, eps):
  """optax.rmsprop(lr, decay, eps, centered=True): eps INSIDE the sqrt."""
  dt = next(iter(params.values())).dtype
  lr, decay, eps = dt.type(lr), dt.type(decay), dt.type(eps)
  new_p, mu, nu = {}, {}, {}
  for k, g in grads.items():
    mu[k] = (dt.type(1) - decay) * g + decay * state['mu'][k]
    nu[k] = (dt.type(1) - decay) * (g * g) + decay * state['nu'][k]
    upd = g / np.sqrt(nu[k] - mu[k] * mu[k] + eps)
    new_p[k] = params[k] + (-lr) * upd
  return new_p, dict(mu=mu, nu=nu)


# --------------------------------------------------------------------------- #
#  Whole-agent learner steps
# --------------------------------------------------------------------------- #
def rainbow_loss_and_grads(online, target, batch, weights, noises, support,
                           num_actions, dt=np.float32):
  """rainbow/agent.py:85-109.  `noises` = 3 noise dicts for the applies
  online(s_tm1), online(s_t), target(s_t) (three different keys, :87-96).
  batch = (s_tm1 u8, a_tm1 int, r_t, discount_t, s_t u8)."""
  s_tm1, a_tm1, r_t, d_t, s_t = batch
  r_t = np.asarray(r_t).astype(dt)       # float64 -> float32 at the jit boundary
  d_t = np.asarray(d_t).astype(dt)
  weights = np.asarray(weights).astype(dt)
  support = support.astype(dt)
  logits_tm1, _, cache = rainbow_fwd(online, s_tm1, noises[0], support,
                                     num_actions, dt)
  _, q_t, _ = rainbow_fwd(online, s_t, noises[1], support, num_actions, dt)
  logits_tgt, _, _ = rainbow_fwd(target, s_t, noises[2], support, num_actions,
                                 dt)
  losses, dlog, _ = categorical_double_q_losses(
      support, logits_tm1, np.asarray(a_tm1), r_t, d_t, logits_tgt, q_t)
  loss = (losses * weights).mean()
  bsz = losses.shape[0]
  dlogits = dlog * (weights / dt(bsz))[:, None, None]
  grads = rainbow_bwd(online, cache, noises[0], dlogits, num_actions)
  aux = dict(logits_tm1=logits_tm1, q_t=q_t, logits_target=logits_tgt,
             dlogits=dlogits)
  return loss, losses, grads, aux


def rainbow_update(online, target, opt_state, batch, weights, noises, support,
                   num_actions, lr=0.00025 / 4, eps=0.005 / 32,
                   max_norm=10.0, dt=np.float32):
  """rainbow/agent.py:111-121 with optax.chain(clip_by_global_norm, adam)
  (rainbow/run_atari.py:229-235).  Returns new params, opt state, losses and
  the priorities clip(|loss|, 0, 100) (rainbow/agent.py:194)."""
  loss, losses, grads, aux = rainbow_loss_and_grads(
      online, target, batch, weights, noises, support, num_actions, dt)
  if max_norm > 0:
    clipped, gnorm = clip_by_global_norm(grads, max_norm)
  else:
    clipped, gnorm = grads, global_norm(grads)
  new_p, new_s = adam_update(online, clipped, opt_state, lr, eps)
  priorities = np.clip(np.abs(losses), 0.0, 100.0)
  return new_p, new_s, dict(loss=loss, losses=losses, grads=grads,
                            gnorm=gnorm, priorities=priorities, **aux)


def dqn_family_loss_and_grads(kind, online, target, batch, weights, bound,
                              dt=np.float32):
  """kind='dqn' (dqn/agent.py:85-107), 'double_q' (double_q/agent.py:85-111),
  'prioritized' (prioritized/agent.py:86-113; weights required)."""
  s_tm1, a_tm1, r_t, d_t, s_t = batch
  r_t = np.asarray(r_t).astype(dt)
  d_t = np.asarray(d_t).astype(dt)
  a_tm1 = np.asarray(a_tm1)
  q_tm1, cache = mlp_head_fwd(online, s_tm1, dt)
  q_tgt, _ = mlp_head_fwd(target, s_t, dt)
  q_sel = None
  if kind in ('double_q', 'prioritized'):
    q_sel, _ = mlp_head_fwd(online, s_t, dt)
  td = td_errors_q(q_tm1, a_tm1, r_t, d_t, q_tgt, q_sel)
  w = None if weights is None else np.asarray(weights).astype(dt)
  loss, dq_a = clipped_l2_loss_and_grad(td, w, dt(bound))
  dq = np.zeros_like(q_tm1)
  dq[np.arange(len(a_tm1)), a_tm1] = dq_a
  grads = mlp_head_bwd(online, cache, dq)
  return loss, td, grads, dict(q_tm1=q_tm1, q_target=q_tgt, q_sel=q_sel)


def c51_loss_and_grads(online, target, batch, support, num_actions, dt=np.float32):
  """c51/agent.py:87-107: categorical_q_learning on the dense C51 head
  (networks.py:316-335: q_logits = reshape(head, (-1, A, K)))."""
  s_tm1, a_tm1, r_t, d_t, s_t = batch
  r_t = np.asarray(r_t).astype(dt)
  d_t = np.asarray(d_t).astype(dt)
  support = support.astype(dt)
  k = len(support)
  out_tm1, cache = mlp_head_fwd(online, s_tm1, dt)
  out_tgt, _ = mlp_head_fwd(target, s_t, dt)
  bsz = out_tm1.shape[0]
  losses, dlog, _ = categorical_q_losses(
      support, out_tm1.reshape(bsz, num_actions, k), np.asarray(a_tm1), r_t, d_t,
      out_tgt.reshape(bsz, num_actions, k))
  loss = losses.mean()
  grads = mlp_head_bwd(online, cache, (dlog / dt(bsz)).reshape(bsz, -1))
  return loss, losses, grads, dict(out_tm1=out_tm1, out_target=out_tgt)


def qr_loss_and_grads(online, target, batch, quantiles, num_actions, kappa,
                      dt=np.float32):
  """qrdqn/agent.py:88-110: quantile_q_learning, no double-Q; head output is
  reshaped (-1, N, A) -- quantile-major (networks.py:308)."""
  s_tm1, a_tm1, r_t, d_t, s_t = batch
  r_t = np.asarray(r_t).astype(dt)
  d_t = np.asarray(d_t).astype(dt)
  n = len(quantiles)
  out_tm1, cache = mlp_head_fwd(online, s_tm1, dt)
  out_tgt, _ = mlp_head_fwd(target, s_t, dt)
  bsz = out_tm1.shape[0]
  dist_tm1 = out_tm1.reshape(bsz, n, num_actions)
  dist_t = out_tgt.reshape(bsz, n, num_actions)
  tau = np.tile(quantiles.astype(dt)[None, :], (bsz, 1))
  losses, dd = quantile_q_losses(dist_tm1, tau, np.asarray(a_tm1), r_t, d_t, dist_t,
                                 dist_t, dt(kappa))
  loss = losses.mean()
  grads = mlp_head_bwd(online, cache, (dd / dt(bsz)).reshape(bsz, -1))
  return loss, losses, grads, dict(out_tm1=out_tm1, out_target=out_tgt)


def iqn_loss_and_grads(online, target, batch, taus, kappa, dt=np.float32):
  """iqn/agent.py:176-216: three tau sets (tau_tm1, tau_t_selector, tau_t), the
  selector and the target distribution both from the TARGET network on s_t,
  vmap(rlax.quantile_q_learning), mean over the batch."""
  s_tm1, a_tm1, r_t, d_t, s_t = batch
  r_t = np.asarray(r_t).astype(dt)
  d_t = np.asarray(d_t).astype(dt)
  tau_tm1, tau_sel, tau_t = [np.asarray(t).astype(dt) for t in taus]
  dist_tm1, _, cache = iqn_fwd(online, s_tm1, tau_tm1, dt)
  dist_sel, _, _ = iqn_fwd(target, s_t, tau_sel, dt)
  dist_t, _, _ = iqn_fwd(target, s_t, tau_t, dt)
  losses, dd = quantile_q_losses(dist_tm1, tau_tm1, np.asarray(a_tm1), r_t, d_t,
                                 dist_sel, dist_t, dt(kappa))
  bsz = losses.shape[0]
  loss = losses.mean()
  grads = iqn_bwd(online, cache, dd / dt(bsz))
  return loss, losses, grads, dict(dist_tm1=dist_tm1, dist_sel=dist_sel, dist_t=dist_t)
