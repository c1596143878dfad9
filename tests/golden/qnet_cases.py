"""Frozen test cases for the Q-loss / update half.

TEST INFRASTRUCTURE.  The arithmetic of this half lives in rlax / optax / haiku /
jax, none of which can run here, so the reference itself cannot produce golden
numbers ("parity unpinned").  What this module freezes instead is
the float64 output of the CPU oracle on seeded inputs -- committed as
tests/golden/qnet_*.npz by gen_qnet_golden.py -- so that
  * an edit to oracle/qnet_oracle.py cannot silently move the target
    (tests/test_qnet_golden.py recomputes and compares at 1e-12),
  * the same numbers are reproduced by an INDEPENDENT torch-autograd float64
    model written with different primitives (tests/torch_models.py),
  * the HIP learners are compared with numbers that are files, not code.

One case = one learner step of one agent: per-sample losses (or TD errors),
the scalar loss, every gradient tensor, and the parameters / optimiser state
after one optimiser step with the reference's hyper-parameters.
"""

import numpy as np

from oracle import qnet_oracle as qo

A = 3            # actions (small: the fixtures hold sampled gradient entries)
B = 4            # batch
K = 51
NQ = 201
SUPPORT = np.linspace(-10.0, 10.0, K)
QUANTILES = (np.arange(NQ) + 0.5) / NQ
IQN_TAUS = (5, 6, 7)

# name -> (network kind, seed, hyper-parameters of the reference's run_atari.py)
CASES = {
    'rainbow': dict(net='rainbow', seed=101, opt='adam', lr=0.00025 / 4,
                    eps=0.005 / 32, max_norm=10.0),       # rainbow/run_atari.py:77-81
    'dqn': dict(net='dqn', seed=102, opt='rmsprop', lr=0.00025, decay=0.95,
                eps=0.01 / 32 ** 2, bound=1.0 / 32),      # dqn/run_atari.py:78-83
    'double_q': dict(net='double_dqn', seed=103, opt='rmsprop', lr=0.00025,
                     decay=0.95, eps=0.01 / 32 ** 2, bound=1.0 / 32),
    'prioritized': dict(net='double_dqn', seed=104, opt='rmsprop',
                        lr=0.00025 / 4, decay=0.95,
                        eps=(0.01 / 32 ** 2) * (1.0 / 4) ** 2, bound=1.0 / 32),
    'c51': dict(net='c51', seed=105, opt='adam', lr=0.00025, eps=0.01 / 32,
                max_norm=10.0),                           # c51/run_atari.py
    'qr': dict(net='qr', seed=106, opt='adam', lr=0.00005, eps=0.01 / 32,
               max_norm=10.0, kappa=1.0),                 # qrdqn/run_atari.py
    'iqn': dict(net='iqn', seed=107, opt='adam', lr=0.00005, eps=0.01 / 32,
                max_norm=0.0, kappa=1.0),                 # iqn/run_atari.py (no clip)
}


def make_inputs(name, dt=np.float64):
  """Seeded parameters, batch, weights and noise / taus of one case."""
  c = CASES[name]
  rs = np.random.RandomState(c['seed'])
  online = qo.init_params(c['net'], A, rs, dt, num_atoms=K, num_quantiles=NQ)
  target = qo.init_params(c['net'], A, rs, dt, num_atoms=K, num_quantiles=NQ)
  if c['net'] == 'rainbow':   # make sigma matter
    for k in online:
      if 'sigma' in k:
        online[k] = (online[k] * 5).astype(dt)
  s_tm1 = rs.randint(0, 256, (B, 84, 84, 4)).astype(np.uint8)
  s_t = rs.randint(0, 256, (B, 84, 84, 4)).astype(np.uint8)
  a = rs.randint(A, size=B).astype(np.int64)
  r = rs.choice([-1.0, 0.0, 1.0], size=B) * 2.5   # some |td| beyond the clip
  d = rs.choice([0.0, 0.99], size=B)
  d[0], d[1] = 0.0, 0.99
  inp = dict(case=c, online=online, target=target, batch=(s_tm1, a, r, d, s_t))
  inp['weights'] = rs.uniform(0.2, 1.0, size=B) if name in (
      'rainbow', 'prioritized') else None
  if name == 'rainbow':
    inp['noises'] = [qo.sample_noise(rs, A, dt, K) for _ in range(3)]
  if name == 'iqn':
    inp['taus'] = [rs.uniform(size=(B, n)).astype(dt) for n in IQN_TAUS]
  return inp


def oracle_step(name, inp, dt=np.float64):
  """losses, loss, gradients and one optimiser step from the CPU oracle."""
  c, on, tg, batch, w = (inp['case'], inp['online'], inp['target'], inp['batch'],
                         inp['weights'])
  if name == 'rainbow':
    loss, losses, grads, _ = qo.rainbow_loss_and_grads(
        on, tg, batch, w, inp['noises'], SUPPORT.astype(dt), A, dt)
  elif name in ('dqn', 'double_q', 'prioritized'):
    loss, losses, grads, _ = qo.dqn_family_loss_and_grads(
        name, on, tg, batch, w, c['bound'], dt)
  elif name == 'c51':
    loss, losses, grads, _ = qo.c51_loss_and_grads(on, tg, batch,
                                                   SUPPORT.astype(dt), A, dt)
  elif name == 'qr':
    loss, losses, grads, _ = qo.qr_loss_and_grads(
        on, tg, batch, QUANTILES.astype(dt), A, c['kappa'], dt)
  elif name == 'iqn':
    loss, losses, grads, _ = qo.iqn_loss_and_grads(on, tg, batch, inp['taus'],
                                                   c['kappa'], dt)
  else:
    raise KeyError(name)
  new_p, state, gnorm = optimizer_step(c, on, grads)
  return dict(loss=loss, losses=losses, grads=grads, params=new_p, opt=state,
              gnorm=gnorm)


def optimizer_step(c, params, grads):
  """One step of the case's optimiser from a zero state (optax 0.1.2 semantics,
  SURVEY.md Appendix A)."""
  if c['opt'] == 'adam':
    if c['max_norm'] > 0:
      clipped, gnorm = qo.clip_by_global_norm(grads, c['max_norm'])
    else:
      clipped, gnorm = grads, qo.global_norm(grads)
    p, st = qo.adam_update(params, clipped, qo.adam_init(params), c['lr'], c['eps'])
    return p, dict(m=st['mu'], v=st['nu']), gnorm
  p, st = qo.rmsprop_centered_update(params, grads, qo.rmsprop_init(params),
                                     c['lr'], c['decay'], c['eps'])
  return p, dict(m=st['mu'], v=st['nu']), qo.global_norm(grads)


SAMPLE = 512


def sample_tensor(x):
  """<= SAMPLE entries of a tensor at a fixed stride (whole tensor if small):
  what the fixtures store per gradient / parameter tensor."""
  f = np.asarray(x).reshape(-1)
  if f.size <= SAMPLE:
    return f.copy()
  return f[::max(1, f.size // SAMPLE)][:SAMPLE].copy()


def pack(result):
  """oracle_step() output -> flat dict of float64 arrays (npz)."""
  out = {'loss': np.float64(result['loss']),
         'losses': np.asarray(result['losses'], np.float64),
         'gnorm': np.float64(result['gnorm'])}
  for k in sorted(result['grads']):
    g = np.asarray(result['grads'][k], np.float64)
    out['g/%s' % k] = sample_tensor(g)
    out['gstat/%s' % k] = np.array([g.sum(), np.sqrt((g * g).sum()),
                                    np.abs(g).max()])
    out['p/%s' % k] = sample_tensor(result['params'][k]).astype(np.float64)
    out['m/%s' % k] = sample_tensor(result['opt']['m'][k]).astype(np.float64)
    out['v/%s' % k] = sample_tensor(result['opt']['v'][k]).astype(np.float64)
  return out


# ---- projection known-answer cases (rlax.categorical_l2_project) --------------
def projection_cases():
  rs = np.random.RandomState(7)
  z = SUPPORT
  cases = []
  for r, g in ((0.0, 1.0), (1.0, 0.99), (-1.0, 0.99 ** 3), (2.5, 0.0),
               (30.0, 0.99), (-30.0, 0.5), (0.3, 0.731)):
    p = rs.dirichlet(np.ones(K))
    cases.append((r + g * z, p))
  return cases
