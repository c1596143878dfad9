"""GPU parity of the Rainbow learner step (through dz_rainbow_learn) against the
NumPy oracle: per-layer forward, selector/target distributions, per-sample
losses (<= 1e-5 relative, the BASELINE.json tolerance), every gradient tensor,
global norm, and the clip+Adam update.  Noise is injected into both sides."""

import numpy as np
import pytest
import torch

from dqn_zoo_amd import _lib
from oracle import qnet_oracle as qo

pytestmark = pytest.mark.gpu

K = 51
SUPPORT = np.linspace(-10.0, 10.0, K).astype(np.float32)


def _problem(A, B, seed, sigma_scale=3.0):
  rs = np.random.RandomState(seed)
  online = qo.init_params('rainbow', A, rs)
  target = qo.init_params('rainbow', A, rs)
  for p in (online, target):
    for k in p:
      if 'sigma' in k:
        p[k] = (p[k] * sigma_scale).astype(np.float32)
  s_tm1 = rs.randint(0, 256, (B, 84, 84, 4)).astype(np.uint8)
  s_t = rs.randint(0, 256, (B, 84, 84, 4)).astype(np.uint8)
  a = rs.randint(A, size=B).astype(np.int64)
  r = rs.choice([-1.0, 0.0, 1.0], size=B) * rs.uniform(0.5, 1.5, size=B)
  d = rs.choice([0.0, 0.99 ** 3], size=B)
  w = rs.uniform(0.1, 1.0, size=B).astype(np.float32)
  noises = [qo.sample_noise(rs, A) for _ in range(3)]
  return online, target, (s_tm1, a, r, d, s_t), w, noises


def _learner(A, B, online, target, noises, max_norm=10.0):
  from dqn_zoo_amd import learner as learner_lib
  from dqn_zoo_amd import networks
  net = networks.RainbowNetwork(A, SUPPORT)
  ln = learner_lib.RainbowLearner(
      net, learner_lib.AdamConfig(max_global_grad_norm=max_norm), B,
      params=online)
  ln.set_params(target, 'target')
  ln.set_noise(noises)
  return ln


def _dev_batch(batch, w):
  s_tm1, a, r, d, s_t = batch
  return (torch.from_numpy(s_tm1).cuda(), torch.from_numpy(a).cuda(),
          torch.from_numpy(r).cuda(), torch.from_numpy(d).cuda(),
          torch.from_numpy(s_t).cuda(), torch.from_numpy(w).cuda())


def _rel(a, b):
  return np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)


@pytest.mark.parametrize('A,B,seed', [(6, 32, 0), (18, 32, 1), (3, 10, 2), (4, 48, 3)])
def test_forward_loss_backward_vs_oracle(A, B, seed):
  online, target, batch, w, noises = _problem(A, B, seed)
  ln = _learner(A, B, online, target, noises)
  from dqn_zoo_amd import _lib
  ln.step(*_dev_batch(batch, w), phases=_lib.PHASE_FORWARD | _lib.PHASE_BACKWARD,
          resample_noise=False)
  torch.cuda.synchronize()
  L = ln.layout
  s_tm1, a, r, d, s_t = batch

  # ---- per-layer forward, all three applies ----
  applies = [(online, s_tm1, noises[0]), (online, s_t, noises[1]),
             (target, s_t, noises[2])]
  act1 = ln.ws_view('act1', 3 * B * 400 * 32).cpu().numpy().reshape(3, B, 20, 20, 32)
  act2 = ln.ws_view('act2', 3 * B * 81 * 64).cpu().numpy().reshape(3, B, 9, 9, 64)
  feat = ln.ws_view('feat', 3 * B * 3136).cpu().numpy().reshape(3, B, 3136)
  h1 = ln.ws_view('h1', 3 * B * 1024).cpu().numpy().reshape(3, B, 1024)
  out2 = ln.ws_view('fc2_out', 3 * B * L.ld2).cpu().numpy().reshape(3, B, L.ld2)
  for g, (p, x, nz) in enumerate(applies):
    logits, q, cache = qo.rainbow_fwd(p, x, nz, SUPPORT, A)
    tc = cache['torso']
    assert _rel(act1[g], np.maximum(tc['conv1'][2], 0)) < 2e-6
    assert _rel(act2[g], np.maximum(tc['conv2'][2], 0)) < 3e-6
    assert _rel(feat[g], cache['feat']) < 5e-6
    assert _rel(h1[g][:, :512], cache['ha']) < 1e-5
    assert _rel(h1[g][:, 512:], cache['hv']) < 1e-5
    adv = out2[g][:, :A * K].reshape(B, A, K)
    val = out2[g][:, L.val_off:L.val_off + K].reshape(B, 1, K)
    glogits = val + adv - adv.mean(axis=1, keepdims=True)
    assert _rel(glogits, logits) < 2e-5
    if g == 1:
      q_sel = ln.ws_view('q_sel', B * A).cpu().numpy().reshape(B, A)
      np.testing.assert_allclose(q_sel, q, rtol=2e-4, atol=2e-5)

  # ---- losses, targets, gradients ----
  loss, losses, grads, aux = qo.rainbow_loss_and_grads(
      online, target, batch, w, noises, SUPPORT, A)
  _, _, tgt = qo.categorical_double_q_losses(
      SUPPORT, aux['logits_tm1'], a, np.asarray(r, np.float32),
      np.asarray(d, np.float32), aux['logits_target'], aux['q_t'])
  tp = ln.ws_view('target_probs', B * K).cpu().numpy().reshape(B, K)
  np.testing.assert_allclose(tp, tgt, rtol=1e-4, atol=2e-7)
  np.testing.assert_allclose(ln.losses.cpu().numpy(), losses, rtol=1e-5)
  np.testing.assert_allclose(ln.priorities.cpu().numpy(),
                             np.clip(np.abs(losses), 0, 100), rtol=1e-5)
  dout2 = ln.ws_view('dout2', B * L.ld2).cpu().numpy().reshape(B, L.ld2)
  dl = aux['dlogits']
  assert _rel(dout2[:, L.val_off:L.val_off + K], dl.sum(axis=1)) < 2e-5
  g_dev = L.unpack(ln.grad.cpu().numpy())
  assert set(g_dev) == set(grads)
  # float64 evaluation of the same step = ground truth for the gradients.  A
  # device tensor passes if it is within 1e-4 of it (relative to the tensor's
  # max-abs) or at least as accurate as the float32 oracle itself (long
  # cancelling sums such as conv1/b = sum over 12800 pixels are
  # summation-order limited in float32 on either side).
  f64 = lambda t: {k: v.astype(np.float64) for k, v in t.items()}
  _, _, g64, _ = qo.rainbow_loss_and_grads(
      f64(online), f64(target), batch, w, [f64(n) for n in noises],
      SUPPORT.astype(np.float64), A, np.float64)
  for k in sorted(grads):
    scale = np.abs(g64[k]).max()
    e_dev = np.abs(g_dev[k] - g64[k]).max() / scale
    e_orc = np.abs(grads[k] - g64[k]).max() / scale
    assert e_dev < max(1e-4, 4 * e_orc), (k, e_dev, e_orc)


@pytest.mark.parametrize('max_norm', [10.0, 1e-3, 0.0])
def test_clip_and_adam_vs_oracle(max_norm):
  A, B = 6, 32
  online, target, batch, w, noises = _problem(A, B, 5)
  ln = _learner(A, B, online, target, noises, max_norm=max_norm)
  ln.keep_all_grads = True   # this test reads every gradient block after a full step
  dev = _dev_batch(batch, w)
  p = {k: v.copy() for k, v in online.items()}
  st = qo.adam_init(p)
  for it in range(3):
    ln.step(*dev, resample_noise=False)
    torch.cuda.synchronize()
    # oracle optimizer fed with the DEVICE gradients: isolates clip+Adam
    g_dev = ln.layout.unpack(ln.grad.cpu().numpy())
    if max_norm > 0:
      clipped, gn = qo.clip_by_global_norm(g_dev, max_norm)
    else:
      clipped, gn = g_dev, qo.global_norm(g_dev)
    p, st = qo.adam_update(p, clipped, st, ln.opt.learning_rate, ln.opt.eps)
    sc = ln.scalars()
    np.testing.assert_allclose(sc['gnorm'], gn, rtol=2e-5)
    assert sc['unclipped'] == (max_norm <= 0 or gn < max_norm)
    assert int(ln.adam_count.item()) == it + 1
    np.testing.assert_allclose(sc['bc1'], 1 - 0.9 ** (it + 1), rtol=1e-5)
    p_dev = ln.get_params()
    m_dev = ln.layout.unpack(ln.adam_m.cpu().numpy())
    v_dev = ln.layout.unpack(ln.adam_v.cpu().numpy())
    for k in p:
      np.testing.assert_allclose(m_dev[k], st['mu'][k], rtol=1e-5, atol=1e-12)
      np.testing.assert_allclose(v_dev[k], st['nu'][k], rtol=1e-5, atol=1e-20)
      assert np.abs(p_dev[k] - p[k]).max() <= 2e-3 * ln.opt.learning_rate + 1e-9, k
    # keep the oracle's parameters locked to the device's for the next step
    p = p_dev
    st = dict(count=st['count'], mu=m_dev, nu=v_dev)


@pytest.mark.parametrize('keep_all', [False, True])
def test_overflowed_global_norm_zeroes_the_update(keep_all):
  """optax's clip_by_global_norm computes (g / g_norm) * max_norm; with a
  float32 norm that overflowed to +inf every finite g becomes 0, Adam's moments decay
  from zero and the parameters do not move.  The reciprocal form of the division must
  not turn that into 0 * inf = NaN."""
  A, B = 6, 32
  online, target, batch, w, noises = _problem(A, B, 21)
  ln = _learner(A, B, online, target, noises)
  ln.keep_all_grads = keep_all
  before = ln.online.clone()
  ln.step(*_dev_batch(batch, (w * 1e25).astype(np.float32)), resample_noise=False)
  torch.cuda.synchronize()
  assert ln.scalars()['gnorm'] == np.inf
  assert bool(torch.isfinite(ln.online).all())
  assert torch.equal(ln.online, before)
  assert float(ln.adam_m.abs().max()) == 0.0 and float(ln.adam_v.abs().max()) == 0.0


@pytest.mark.parametrize('A,B,seed', [(6, 32, 31), (4, 20, 32)])
def test_on_the_fly_fc1_gradient_elements_and_norm_share(A, B, seed):
  """The DEFAULT step never stores fc1's two weight-gradient blocks: the optimiser forms
  each element from the layer's input and output gradient (dz_fc1_onfly.h) and the blocks'
  share of the global norm comes from Gram matrices on the f64 matrix pipe (GramDSide).
  Both are read back directly here: with the clip out of reach the
  first Adam step leaves m = (1 - b1) g and v = (1 - b2) g^2, element by element, and
  gnorm^2 minus the stored blocks' sums of squares is the fc1 share.  Ground truth: the
  float64 oracle."""
  online, target, batch, w, noises = _problem(A, B, seed)
  ln = _learner(A, B, online, target, noises, max_norm=1e9)
  assert not ln.keep_all_grads
  ln.step(*_dev_batch(batch, w), resample_noise=False)
  torch.cuda.synchronize()
  f64 = lambda t: {k: v.astype(np.float64) for k, v in t.items()}
  _, _, g64, _ = qo.rainbow_loss_and_grads(
      f64(online), f64(target), batch, w, [f64(n) for n in noises],
      SUPPORT.astype(np.float64), A, np.float64)
  m = ln.layout.unpack(ln.adam_m.cpu().numpy())
  v = ln.layout.unpack(ln.adam_v.cpu().numpy())
  fc1 = [k for k in g64 if k.split('/')[0] in ('adv1', 'val1') and k.endswith('/w')]
  assert len(fc1) == 4
  for k in fc1:
    g_dev = m[k].astype(np.float64) / (1.0 - ln.opt.b1)
    scale = np.abs(g64[k]).max()
    assert np.abs(g_dev - g64[k]).max() / scale < 2e-6, k       # float32 sums of <= 32 terms
    g2 = v[k].astype(np.float64) / (1.0 - ln.opt.b2)
    np.testing.assert_allclose(g2, g64[k] ** 2, rtol=1e-5, atol=5e-6 * scale ** 2)
  # the norm: whole vector, then the on-the-fly blocks' share alone
  sc = ln.scalars()
  gn64 = np.sqrt(sum(float((g ** 2).sum()) for g in g64.values()))
  np.testing.assert_allclose(sc['gnorm'], gn64, rtol=1e-5)
  stored = ln.layout.unpack(ln.grad.cpu().numpy())
  other = sum(float((stored[k].astype(np.float64) ** 2).sum()) for k in g64 if k not in fc1)
  share_dev = float(sc['gnorm']) ** 2 - other
  share64 = sum(float((g64[k] ** 2).sum()) for k in fc1)
  assert share64 > 0.05 * gn64 ** 2        # (a share that matters, so the difference is meaningful)
  np.testing.assert_allclose(share_dev, share64, rtol=5e-5)


def test_full_step_vs_oracle_update():
  """End-to-end: oracle rainbow_update vs device step, same inputs."""
  A, B = 6, 32
  online, target, batch, w, noises = _problem(A, B, 9)
  ln = _learner(A, B, online, target, noises)
  ln.step(*_dev_batch(batch, w), resample_noise=False)
  torch.cuda.synchronize()
  newp, st, out = qo.rainbow_update(online, target, qo.adam_init(online), batch,
                                    w, noises, SUPPORT, A)
  sc = ln.scalars()
  np.testing.assert_allclose(sc['loss'], out['loss'], rtol=1e-5)
  np.testing.assert_allclose(sc['gnorm'], out['gnorm'], rtol=1e-4)
  np.testing.assert_allclose(ln.losses.cpu().numpy(), out['losses'], rtol=1e-5)
  p_dev = ln.get_params()
  lr = ln.opt.learning_rate
  for k in newp:
    diff = np.abs(p_dev[k] - newp[k])
    # Adam's first step moves every weight by lr*g/(|g|+eps): a flipped sign
    # would be a 2*lr discrepancy, a missed update 1*lr.  Both excluded by 20x
    # (the 10-step trajectory is in tests/test_trajectory_gpu.py).
    assert diff.max() <= 0.05 * lr, (k, diff.max() / lr)
  # target untouched, then synced
  t_dev = ln.get_params('target')
  for k in target:
    np.testing.assert_array_equal(t_dev[k], target[k])
  ln.sync_target()
  t_dev = ln.get_params('target')
  for k in target:
    np.testing.assert_array_equal(t_dev[k], p_dev[k])


@pytest.mark.parametrize('max_norm', [10.0, 1e-3])
def test_derived_sigma_gradient_is_bit_identical(max_norm):
  """A step split into two calls (forward | backward + optimiser) stores fc1's mu-weight
  gradient and never its sigma-weight gradient: the optimiser derives it from the mu
  gradient and the noise.  == the same split step with every gradient block
  materialised: parameters and both Adam moments bit for bit."""
  from dqn_zoo_amd import _lib
  A, B = 6, 32
  online, target, batch, w, noises = _problem(A, B, 13)
  dev = _dev_batch(batch, w)
  lns = []
  for keep in (False, True):
    ln = _learner(A, B, online, target, noises, max_norm=max_norm)
    ln.keep_all_grads = keep
    for _ in range(3):
      ln.step(*dev, resample_noise=False, phases=_lib.PHASE_FORWARD)
      ln.step(*dev, resample_noise=False, phases=_lib.PHASE_BACKWARD | _lib.PHASE_OPTIMIZER)
    torch.cuda.synchronize()
    lns.append(ln)
  a, b = lns
  assert torch.equal(a.online, b.online)
  assert torch.equal(a.adam_m, b.adam_m) and torch.equal(a.adam_v, b.adam_v)
  assert a.scalars()['gnorm'] == b.scalars()['gnorm']
  ga = a.layout.unpack(a.grad.cpu().numpy())
  gb = b.layout.unpack(b.grad.cpu().numpy())
  for k in ('conv1/w', 'conv3/b', 'adv1/mu/b', 'val1/sigma/b', 'adv2/mu/w', 'val2/sigma/w'):
    np.testing.assert_array_equal(ga[k], gb[k])
  assert np.abs(gb['adv1/mu/w']).max() > 0 and np.abs(gb['adv1/sigma/w']).max() > 0


@pytest.mark.parametrize('max_norm,B,A', [(10.0, 32, 6), (1e-3, 32, 6), (0.05, 7, 6),
                                          (10.0, 32, 18), (10.0, 20, 3), (10.0, 32, 11),
                                          (10.0, 1, 6), (10.0, 2, 4)])
def test_fc1_gradient_formed_in_the_optimiser_matches_the_stored_gradient(max_norm, B, A):
  """The one-call step never stores fc1's weight gradient (dz_fc1_onfly.h: every
  element is formed from the L2-resident factors inside the optimiser, the layer's share
  of the global norm comes from two Gram matrices on the f64 matrix pipe).  Against the
  step with every gradient block materialised: the same global norm to float32 rounding
  -- also under a binding clip, where the norm scales every update -- and the same
  parameters and moments up to the rounding order of the 32-term sums (one-call: batch
  index ascending in one FMA chain; stored: MFMA partial sums).  The one-call step's
  input gradients are the row-owning streams of dz_row_dgrad.h: 3 / 6 / 11 / 18 actions
  = 1 / 2 / 3 / 4 column chunks of the advantage head (18: five jobs on four waves)."""
  online, target, batch, w, noises = _problem(A, B, 13)
  dev = _dev_batch(batch, w)
  lns, norms, first = [], [], []
  for keep in (False, True):
    ln = _learner(A, B, online, target, noises, max_norm=max_norm)
    ln.keep_all_grads = keep
    gn = []
    for it in range(4):
      ln.step(*dev, resample_noise=False)
      gn.append(ln.scalars()['gnorm'])
      if it == 0:
        first.append([t.cpu().numpy().copy() for t in (ln.online, ln.adam_m, ln.adam_v)])
    torch.cuda.synchronize()
    lns.append(ln)
    norms.append(gn)
  a, b = lns
  lr = a.opt.learning_rate
  # one step: only the rounding order of the 32-term sums separates the two forms
  assert abs(norms[0][0] - norms[1][0]) <= 2e-6 * norms[1][0]
  (pa, ma, va), (pb, mb, vb) = first
  assert np.abs(pa - pb).max() <= 2e-3 * lr, np.abs(pa - pb).max() / lr
  # (a float32 sum in another order: absolute error ~ 32 eps x the largest term)
  np.testing.assert_allclose(ma, mb, rtol=2e-4, atol=3e-6 * np.abs(mb).max())
  np.testing.assert_allclose(va, vb, rtol=4e-4, atol=3e-6 * np.abs(vb).max())
  # further steps: Adam's first updates are ~lr * sign(g), so an entry whose gradient is
  # a near-cancelling sum can move the other way in the two forms; the trajectories stay
  # close but no longer to rounding
  np.testing.assert_allclose(norms[0], norms[1], rtol=2e-3)
  assert a.scalars()['unclipped'] == b.scalars()['unclipped']
  assert np.abs(a.online.cpu().numpy() - b.online.cpu().numpy()).max() <= 8.5 * lr
  # the fc1 matrices did move
  pa = a.get_params('online')
  assert np.abs(pa['adv1/mu/w'] - online['adv1/mu/w']).max() > 0
  assert np.abs(pa['val1/sigma/w'] - online['val1/sigma/w']).max() > 0


def test_device_noise_distribution():
  A, B = 6, 32
  online, target, batch, w, noises = _problem(A, B, 3)
  ln = _learner(A, B, online, target, noises)
  ln.resample_noise()
  x = ln.noise.cpu().numpy()
  ln.resample_noise()
  y = ln.noise.cpu().numpy()
  assert not np.array_equal(x, y)
  # f(n) = sign(n) sqrt|n|, n ~ N(0,1) truncated to [-2,2]
  assert np.abs(x).max() <= np.sqrt(2.0) + 1e-6
  n = np.sign(x) * x * x
  assert abs(n.mean()) < 0.02 and abs(n.std() - 0.8796) < 0.02
  assert abs((np.abs(n) < 1).mean() - 0.6827 / 0.9545) < 0.01
