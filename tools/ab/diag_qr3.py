import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import tests.test_dense_gpu as T
from oracle import qnet_oracle as qo
from dqn_zoo_amd import learner as ll, _lib
T.A = 3
opt = ll.AdamConfig(learning_rate=0.00025, eps=0.01 / 32, max_global_grad_norm=10.0)
rs, online, target, ln = T._make('qr', 'quantile', opt, 12, huber_param=1.0)
batch = T._batch(rs)
ln.step(*T._dev(batch), phases=_lib.PHASE_FORWARD | _lib.PHASE_BACKWARD)
torch.cuda.synchronize()
B = T.B
feat = ln.ws_view('feat', 2 * B * 3136).cpu().numpy().reshape(2, B, 3136)
act2 = ln.ws_view('act2', 2 * B * 81 * 64).cpu().numpy().reshape(2, B, -1)
act1 = ln.ws_view('act1', 2 * B * 400 * 32).cpu().numpy().reshape(2, B, -1)
for g, (p, x) in enumerate([(online, batch[0]), (target, batch[4])]):
  f32, c32 = qo.torso_fwd(p, x, np.float32)
  f64, c64 = qo.torso_fwd(T._f64(p), x, np.float64)
  for name, dev, o32, o64 in (('feat', feat[g], f32, f64),):
    flips_dev = ((dev > 0) != (o64 > 0)).sum()
    flips_32 = ((o32 > 0) != (o64 > 0)).sum()
    idx = np.argwhere((dev > 0) != (o64 > 0))
    print(g, name, 'flips dev', flips_dev, 'flips o32', flips_32, 'rel', np.abs(dev - o64).max() / np.abs(o64).max())
    for i in idx[:5]:
      print('   ', tuple(i), dev[tuple(i)], o32[tuple(i)], o64[tuple(i)])
  print(list(c32.keys()) if isinstance(c32, dict) else type(c32))
f = lambda o, t, dt: qo.qr_loss_and_grads(o, t, batch, T.QUANTILES.astype(dt), T.A, 1.0, dt)
l32, losses, g32, aux = f(online, target, np.float32)
_, _, g64, _ = f(T._f64(online), T._f64(target), np.float64)
gd = ln.layout.unpack(ln.grad.cpu().numpy())
for k in sorted(g64):
  sc = max(np.abs(g64[k]).max(), 1e-30)
  print(k, np.abs(gd[k] - g64[k]).max() / sc, np.abs(g32[k] - g64[k]).max() / sc)
out = {}
for name, n in (('act1', 2*B*400*32), ('act2', 2*B*81*64), ('feat', 2*B*3136), ('dfeat', B*3136), ('dact2', B*81*64), ('dact1', B*400*32)):
  try:
    out[name] = ln.ws_view(name, n).cpu().numpy()
  except Exception as e:
    print('no view', name, e)
np.savez('/tmp/diag_%s.npz' % sys.argv[1], **out)
if len(sys.argv) > 2:
  a = np.load('/tmp/diag_%s.npz' % sys.argv[2])
  for k in out:
    d = np.abs(out[k] - a[k])
    print('CMP', k, d.max() / max(np.abs(a[k]).max(), 1e-30), 'n_bad', int((d > 1e-5 * np.abs(a[k]).max()).sum()), 'first bad', np.argwhere(d > 1e-5 * np.abs(a[k]).max())[:3].ravel())
