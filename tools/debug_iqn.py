import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import qnet_oracle as qo
from tests.test_iqn_gpu import _make, _batch, _dev, _f64, A
from dqn_zoo_amd import _lib
b, samples = 32, (64, 64, 64)
rs, online, target, ln, taus = _make(b, samples, 40 + b)
batch = _batch(rs, b, scale_r=2.0)
ln.step(*_dev(batch), taus=_dev(taus), phases=3)
torch.cuda.synchronize()
dist, _, cache = qo.iqn_fwd(online, batch[0], taus[0])
d64, _, c64 = qo.iqn_fwd(_f64(online), batch[0], taus[0].astype(np.float64), np.float64)
M0 = b * 64
h1 = ln.ws_view('h1', M0 * 512).cpu().numpy().reshape(M0, 512)
print('h1 err dev', np.abs(h1 - c64['h']).max(), 'orc', np.abs(cache['h'] - c64['h']).max())
print('mask flips dev', ((h1 > 0) != (c64['z1'] > 0)).sum(), 'orc', ((cache['z1'] > 0) != (c64['z1'] > 0)).sum())
hin = ln.ws_view('hin', M0 * 3136).cpu().numpy().reshape(M0, 3136)
print('hin err dev', np.abs(hin - c64['hin']).max(), 'orc', np.abs(cache['hin'] - c64['hin']).max(), 'max', np.abs(c64['hin']).max())
cos = ln.ws_view('cos', M0 * 64).cpu().numpy().reshape(M0, 64)
print('cos err dev', np.abs(cos - c64['cosemb']).max(), 'orc', np.abs(cache['cosemb'] - c64['cosemb']).max())
fl = np.argwhere((h1 > 0) != (c64['z1'] > 0))
for r, c in fl[:5]:
  print(r, c, h1[r, c], cache['z1'][r, c], c64['z1'][r, c])
