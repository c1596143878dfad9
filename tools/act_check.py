"""The one-launch decision (batch 1) against the oracle and against the multi-launch apply,
plus its latency: back-to-back (events) and per decision with a host wait in between."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from dqn_zoo_amd import learner as ll, networks
from oracle import qnet_oracle as qo
A = int(os.environ.get('A', 6))
sup = np.linspace(-10, 10, 51).astype(np.float32)
rs = np.random.RandomState(3)
params = qo.init_params('rainbow', A, rs)
for k in params:
  if 'sigma' in k:
    params[k] = (params[k] * 3).astype(np.float32)
ln = ll.RainbowLearner(networks.RainbowNetwork(A, sup), ll.AdamConfig(), 32, params=params)
ln.act_graphs = False
worst = 0.0
for i in range(5):
  x = rs.randint(0, 256, (1, 84, 84, 4)).astype(np.uint8)
  xd = torch.from_numpy(x).cuda()
  q, g, v = ln.apply(xd)
  torch.cuda.synchronize()
  nz = ln.layout.unpack_noise(ln._act_noise.cpu().numpy())
  _, q_ref, _ = qo.rainbow_fwd(params, x, nz, sup, A)
  q5, g5, v5 = ln.apply(xd, noise=nz)     # stored-noise apply: the multi-launch path
  err = float(np.abs(q.cpu().numpy() - q_ref).max())
  err5 = float((q - q5).abs().max())
  worst = max(worst, err)
  print('decision', i, 'max|q - oracle| %.2e' % err, 'max|q - multi-launch| %.2e' % err5,
        'greedy', int(g[0]), int(q_ref[0].argmax()), 'step', ln.act_step(),
        'seam words', ln._act_ws[int(ln.network.layout(1).c.ws_act_seams):][:64 * 8:64]
        .view(torch.int32).tolist())
assert worst < 2e-4, worst
x = torch.randint(0, 256, (1, 84, 84, 4), dtype=torch.uint8, device='cuda')
for _ in range(2000):
  ln.apply(x)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(500):
  ln.apply(x)
e1.record(); torch.cuda.synchronize()
print('back to back: %.2f us per decision' % (e0.elapsed_time(e1) * 1e3 / 500))
t0 = time.perf_counter()
for _ in range(500):
  q, g, v = ln.apply(x)
  torch.cuda.synchronize()
print('with a host wait per decision: %.2f us' % ((time.perf_counter() - t0) * 1e6 / 500))
