"""Feasibility probe (round 5): how much do the HBM-bound optimiser launch and the MFMA / latency-bound
forward launches of ANOTHER learner overlap when they run concurrently (two streams, no dependencies)?
Prints us per iteration for: optimiser alone, forward alone, both concurrently."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from dqn_zoo_amd import _lib, learner as ll, networks

sup = np.linspace(-10, 10, 51).astype(np.float32)
def mk(seed):
  ln = ll.RainbowLearner(networks.RainbowNetwork(6, sup), ll.AdamConfig(), 32, seed=seed)
  ln.use_graphs = False
  return ln
X, Y = mk(1), mk(2)
rs = np.random.RandomState(0)
s = torch.from_numpy(rs.randint(0, 256, (2, 32, 84, 84, 4)).astype(np.uint8)).cuda()
a = torch.from_numpy(rs.randint(0, 6, 32).astype(np.int64)).cuda()
r = torch.from_numpy(rs.uniform(-1, 1, 32)).cuda()
d = torch.full((32,), 0.97, dtype=torch.float64, device='cuda')
w = torch.ones(32, dtype=torch.float32, device='cuda')
X.keep_all_grads = True
X.step(s[0], a, r, d, s[1], w)   # a full step so that gradients exist
Y.step(s[0], a, r, d, s[1], w)
torch.cuda.synchronize()
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
which = sys.argv[1] if len(sys.argv) > 1 else 'nets'
fwd_phase = _lib.PHASE_FWD_NETS
def opt():
  with torch.cuda.stream(sa):
    X.step(s[0], a, r, d, s[1], w, phases=_lib.PHASE_OPTIMIZER)
def fwd():
  with torch.cuda.stream(sb):
    Y.step(s[0], a, r, d, s[1], w, phases=fwd_phase, resample_noise=False)
def run(fns, n=300):
  for f in fns: f()
  torch.cuda.synchronize()
  t0 = time.perf_counter()
  for _ in range(n):
    for f in fns: f()
  torch.cuda.synchronize()
  return 1e6 * (time.perf_counter() - t0) / n
print('optimiser alone      %.1f us' % run([opt]))
print('forward nets alone   %.1f us' % run([fwd]))
print('both, two streams    %.1f us' % run([opt, fwd]))
print('both, two streams    %.1f us' % run([fwd, opt]))
