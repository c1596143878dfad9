"""Correctness side of a library A/B (tools/ab.sh): one Rainbow learner step (B = 32, A = 6) on
fixed inputs; dumps the activations / gradients / parameters to an .npz, or compares two dumps.

    python tools/ab_dump.py dump out.npz
    python tools/ab_dump.py cmp base.npz new.npz      (max |a - b| / max |a| per tensor)
"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np


def dump(path):
  import torch
  from dqn_zoo_amd import _lib, learner as learner_lib, networks
  A, B = int(os.environ.get('AB_A', 6)), 32
  sup = np.linspace(-10, 10, 51).astype(np.float32)
  ln = learner_lib.RainbowLearner(networks.RainbowNetwork(A, sup), learner_lib.AdamConfig(), B, seed=3)
  g = torch.Generator(device='cuda'); g.manual_seed(0)
  dev = (torch.randint(0, 256, (B, 84, 84, 4), dtype=torch.uint8, device='cuda', generator=g),
         torch.randint(0, A, (B,), device='cuda', generator=g),
         torch.randn(B, dtype=torch.float64, device='cuda', generator=g),
         torch.full((B,), 0.97, dtype=torch.float64, device='cuda'),
         torch.randint(0, 256, (B, 84, 84, 4), dtype=torch.uint8, device='cuda', generator=g),
         torch.rand(B, dtype=torch.float32, device='cuda', generator=g))
  ln.resample_noise()
  out = {}
  for it in range(2):
    ln.step(*dev, resample_noise=False)
    torch.cuda.synchronize()
    for name, n in (('act1', 3 * B * 400 * 32), ('act2', 3 * B * 81 * 64), ('feat', 3 * B * 3136),
                    ('h1', 3 * B * 1024)):
      out['%s_%d' % (name, it)] = ln.ws_view(name, n).cpu().numpy()
    out['losses_%d' % it] = ln.losses.cpu().numpy()
    out['params_%d' % it] = ln.online.cpu().numpy()
    out['grad_%d' % it] = ln.grad.cpu().numpy()[:77984]   # the conv gradients (fc1's is never stored)
  np.savez(path, **out)
  print('dumped', path)


def cmp(a, b):
  x, y = np.load(a), np.load(b)
  worst = 0.0
  for k in x.files:
    d = float(np.max(np.abs(x[k].astype(np.float64) - y[k])) / max(np.max(np.abs(x[k])), 1e-30))
    nan = int(np.isnan(y[k]).sum())
    print('%-12s rel %.3e  nan %d' % (k, d, nan))
    worst = max(worst, d if not nan else 1.0)
  print('WORST', worst)


if __name__ == '__main__':
  if sys.argv[1] == 'dump':
    dump(sys.argv[2])
  else:
    cmp(sys.argv[2], sys.argv[3])
