import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from dqn_zoo_amd import _lib, learner as ll, networks
A, B = 4, int(sys.argv[1]) if len(sys.argv) > 1 else 10
sup = np.linspace(-10, 10, 51).astype(np.float32)
ln = ll.RainbowLearner(networks.RainbowNetwork(A, sup, 0.1), ll.AdamConfig(), B)
g = torch.Generator(device='cuda'); g.manual_seed(0)
dev = (torch.randint(0, 256, (B, 84, 84, 4), dtype=torch.uint8, device='cuda', generator=g),
       torch.randint(0, A, (B,), device='cuda', generator=g),
       torch.randn(B, dtype=torch.float64, device='cuda', generator=g),
       torch.full((B,), 0.97, dtype=torch.float64, device='cuda'),
       torch.randint(0, 256, (B, 84, 84, 4), dtype=torch.uint8, device='cuda', generator=g),
       torch.rand(B, dtype=torch.float32, device='cuda', generator=g))
lib = _lib.load()
for key, val in [(None, None)] + [tuple(int(x) for x in a.split('=')) for a in sys.argv[2:]]:
  if key is not None:
    lib.dz_set_tuning(key, val)
for ph, name in ((1, 'fwd'), (2, 'bwd'), (4, 'opt'), (7, 'all')):
  ln.step(*dev, phases=ph)
  torch.cuda.synchronize()
  print(name, 'ok', flush=True)
