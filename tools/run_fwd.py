"""Runs the forward phase a few times (for rocprofv3 --pmc sessions)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from dqn_zoo_amd import _lib, learner as learner_lib, networks

A, B = 6, 32
sup = np.linspace(-10, 10, 51).astype(np.float32)
ln = learner_lib.RainbowLearner(networks.RainbowNetwork(A, sup), learner_lib.AdamConfig(), B)
g = torch.Generator(device='cuda'); g.manual_seed(0)
dev = (torch.randint(0, 256, (B, 84, 84, 4), dtype=torch.uint8, device='cuda', generator=g),
       torch.randint(0, A, (B,), device='cuda', generator=g),
       torch.randn(B, dtype=torch.float64, device='cuda', generator=g),
       torch.full((B,), 0.97, dtype=torch.float64, device='cuda'),
       torch.randint(0, 256, (B, 84, 84, 4), dtype=torch.uint8, device='cuda', generator=g),
       torch.rand(B, dtype=torch.float32, device='cuda', generator=g))
ln.resample_noise()
lib = _lib.load()
phases = int(sys.argv[1]) if len(sys.argv) > 1 else _lib.PHASE_ALL
for _ in range(12):
  ln.step(*dev, phases=phases, resample_noise=False)
torch.cuda.synchronize()
print('done')
