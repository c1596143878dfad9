"""Runs 12 full learner steps of BASELINE config 2 (DQN, `dqn`) or 3 (double-Q + IS weights +
|td| priorities, `double_q`) for rocprofv3 --pmc sessions (tools/profile_round.sh)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from dqn_zoo_amd import learner as ll, networks

which = sys.argv[1] if len(sys.argv) > 1 else 'dqn'
A, B = 6, 32
if which == 'dqn':
  ln = ll.DenseLearner(networks.DenseNetwork('dqn', A), 'q',
                       ll.RmsPropConfig(learning_rate=0.00025, decay=0.95, eps=0.01 / 32 ** 2), B)
else:
  ln = ll.DenseLearner(networks.DenseNetwork('double_dqn', A), 'double_q',
                       ll.RmsPropConfig(learning_rate=0.00025 / 4, decay=0.95,
                                        eps=(0.01 / 32 ** 2) / 16), B)
ln.use_graphs = False
g = torch.Generator(device='cuda'); g.manual_seed(0)
dev = (torch.randint(0, 256, (B, 84, 84, 4), dtype=torch.uint8, device='cuda', generator=g),
       torch.randint(0, A, (B,), device='cuda', generator=g),
       torch.randn(B, dtype=torch.float64, device='cuda', generator=g),
       torch.full((B,), 0.97, dtype=torch.float64, device='cuda'),
       torch.randint(0, 256, (B, 84, 84, 4), dtype=torch.uint8, device='cuda', generator=g))
w = None if which == 'dqn' else torch.rand(B, dtype=torch.float32, device='cuda', generator=g)
for _ in range(int(sys.argv[2]) if len(sys.argv) > 2 else 12):
  ln.step(*dev, w)
torch.cuda.synchronize()
print('done')
