"""200 acting applies (batch 1) for rocprofv3 --kernel-trace."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from dqn_zoo_amd import learner as ll, networks
sup = np.linspace(-10, 10, 51).astype(np.float32)
ln = ll.RainbowLearner(networks.RainbowNetwork(6, sup), ll.AdamConfig(), 32)
ln.act_graphs = False
x = torch.randint(0, 256, (1, 84, 84, 4), dtype=torch.uint8, device='cuda')
for _ in range(220):
  ln.apply(x)
torch.cuda.synchronize()
