"""Runs a few IQN learner steps (for rocprofv3 --pmc sessions)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from dqn_zoo_amd import _lib, learner as ll, networks
A, B = 18, 32
ln = ll.IqnLearner(networks.IqnNetwork(A, 64), ll.AdamConfig(learning_rate=5e-5, eps=0.01 / 32,
                                                              max_global_grad_norm=0.0), B)
rs = np.random.RandomState(0)
dev = [torch.from_numpy(x).cuda() for x in (
    rs.randint(0, 256, (B, 84, 84, 4)).astype(np.uint8), rs.randint(A, size=B).astype(np.int64),
    rs.choice([-1.0, 0.0, 1.0], size=B), rs.choice([0.0, 0.99], size=B),
    rs.randint(0, 256, (B, 84, 84, 4)).astype(np.uint8))]
for a in sys.argv[1:]:
  k, v = a.split('=')
  _lib.load().dz_set_tuning(int(k), int(v))
for _ in range(6):
  ln.step(*dev)
torch.cuda.synchronize()
print('done')
