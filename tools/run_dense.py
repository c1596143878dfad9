"""Runs 12 (or argv[2]) full learner steps of BASELINE config 2 (DQN, `dqn`), 3 (double-Q + IS
weights + |td| priorities, `double_q`), or the C51 / QR-DQN learners (`c51`, `qr`) for rocprofv3
sessions (tools/profile_round.sh).  argv[3] = `prof`: HIP-event duration of every launch and
the un-profiled steps per second instead."""
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
elif which == 'c51':   # c51/run_atari.py:200-216
  ln = ll.DenseLearner(networks.DenseNetwork('c51', A, support=np.linspace(-10, 10, 51)),
                       'categorical', ll.AdamConfig(learning_rate=0.00025, eps=0.01 / 32,
                                                    max_global_grad_norm=10.0), B)
elif which == 'qr':    # qrdqn/run_atari.py:196-214
  ln = ll.DenseLearner(networks.DenseNetwork('qr', A, quantiles=(np.arange(201) + 0.5) / 201),
                       'quantile', ll.AdamConfig(learning_rate=0.00005, eps=0.01 / 32,
                                                 max_global_grad_norm=10.0), B)
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
w = None if which != 'double_q' else torch.rand(B, dtype=torch.float32, device='cuda', generator=g)
if len(sys.argv) > 3 and sys.argv[3] == 'prof':
  import time
  sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
  import bench
  avg = bench.profile_kernels(lambda: ln.step(*dev, w), 30)
  for k, v in avg.items(): print('%-24s %7.2f us' % (k, v * 1e6))
  print('sum %.1f us' % (1e6 * sum(avg.values())))
  ln.use_graphs = None
  st = torch.cuda.Stream()
  with torch.cuda.stream(st):
    for _ in range(200): ln.step(*dev, w)
    for rep in range(3):
      torch.cuda.synchronize(); t0 = time.perf_counter()
      for _ in range(1000): ln.step(*dev, w)
      torch.cuda.synchronize(); print('%s us/step %.1f' % (which, 1e6 * (time.perf_counter() - t0) / 1000))
  sys.exit(0)
for _ in range(int(sys.argv[2]) if len(sys.argv) > 2 else 12):
  ln.step(*dev, w)
torch.cuda.synchronize()
print('done')
