"""Per-kernel event timings of the Rainbow step restricted to phases (warm-cache
experiment): python tools/fwd_only_prof.py [phases bitmask, default 1 = forward]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from dqn_zoo_amd import _lib, learner as ll, networks
import bench
A, B = 6, 32
phases = int(sys.argv[1]) if len(sys.argv) > 1 else _lib.PHASE_FORWARD
sup = np.linspace(-10, 10, 51).astype(np.float32)
ln = ll.RainbowLearner(networks.RainbowNetwork(A, sup), ll.AdamConfig(), B)
ln.use_graphs = False   # per-kernel event marks need eager launches
g = torch.Generator(device='cuda'); g.manual_seed(0)
dev = (torch.randint(0, 256, (B, 84, 84, 4), dtype=torch.uint8, device='cuda', generator=g),
       torch.randint(0, A, (B,), device='cuda', generator=g),
       torch.randn(B, dtype=torch.float64, device='cuda', generator=g),
       torch.full((B,), 0.97, dtype=torch.float64, device='cuda'),
       torch.randint(0, 256, (B, 84, 84, 4), dtype=torch.uint8, device='cuda', generator=g),
       torch.rand(B, dtype=torch.float32, device='cuda', generator=g))
step = lambda: ln.step(*dev, phases=phases)
for _ in range(20): step()
avg = bench.profile_kernels(step, 50)
for k, v in sorted(avg.items(), key=lambda kv: -kv[1]):
  print('%-22s %7.2f us' % (k, v * 1e6))
print('sum %.1f us' % (sum(avg.values()) * 1e6))
