"""The Rainbow learner step at another action count on a small store (for kernel traces):
    python tools/run_rainbow_actions.py <num_actions> <steps>"""
import os, sys, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
from dqn_zoo_amd import learner as learner_lib, networks

A, steps = int(sys.argv[1]), int(sys.argv[2])
args = types.SimpleNamespace(capacity=20000, batch=32)
dev = torch.device('cuda', 0)
replay, _, _ = bench.build_workload(args, dev, seed=3)
torch.cuda.set_stream(torch.cuda.Stream(dev))
sup = np.linspace(-bench.VMAX, bench.VMAX, bench.NUM_ATOMS).astype(np.float32)
ln = learner_lib.RainbowLearner(networks.RainbowNetwork(A, sup, 0.1), learner_lib.AdamConfig(), 32, seed=3, device=dev)
ln.use_graphs = False
step = bench.make_step(replay, ln, 32, fused_next_sample=True)
for _ in range(steps):
  step()
torch.cuda.synchronize()
replay.check_status(); ln.check_status()
print('done')
