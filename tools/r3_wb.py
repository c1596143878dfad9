"""Times the standalone priority write-back kernel and the sample+gather launch."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, types
import bench
from dqn_zoo_amd import _lib
args = types.SimpleNamespace(capacity=1000000, batch=32)
dev = torch.device('cuda', 0)
replay, learner, _ = bench.build_workload(args, dev, seed=3)
torch.cuda.set_stream(torch.cuda.Stream(dev))
r = bench.measure_replay(replay, learner, 32, n=200)
print(r)
