import types, torch, sys
sys.path.insert(0, '/root/repo')
import bench
args = types.SimpleNamespace(capacity=2048, batch=32)
dev = torch.device('cuda', 0)
replay, learner, _ = bench.build_workload(args, dev, seed=3)
step = bench.make_step(replay, learner, 32, fused_next_sample=True)
for k in range(300):
  step()
  if k in (0, 1, 5, 20, 100, 299):
    print(k, learner.scalars())
