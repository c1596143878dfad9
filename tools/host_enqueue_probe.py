"""Host cost of enqueueing one fused Rainbow step (default bench mode): the first N steps
after a synchronise are enqueued into an empty queue, so the loop time is host time."""
import os, sys, time, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

args = types.SimpleNamespace(capacity=100000, batch=32)
dev = torch.device('cuda', 0)
replay, learner, _ = bench.build_workload(args, dev, seed=1)
torch.cuda.set_stream(torch.cuda.Stream(dev))
step = bench.make_step(replay, learner, 32, fused_next_sample=True)
for _ in range(100):
  step()
torch.cuda.synchronize()
for n in (10, 30, 100, 300):
  torch.cuda.synchronize()
  t0 = time.perf_counter()
  for _ in range(n):
    step()
  t1 = time.perf_counter()
  torch.cuda.synchronize()
  t2 = time.perf_counter()
  print('%4d steps: host loop %.1f us/step, until drained %.1f us/step' % (
      n, (t1 - t0) / n * 1e6, (t2 - t0) / n * 1e6))
