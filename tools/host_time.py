"""Host enqueue time vs GPU time of the bench step (eager vs hipGraph)."""
import os, sys, time, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

args = types.SimpleNamespace(capacity=100000, batch=32)
dev = torch.device('cuda', 0)
replay, learner, _ = bench.build_workload(args, dev, seed=1)
torch.cuda.synchronize()
torch.cuda.set_stream(torch.cuda.Stream(dev))
for graphs in (False, True):
  learner.use_graphs = graphs
  step = bench.make_step(replay, learner, 32)
  for _ in range(50):
    step()
  torch.cuda.synchronize()
  n = 500
  t0 = time.perf_counter()
  for _ in range(n):
    step()
  t1 = time.perf_counter()
  torch.cuda.synchronize()
  t2 = time.perf_counter()
  # host-only cost of each piece
  t3 = time.perf_counter()
  for _ in range(200):
    s = replay.sample_device(32)
  t4 = time.perf_counter()
  torch.cuda.synchronize()
  tr = s.transitions
  t5 = time.perf_counter()
  for _ in range(200):
    learner.step(tr.s_tm1, tr.a_tm1, tr.r_t, tr.discount_t, tr.s_t, s.weights32)
  t6 = time.perf_counter()
  torch.cuda.synchronize()
  t7 = time.perf_counter()
  for _ in range(200):
    replay.update_priorities(s.ids, learner.priorities)
  t8 = time.perf_counter()
  torch.cuda.synchronize()
  print('graphs=%s: enqueue %.1f us/step, total %.1f us/step | host: sample %.1f learn %.1f update %.1f' % (
      graphs, 1e6 * (t1 - t0) / n, 1e6 * (t2 - t0) / n, 1e6 * (t4 - t3) / 200,
      1e6 * (t6 - t5) / 200, 1e6 * (t8 - t7) / 200))
