"""Per-step GPU time of the first steps after a synchronize (the driver's 20-step window), from one
event per step: where does the window lose its 1-2 % against the sustained rate?"""
import os, sys, types, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
if os.environ.get('MODE') == 'schedspin':
  import ctypes
  hip = ctypes.CDLL('libamdhip64.so')
  print('hipSetDeviceFlags(spin) ->', hip.hipSetDeviceFlags(1))
import bench
args = types.SimpleNamespace(capacity=1000000, batch=32)
dev = torch.device('cuda', 0)
replay, learner, _ = bench.build_workload(args, dev, seed=3)
learner.use_graphs = False
torch.cuda.set_stream(torch.cuda.Stream(dev))
step = bench.make_step(replay, learner, 32, fused_next_sample=True)
for _ in range(64 + 5):
  step()
mode = os.environ.get('MODE', '')
for trial in range(4):
  torch.cuda.synchronize()
  torch.cuda.synchronize()
  if mode == 'spin':       # keep the core busy for 300 us before the window
    t = time.perf_counter()
    while time.perf_counter() - t < 300e-6:
      pass
  n = 40
  ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
  t0 = time.perf_counter()
  ev[0].record()
  host = []
  for i in range(n):
    h0 = time.perf_counter()
    step()
    ev[i + 1].record()
    host.append((time.perf_counter() - h0) * 1e6)
  torch.cuda.synchronize()
  wall = (time.perf_counter() - t0) * 1e6
  d = [ev[i].elapsed_time(ev[i + 1]) * 1e3 for i in range(n)]
  print('trial', trial, 'wall/step %.1f' % (wall / n), 'first 20 steps: gpu %.1f us/step; steps 21-40: %.1f' % (sum(d[:20]) / 20, sum(d[20:]) / 20))
  print('  per step:', ' '.join('%.0f' % x for x in d))
  print('  host enqueue:', ' '.join('%.0f' % x for x in host[:20]))
