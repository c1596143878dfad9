"""Host cost of each launch's enqueue in the FIRST learner step after a synchronize() against
the steady state (dz_prof_enable(2): the library's launch marks take the host clock)."""
import os, sys, types, time, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from dqn_zoo_amd import _lib
args = types.SimpleNamespace(capacity=1000000, batch=32)
dev = torch.device('cuda', 0)
replay, learner, _ = bench.build_workload(args, dev, seed=3)
learner.use_graphs = False
torch.cuda.set_stream(torch.cuda.Stream(dev))
step = bench.make_step(replay, learner, 32, fused_next_sample=True)
lib = _lib.load()
for _ in range(64 + 5):
  step()
torch.cuda.synchronize()
lib.dz_prof_enable(2)
ms = (ctypes.c_float * 96)(); names = ctypes.create_string_buffer(96 * 32)

def read():
  n = lib.dz_prof_read(96, ctypes.addressof(ms), ctypes.addressof(names))
  return [(names.raw[32 * i:32 * i + 32].split(b'\0')[0].decode(), ms[i] * 1e3) for i in range(n)]

mode = os.environ.get('MODE', '')
for trial in range(5):
  torch.cuda.synchronize()
  if mode == 'sleep':
    time.sleep(0.002)
  rows = []
  tot = []
  for i in range(12):
    t0 = time.perf_counter()
    step()
    tot.append((time.perf_counter() - t0) * 1e6)
    rows.append(read())
  print('trial', trial, 'python step() us:', ' '.join('%.0f' % x for x in tot))
  for k in (0, 1, 11):
    print('   step %2d:' % k, ' '.join('%s %.1f' % (n[:10], u) for n, u in rows[k]), '| C total %.1f' % sum(u for _, u in rows[k]))
lib.dz_prof_enable(0)
