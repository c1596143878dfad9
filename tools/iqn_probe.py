"""IQN drop-in loop: where a frame's time goes (GPU step alone, decisions alone, host profile)."""
import os, sys, time, cProfile, pstats, io
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import numpy as np, torch
import bench
from dqn_zoo_amd import parts

ag, rep = bench.make_loop_agent('iqn', 4)
loop = parts.run_loop(ag, bench.SyntheticFrames(3), max_steps_per_episode=0)
for _ in range(1500): next(loop)
torch.cuda.synchronize()
if len(sys.argv) > 1 and sys.argv[1] == 'prof':   # HIP-event duration of every launch of the step
  ag._learner.use_graphs = False
  avg = bench.profile_kernels(ag._learn, 30)
  for k, v in avg.items(): print('%-24s %7.2f us' % (k, v * 1e6))
  print('sum %.1f us' % (1e6 * sum(avg.values())))
  sys.exit(0)
if len(sys.argv) > 1:   # learner steps only (`learn`: for rocprofv3 --kernel-trace --stats; `time`: us per step)
  for rep_ in range(3 if sys.argv[1] == 'time' else 1):
    t0 = time.perf_counter()
    for _ in range(300): ag._learn()
    torch.cuda.synchronize()
    print('learn us/step %.1f' % (1e6 * (time.perf_counter() - t0) / 300))
  sys.exit(0)
# (a) the loop
t0 = time.perf_counter()
for _ in range(4000): next(loop)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print('loop us/frame %.1f' % (1e6 * dt / 4000))
# (b) learner steps alone, as the agent enqueues them
t0 = time.perf_counter()
for _ in range(500): ag._learn()
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print('learn us/step %.1f' % (1e6 * dt / 500))
# (c) decisions alone, awaited
ts = bench.SyntheticFrames(3)
x = np.zeros((84, 84, 4), np.uint8)
class T: observation = x
t0 = time.perf_counter()
for _ in range(2000): ag._act(T).resolve()
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print('decision awaited us %.1f' % (1e6 * dt / 2000))
pr = cProfile.Profile(); pr.enable()
for _ in range(4000): next(loop)
torch.cuda.synchronize(); pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(22)
print(s.getvalue()[:5000])
