"""Host cost (us per call, GPU drained every 64 calls) of every piece of a learning frame of the
Rainbow drop-in loop: what stands between the frame's start and its first learner launch."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from dqn_zoo_amd import parts

torch.cuda.set_stream(torch.cuda.Stream())
ag, rep = bench.make_loop_agent('rainbow', 4)
loop = parts.run_loop(ag, bench.SyntheticFrames(3), max_steps_per_episode=0)
for _ in range(1200):
  next(loop)
torch.cuda.synchronize()
ln = ag._learner
env = bench.SyntheticFrames(5)
ts = env.reset()
obs = [np.random.randint(0, 256, (84, 84, 4)).astype(np.uint8) for _ in range(8)]

def bench_piece(name, fn, n=1500):
  for i in range(64):
    fn(i)
  torch.cuda.synchronize()
  tot = 0.0
  for i in range(n):
    t0 = time.perf_counter()
    fn(i)
    tot += time.perf_counter() - t0
    if i % 64 == 63:
      torch.cuda.synchronize()
  torch.cuda.synchronize()
  print('%-44s %6.2f us' % (name, 1e6 * tot / n))

bench_piece('obs.upload', lambda i: ag._obs.upload(obs[i & 7]))
od = ag._obs.upload(obs[0])
def act(i):
  r = ln.apply_async(od)
  r()
bench_piece('apply_async + read (20.5 us kernel inside)', act)
bench_piece('apply_async (no read)', lambda i: ln.apply_async(od), 600)
tr = None
from dqn_zoo_amd import replay as replay_lib
o1, o2 = obs[1], obs[2]
ag._obs.upload(o1); ag._obs.upload(o2)
t_host = replay_lib.Transition(o1, 3, 1.0, 0.99, o2)
bench_piece('obs.on_device(transition)', lambda i: ag._obs.on_device(t_host))
t_dev = ag._obs.on_device(t_host)
bench_piece('replay.add_with_device_priority', lambda i: rep.add_with_device_priority(t_dev))
bench_piece('  ring.insert_fields', lambda i: rep._ring.insert_fields(t_dev))
bench_piece('replay.sample_device', lambda i: rep.sample_device(32))
rs = rep._random_state
bench_piece('  rs.randint(size, 32)', lambda i: rs.randint(rep._size, size=32))
bench_piece('  rs.uniform(size=32)', lambda i: rs.uniform(size=32))
bench_piece('  rs.random_sample(32)', lambda i: rs.random_sample(32))
bench_piece('  importance_sampling_exponent', lambda i: rep.importance_sampling_exponent)
s = rep.sample_device(32)
t = s.transitions
bench_piece('replay.priority_sink(ids)', lambda i: rep.priority_sink(s.ids))
sink = rep.priority_sink(s.ids)
bench_piece('learner.step (eager, 11 launches)', lambda i: ln.step(t.s_tm1, t.a_tm1, t.r_t, t.discount_t, t.s_t, s.weights32, priority_sink=sink), 600)
bench_piece('replay.poll_status', lambda i: rep.poll_status())
acc = ag._transition_accumulator
import dqn_zoo_amd.dm_env_shim as dm_env
tsx = dm_env.transition(1.0, obs[3], 0.99)
bench_piece('accumulator.step (n-step 3)', lambda i: list(acc.step(tsx, 2)))
