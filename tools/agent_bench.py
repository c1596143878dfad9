"""Learner-step throughput and per-kernel times of every agent's update on a
fixed on-device batch (no replay): python tools/agent_bench.py [names...]."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from dqn_zoo_amd import _lib, learner as ll, networks

A, B = 18, 32


def prof(lib, step, n=20):
  lib.dz_prof_enable(1)
  ms = (ctypes.c_float * 96)()
  names = ctypes.create_string_buffer(96 * 32)
  acc = {}
  for _ in range(n):
    step()
    torch.cuda.synchronize()
    k = lib.dz_prof_read(96, ctypes.addressof(ms), ctypes.addressof(names))
    for i in range(k):
      nm = names.raw[32 * i:32 * i + 32].split(b'\0')[0].decode()
      acc.setdefault(nm, []).append(ms[i] * 1e3)
  lib.dz_prof_enable(0)
  return {k: float(np.mean(v)) for k, v in acc.items()}


def main():
  names = sys.argv[1:] or ['dqn', 'double_q', 'c51', 'qr', 'iqn']
  lib = _lib.load()
  dev = torch.device('cuda', 0)
  torch.cuda.set_stream(torch.cuda.Stream(dev))
  rs = np.random.RandomState(0)
  s_tm1 = torch.from_numpy(rs.randint(0, 256, (B, 84, 84, 4)).astype(np.uint8)).to(dev)
  s_t = torch.from_numpy(rs.randint(0, 256, (B, 84, 84, 4)).astype(np.uint8)).to(dev)
  a = torch.from_numpy(rs.randint(A, size=B).astype(np.int64)).to(dev)
  r = torch.from_numpy(rs.choice([-1.0, 0.0, 1.0], size=B)).to(dev)
  d = torch.from_numpy(rs.choice([0.0, 0.99], size=B)).to(dev)
  w = torch.from_numpy(rs.uniform(0.2, 1, size=B).astype(np.float32)).to(dev)
  rms, adam = ll.RmsPropConfig(), ll.AdamConfig(learning_rate=5e-5, eps=0.01 / 32)
  for name in names:
    if name == 'dqn_full':  # uniform replay sample + DQN update (BASELINE configs[1] shape)
      from dqn_zoo_amd import replay as rl
      cap = 100000
      rep = rl.TransitionReplay(cap, rl.Transition(None, None, None, None, None),
                                np.random.RandomState(1))
      g = torch.Generator(device=dev); g.manual_seed(0)
      pool = torch.randint(0, 256, (256, 84, 84, 4), dtype=torch.uint8, device=dev, generator=g)
      for lo in range(0, cap, 4096):
        n = min(4096, cap - lo)
        i1 = torch.randint(0, 256, (n,), device=dev, generator=g)
        rep.bulk_fill([pool[i1], torch.randint(0, A, (n,), device=dev, generator=g),
                       torch.zeros(n, dtype=torch.float64, device=dev),
                       torch.full((n,), 0.99, dtype=torch.float64, device=dev), pool[i1]])
      ln = ll.DenseLearner(networks.DenseNetwork('dqn', A), 'q', rms, B)
      def step():
        t, _ = rep.sample_device(B)
        ln.step(t.s_tm1, t.a_tm1, t.r_t, t.discount_t, t.s_t, None)
    elif name == 'iqn':
      ln = ll.IqnLearner(networks.IqnNetwork(A, 64), adam._replace(max_global_grad_norm=0.0), B)
      step = lambda: ln.step(s_tm1, a, r, d, s_t)
    else:
      kind = {'dqn': 'dqn', 'double_q': 'double_dqn', 'c51': 'c51', 'qr': 'qr'}[name]
      loss = {'dqn': 'q', 'double_q': 'double_q', 'c51': 'categorical', 'qr': 'quantile'}[name]
      net = networks.DenseNetwork(kind, A, support=np.linspace(-10, 10, 51),
                                  quantiles=(np.arange(201) + 0.5) / 201)
      ln = ll.DenseLearner(net, loss, rms if name in ('dqn', 'double_q') else adam, B)
      wt = w if name == 'double_q' else None
      step = lambda: ln.step(s_tm1, a, r, d, s_t, wt)
    for _ in range(20):
      step()
    torch.cuda.synchronize()
    n = 200
    t0 = time.perf_counter()
    for _ in range(n):
      step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    ln.use_graphs = False   # per-kernel event marks need eager launches
    pk = prof(lib, step)
    print('%-9s %8.1f steps/s  %7.1f us/step  (kernel sum %.1f us)' % (
        name, 1 / dt, dt * 1e6, sum(pk.values())))
    print('   ' + '  '.join('%s %.1f' % kv for kv in sorted(pk.items(), key=lambda kv: -kv[1])))


if __name__ == '__main__':
  main()
