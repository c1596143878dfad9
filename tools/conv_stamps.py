"""(DZ_GEMM_STAMPS build only) per-workgroup wall-clock stamps of the three forward conv launches
of the Rainbow step: start, first loads issued, first stage in LDS, MFMAs done, exchange done, stored."""
import os, sys, types, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
args = types.SimpleNamespace(capacity=20000, batch=32)
dev = torch.device('cuda', 0)
replay, learner, _ = bench.build_workload(args, dev, seed=3)
learner.use_graphs = False
torch.cuda.set_stream(torch.cuda.Stream(dev))
step = bench.make_step(replay, learner, 32, fused_next_sample=True)
for _ in range(50): step()
torch.cuda.synchronize()
off = int(learner.layout.c.ws_dfeat_part)
names = ['start', 'loads issued', 'stage0 in LDS', 'mfma done', 'exchanged', 'stored']
for li, (nm, nx, ny) in enumerate((('conv1', 1, 1200), ('conv2', 1, 243), ('conv3', 1, 147))):
  raw = learner.ws[off + li * 65536 * 16: off + (li + 1) * 65536 * 16].cpu().numpy().view(np.int64).reshape(-1, 8)
  idx = [x + 4 * y for y in range(ny) for x in range(nx)]
  r = raw[idx][:, :6]
  t0 = r[:, 0].min()
  us = (r - t0) / 100.0
  print(nm, '%d workgroups; us since the first one started: mean (min..max)' % len(idx))
  for i, n in enumerate(names):
    print('   %-14s %6.2f (%5.2f..%5.2f)' % (n, us[:, i].mean(), us[:, i].min(), us[:, i].max()))
  d = np.diff(us, axis=1).mean(axis=0)
  print('   phases:', ' '.join('%s %.2f' % (names[i + 1], d[i]) for i in range(5)), ' total in-WG %.2f' % (us[:, 5] - us[:, 0]).mean())
