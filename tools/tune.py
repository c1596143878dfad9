"""Sweeps kernel variants / split factors of the learner step on the GPU and
prints per-kernel HIP-event timings (one gpurun session for many configs)."""
import ctypes
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from dqn_zoo_amd import _lib, learner as learner_lib, networks


def timings(ln, dev, steps=30, phases=_lib.PHASE_ALL):
  lib = _lib.load()
  lib.dz_prof_enable(1)
  ms = (ctypes.c_float * 96)()
  names = ctypes.create_string_buffer(96 * 32)
  acc = {}
  for i in range(steps + 5):
    ln.step(*dev, phases=phases, resample_noise=False)
    torch.cuda.synchronize()
    n = lib.dz_prof_read(96, ctypes.addressof(ms), ctypes.addressof(names))
    if i < 5:
      continue
    for j in range(n):
      nm = names.raw[32 * j:32 * j + 32].split(b'\0')[0].decode()
      acc.setdefault(nm, []).append(ms[j] * 1e3)
  lib.dz_prof_enable(0)
  return {k: float(np.median(v)) for k, v in acc.items()}


def main():
  A, B = 6, 32
  sup = np.linspace(-10, 10, 51).astype(np.float32)
  ln = learner_lib.RainbowLearner(networks.RainbowNetwork(A, sup),
                                  learner_lib.AdamConfig(), B)
  g = torch.Generator(device='cuda'); g.manual_seed(0)
  dev = (torch.randint(0, 256, (B, 84, 84, 4), dtype=torch.uint8, device='cuda', generator=g),
         torch.randint(0, A, (B,), device='cuda', generator=g),
         torch.randn(B, dtype=torch.float64, device='cuda', generator=g),
         torch.full((B,), 0.97, dtype=torch.float64, device='cuda'),
         torch.randint(0, 256, (B, 84, 84, 4), dtype=torch.uint8, device='cuda', generator=g),
         torch.rand(B, dtype=torch.float32, device='cuda', generator=g))
  ln.resample_noise()
  lib = _lib.load()
  sweep = [a.split('=') for a in sys.argv[1:]] or [('18', '0'), ('18', '1')] * 3
  for key, val in sweep:  # e.g. `tune.py 18=0 18=1`: one timing table per setting
    lib.dz_set_tuning(int(key), int(val))
    t = timings(ln, dev, steps=20, phases=_lib.PHASE_ALL)
    top = sorted(t.items(), key=lambda kv: -kv[1])
    if os.environ.get('TUNE_ONLY'):
      top = [kv for kv in top if os.environ['TUNE_ONLY'] in kv[0]]
    elif len(sweep) > 2:
      top = top[:4]
    print('key %s = %s: total %.1f us  %s' % (
        key, val, sum(t.values()), '  '.join('%s %.2f' % kv for kv in top)), flush=True)


if __name__ == '__main__':
  main()
