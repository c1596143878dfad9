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
  names = {0: '<1,2,2,KT4>', 1: '<1,2,2,KT2>', 2: '<1,2,2,KT1>', 3: '<1,4,1,KT4>',
           4: '<1,4,1,KT2>', 5: '<1,1,4,KT2>', 6: '<1,1,4,KT1>', 7: '<1,4,1,KT1>'}
  names[8] = 'stream'; names[9] = 'stream2'
  for dg in (0, 1):
    lib.dz_set_tuning(2, dg)
    lib.dz_set_tuning(0, 8); lib.dz_set_tuning(1, 8)
    t = timings(ln, dev, steps=20, phases=_lib.PHASE_ALL)
    print('fc1_dgrad stream=%d: %.2f us (+reduce %.2f)  fc1_wgrad %.2f adam %.2f' % (
        dg, t['fc1_dgrad'], t.get('fc1_dgrad_reduce', 0.0), t['fc1_wgrad'], t['adam']))
  lib.dz_set_tuning(3, 1); lib.dz_set_tuning(0, 9); lib.dz_set_tuning(1, 32)
  t = timings(ln, dev, steps=20, phases=_lib.PHASE_FORWARD)
  print('fc1 stream2 BLOCKED-ADDRESS experiment S=32: fc1_fwd %.2f us' % t['fc1_fwd'])
  lib.dz_set_tuning(3, 0)
  for var in (9, 8):
    for splits in ((32,) if var == 9 else (8,)):
      lib.dz_set_tuning(0, var)
      lib.dz_set_tuning(1, splits)
      t = timings(ln, dev, steps=20, phases=_lib.PHASE_FORWARD)
      print('fc1 %-12s S=%2d  fc1_fwd %7.2f us  fc1_epi %5.2f  (conv1 %5.1f conv2 %5.1f conv3 %5.1f fc2 %5.1f head %5.1f)' % (
          names[var], splits, t['fc1_fwd'], t['fc1_epilogue'], t['conv1_fwd'],
          t['conv2_fwd'], t['conv3_fwd'], t['fc2_fwd'], t['head_loss']), flush=True)


if __name__ == '__main__':
  main()
