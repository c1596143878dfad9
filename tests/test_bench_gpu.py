"""bench.py under the DRIVER's command line (`--steps 20 --warmup 5`): the timed
window must contain the 20 steps and nothing else.  Round 1 failed this by 5x
(first-use torch kernels, ReplicaStats and the all-reduce sat inside the clock)."""

import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*flags):
  r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py')] + list(flags),
                     capture_output=True, text=True, timeout=900, cwd=ROOT)
  assert r.returncode == 0, r.stderr[-2000:]
  lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
  assert len(lines) == 1, r.stdout[-2000:]
  return json.loads(lines[0])


@pytest.mark.gpu
def test_driver_command_measures_the_step():
  out = _run('--steps', '20', '--warmup', '5', '--cpu-seconds', '0',
             '--prof-steps', '20', '--other-configs', '0')
  assert out['steps'] == 20 and out['warmup'] == 5 and out['n_gpus'] == 1
  device_us = out['roofline']['learn_kernels_us'] + \
      out['replay']['sample_plus_gather_us']
  # the clock may not hold more than the kernels of the step (+25 % for the
  # event overhead in `device_us` going the other way and host jitter)
  assert out['ms_per_step'] <= 1.25 * device_us / 1000.0, (out['ms_per_step'], device_us)
  assert abs(out['value'] - 1e3 / out['ms_per_step']) / out['value'] < 1e-3
  for key in ('metric', 'unit', 'roofline', 'config', 'dtype', 'scaling'):
    assert key in out
  r = out['roofline']
  assert r['bound'] in ('hbm', 'mfma') and 0 < r['frac'] <= 1


@pytest.mark.gpu
def test_gpus_flag_degrades_to_visible_devices():
  """`bench.py --gpus 8` by hand on a smaller box runs what is there (the
  driver launches N > 1 through torchrun itself)."""
  import torch
  have = torch.cuda.device_count()
  out = _run('--gpus', '8', '--steps', '20', '--warmup', '5', '--cpu-seconds', '0',
             '--prof-steps', '0', '--other-configs', '0', '--capacity', '8192')
  assert out['n_gpus'] == min(8, have)
  assert out['config']['parallelism'] == 'replicas x%d' % min(8, have)


@pytest.mark.gpu
def test_rccl_statistics_all_reduce_under_torchrun_world1():
  """The N > 1 code path (process group on RCCL, barrier, the packed float64
  all-reduce of ReplicaStats, MAX of durations) with one rank: what the driver's
  scaling run executes, minus the peers."""
  import socket
  sk = socket.socket(); sk.bind(('127.0.0.1', 0)); port = sk.getsockname()[1]; sk.close()
  r = subprocess.run(
      [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1',
       '--nproc-per-node', '1', '--master-addr', '127.0.0.1', '--master-port',
       str(port), os.path.join(ROOT, 'bench.py'), '--gpus', '1', '--steps', '20',
       '--warmup', '5', '--cpu-seconds', '0', '--prof-steps', '0',
       '--other-configs', '0', '--capacity', '8192'],
      capture_output=True, text=True, timeout=900, cwd=ROOT)
  assert r.returncode == 0, r.stderr[-2000:]
  lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
  assert len(lines) == 1, r.stdout[-2000:]
  out = json.loads(lines[0])
  assert out['n_gpus'] == 1 and out['value'] > 0
  assert out['config']['collective'] == 'rccl'
