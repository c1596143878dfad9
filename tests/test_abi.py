"""The C-ABI library loads and exports every symbol include/*.h declares.

CPU-only: no compute calls are made (cross-compiled device code cannot run
here); this is the "does the boundary exist" check of task §③.
"""

import ctypes
import glob
import os
import re

import pytest

from dqn_zoo_amd import _lib
from dqn_zoo_amd import build as dz_build

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_functions():
  names = {}
  for h in glob.glob(os.path.join(ROOT, 'include', '*.h')):
    text = open(h).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    for m in re.finditer(r'\b(?:int|const char\*)\s+(dz_\w+)\s*\(([^;{]*?)\)\s*;',
                         text, flags=re.S):
      args = m.group(2).strip()
      n = 0 if args in ('', 'void') else args.count(',') + 1
      names[m.group(1)] = n
  return names


@pytest.fixture(scope='module')
def lib():
  dz_build.build()
  return _lib.load()


def test_header_declares_something():
  assert len(_declared_functions()) >= 10


def test_every_declared_symbol_is_exported_and_bound(lib):
  decl = _declared_functions()
  for name, nargs in decl.items():
    assert hasattr(lib, name), '%s declared in include/ but not exported' % name
    assert name in _lib.SIGNATURES, '%s has no ctypes signature' % name
    assert len(_lib.SIGNATURES[name][1]) == nargs, (
        '%s: header has %d args, ctypes table %d' %
        (name, nargs, len(_lib.SIGNATURES[name][1])))
  for name in _lib.SIGNATURES:
    assert name in decl, '%s bound in _lib.py but not declared in include/' % name


def test_identification(lib):
  assert lib.dz_built_arch() == b'gfx950'
  assert b'dqnzoo_hip' in lib.dz_version()


def test_struct_layouts_match_header(lib):
  # sizes computed from the C declarations (LP64): 3x8 and the sample args.
  assert ctypes.sizeof(_lib.FieldDesc) == 24
  assert ctypes.sizeof(_lib.PrioSampleArgs) == 8 * 8 + 5 * 8 + 3 * 4 + 4
  for which, cls in _lib.STRUCT_IDS.items():
    assert lib.dz_struct_size(which) == ctypes.sizeof(cls), cls.__name__
  assert lib.dz_struct_size(99) == -1


def test_rainbow_layout_matches_reference_counts(lib):
  L = _lib.RainbowLayout()
  assert lib.dz_rainbow_layout(6, 51, 32, ctypes.byref(L)) == 0
  assert L.param_count_ref == 6868485
  assert L.param_count >= L.param_count_ref and L.param_count % 4 == 0
  assert L.noise_stride >= 2 * 3136 + 1024 + 1024 + 6 * 51 + 51
  assert L.adv2_ld % 4 == 0 and L.val2_ld % 4 == 0 and L.fc1_ld % 4 == 0
  assert L.conv_b[0] == L.conv_w[0] + 256 * 32   # bias follows weights (wgrad reduce)
  assert lib.dz_rainbow_layout(6, 65, 32, ctypes.byref(L)) == _lib.DZ_ERR_INVALID_ARG


def test_code_object_is_gfx950_only():
  """No multi-arch / compat builds: the fat binary holds exactly gfx950."""
  data = open(_lib.LIB_PATH, 'rb').read()
  targets = set(re.findall(rb'amdgcn-amd-amdhsa--(gfx[0-9a-f]+)', data))
  assert targets == {b'gfx950'}, targets


def test_product_fails_loudly_without_gpu():
  import numpy as np
  import torch
  from dqn_zoo_amd import replay
  if torch.cuda.is_available():
    pytest.skip('GPU present')
  with pytest.raises(_lib.HipLibraryError):
    replay.TransitionReplay(8, replay.Transition(None, None, None, None, None),
                            np.random.RandomState(0))


def test_div255_identity():
  """dz_div255 (csrc/dz_qnet_ops.h): y = x*fl(1/255); r = fma(-y, 255, x);
  y' = fma(r, fl(1/255), y) equals the IEEE float32 division x/255 for every
  uint8 x (networks.py:193).  FMAs are emulated exactly in float64 (24-bit x
  8-bit products and their sums fit in 53 bits)."""
  import numpy as np
  x = np.arange(256, dtype=np.float32)
  true = x / np.float32(255.0)
  rc = np.float32(1.0) / np.float32(255.0)
  y = x * rc
  r = (-y.astype(np.float64) * 255.0 + x.astype(np.float64)).astype(np.float32)
  y2 = (r.astype(np.float64) * np.float64(rc) + y.astype(np.float64)).astype(np.float32)
  assert (y != true).any()          # the plain reciprocal multiply is NOT exact
  np.testing.assert_array_equal(y2, true)


import numpy as np  # noqa: E402


def _rn32(x):
  """Fraction -> nearest float32 (ties to even), normal range, exact integer arithmetic."""
  from fractions import Fraction
  if x == 0:
    return np.float32(0.0)
  s = -1 if x < 0 else 1
  x = abs(x)
  e = x.numerator.bit_length() - x.denominator.bit_length() - 24
  while x / Fraction(2) ** e >= 2 ** 24:
    e += 1
  while x / Fraction(2) ** e < 2 ** 23:
    e -= 1
  y = x / Fraction(2) ** e
  n = y.numerator // y.denominator
  rem = y - n
  if rem > Fraction(1, 2) or (rem == Fraction(1, 2) and n % 2 == 1):
    n += 1
  return np.float32(s * float(n) * 2.0 ** e)   # n <= 2^24 and the power: exact in float64


def test_div_by_uniform_divisor_is_the_ieee_quotient():
  """dz_div_by (csrc/dz_qnet_kernels.h; Adam's M / bc1, V / bc2, G / gnorm):
  q = a * fl(1/b); q' = fma(fma(-q, b, a), fl(1/b), q) is the correctly rounded a / b
  (optax: mu_hat = mu / bc1 ...), checked with exact rational arithmetic on the divisors
  an optimiser run produces (bias corrections of every early step, a spread of later
  ones, gradient norms) against moments from 1e-30 to 1e3."""
  from fractions import Fraction
  rs = np.random.RandomState(0)
  divisors = [np.float32(1.0) - np.float32(0.9) ** np.float32(c) for c in range(1, 60)]
  divisors += [np.float32(1.0) - np.float32(0.999) ** np.float32(c) for c in (1, 2, 3, 7, 50, 333, 2500, 9000)]
  divisors += list(rs.uniform(0.01, 30.0, 40).astype(np.float32))
  bad = 0
  for b in divisors:
    b = np.float32(b)
    rb = np.float32(1.0) / b
    for a in (10.0 ** rs.uniform(-30, 3, 60) * rs.choice([-1.0, 1.0], 60)).astype(np.float32):
      q = np.float32(a * rb)
      r = _rn32(Fraction(float(a)) - Fraction(float(q)) * Fraction(float(b)))
      got = _rn32(Fraction(float(q)) + Fraction(float(r)) * Fraction(float(rb)))
      want = _rn32(Fraction(float(a)) / Fraction(float(b)))
      assert want == np.float32(a / b)          # (numpy's division is the IEEE one)
      bad += got != want
  assert bad == 0


def test_kernels_with_hand_counted_waits_do_not_spill():
  """`fc1_dgrad_mfma_kernel` issues LDS-DMA loads from inline assembly and counts its own
  `s_waitcnt vmcnt(N)`; a register spill (scratch loads and stores count in vmcnt) between the
  first DMA and the last wait would make the MFMAs read LDS too early.  The build must report
  ScratchSize 0 for it -- and for the seam kernels, whose polling loops were the round-4 spill
  source (rainbow_act_one_kernel) -- and 135 KB of LDS needs gfx950's 160 KB."""
  import subprocess
  out = ''
  for unit in ('dz_rainbow.hip', 'dz_iqn.hip'):   # (dz_iqn: the LDS-DMA contractions, dz_dma_gemm.h)
    src = os.path.join(ROOT, 'dqn_zoo_amd', 'csrc', unit)
    cmd = ['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC',
           '-ffp-contract=off', '-fno-fast-math', '-I', os.path.join(ROOT, 'include'),
           '-Rpass-analysis=kernel-resource-usage', '-c', src, '-o', os.devnull]
    out += subprocess.run(cmd, capture_output=True, text=True, timeout=600).stderr
  usage, name = {}, None
  for line in out.splitlines():
    m = re.search(r'Function Name: (\S+)', line)
    if m:
      name = m.group(1)
    m = re.search(r'ScratchSize \[bytes/lane\]: (\d+)', line)
    if m and name:
      usage.setdefault(name, {})['scratch'] = int(m.group(1))
    m = re.search(r'LDS Size \[bytes/block\]: (\d+)', line)
    if m and name:
      usage.setdefault(name, {})['lds'] = int(m.group(1))
  checked = 0
  for k, u in usage.items():
    # (every instantiation of the head launch -- 1..4 column chunks of the advantage head, i.e. up
    # to the 18-action games -- and the kernels with hand-counted waits: no scratch)
    if any(t in k for t in ('fc1_dgrad_mfma_kernel', 'rainbow_head_chain_kernel', 'fc2_bwd_rows_kernel',
                            'rainbow_act_one_kernel', 'dz_dma_gemm2_kernel', 'dz_conv_dma_fwd_kernel',
                            'dz_conv1_dma_kernel', 'dz_dmaop2_kernel')):
      assert u['scratch'] == 0, (k, u)
      assert u['lds'] <= 160 * 1024, (k, u)
      checked += 1
  assert checked >= 14, sorted(usage)
