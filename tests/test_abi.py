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
