"""Loads the *actual* reference `dqn_zoo/processors.py` from /root/reference.

TEST INFRASTRUCTURE ONLY (same rules as ref_loader.py): used by the golden-frame
generator tests/golden/gen_processors_golden.py and by CPU tests in the dev
container; returns None on the GPU box, where /root/reference does not exist.

The module needs `dm_env` (TimeStep, StepType, Environment, specs) and
`chex.assert_rank`, neither installed here; they are stubbed in `sys.modules`
for the duration of the import with the package's own dm_env stand-in
(dqn_zoo_amd/dm_env_shim.py: same field order, FIRST == 0).  `numpy` and
`PIL.Image` are the real ones: the pixel arithmetic that is being pinned (BLAS
tensordot for `rgb2y`, Pillow's BILINEAR resample) is the reference's own.
"""

import importlib.util
import os
import sys
import types

import numpy as np

from oracle import ref_loader


def reference_available() -> bool:
  return os.path.isfile(os.path.join(ref_loader.REFERENCE_ROOT, 'dqn_zoo',
                                     'processors.py'))


def load_reference_processors():
  if not reference_available():
    return None
  cached = sys.modules.get('_ref_dqn_zoo_processors')
  if cached is not None:
    return cached
  sys.dont_write_bytecode = True
  from dqn_zoo_amd import dm_env_shim as shim

  dm_env = types.ModuleType('dm_env')
  dm_env.TimeStep = shim.TimeStep
  dm_env.StepType = shim.StepType
  dm_env.Environment = type('Environment', (), {})
  specs = types.ModuleType('dm_env.specs')

  class Array:  # annotation / wrapper use only
    def __init__(self, shape=(), dtype=None, name=None):
      self.shape, self.dtype, self.name = shape, dtype, name

  specs.Array = Array
  specs.DiscreteArray = type('DiscreteArray', (Array,), {})
  dm_env.specs = specs
  chex = types.ModuleType('chex')

  def assert_rank(a, rank):
    if np.ndim(a) != rank:
      raise AssertionError('rank %d != %d' % (np.ndim(a), rank))

  chex.assert_rank = assert_rank
  saved = {k: sys.modules.get(k) for k in ('dm_env', 'dm_env.specs', 'chex')}
  try:
    sys.modules.update({'dm_env': dm_env, 'dm_env.specs': specs, 'chex': chex})
    spec = importlib.util.spec_from_file_location(
        '_ref_dqn_zoo_processors',
        os.path.join(ref_loader.REFERENCE_ROOT, 'dqn_zoo', 'processors.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    sys.modules['_ref_dqn_zoo_processors'] = mod
    return mod
  finally:
    for k, v in saved.items():
      if v is None:
        sys.modules.pop(k, None)
      else:
        sys.modules[k] = v
