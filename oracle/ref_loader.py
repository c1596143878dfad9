"""Loads the *actual* reference `dqn_zoo/replay.py` from /root/reference.

TEST INFRASTRUCTURE ONLY.  Nothing under `dqn_zoo_amd/` may import this module;
only `tests/`, golden-vector generators, and `bench.py`'s cpu_baseline leg do.

The reference module needs three imports that are not installed in this image
(`dm_env`, `snappy`, `dqn_zoo.parts`); SURVEY.md §8c shows that stubbing them in
`sys.modules` is enough because `replay.py` only touches `dm_env.TimeStep` (as a
type annotation), `snappy.compress/uncompress` (inside `compress_array`, unused
here) and `parts.Action` (= int).

The reference tree is read-only and does not exist on the GPU box, so
`load_reference_replay()` returns None when it is absent and callers must skip.
"""

import importlib.util
import os
import sys
import types

REFERENCE_ROOT = os.environ.get('DQN_ZOO_REFERENCE', '/root/reference')


def reference_available() -> bool:
  return os.path.isfile(os.path.join(REFERENCE_ROOT, 'dqn_zoo', 'replay.py'))


def load_reference_replay():
  """Returns the reference `replay` module object, or None if unavailable."""
  if not reference_available():
    return None
  cached = sys.modules.get('_ref_dqn_zoo_replay')
  if cached is not None:
    return cached

  sys.dont_write_bytecode = True  # never write __pycache__ into the reference.

  saved = {k: sys.modules.get(k) for k in ('dm_env', 'snappy', 'dqn_zoo',
                                           'dqn_zoo.parts')}
  try:
    dm_env = types.ModuleType('dm_env')

    class TimeStep(tuple):  # annotation-only use in replay.py
      pass

    dm_env.TimeStep = TimeStep
    snappy = types.ModuleType('snappy')
    pkg = types.ModuleType('dqn_zoo')
    pkg.__path__ = []
    parts = types.ModuleType('dqn_zoo.parts')
    parts.Action = int
    pkg.parts = parts
    sys.modules.update({'dm_env': dm_env, 'snappy': snappy, 'dqn_zoo': pkg,
                        'dqn_zoo.parts': parts})

    spec = importlib.util.spec_from_file_location(
        '_ref_dqn_zoo_replay',
        os.path.join(REFERENCE_ROOT, 'dqn_zoo', 'replay.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    sys.modules['_ref_dqn_zoo_replay'] = mod
    return mod
  finally:
    for k, v in saved.items():
      if v is None:
        sys.modules.pop(k, None)
      else:
        sys.modules[k] = v
