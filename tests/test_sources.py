"""Every Python source of the repo parses (bench.py and __graft_entry__.py are
run by the driver, not imported by the other tests) and the bench's argument
contract is intact."""

import ast
import glob
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_python_sources_parse():
  files = [os.path.join(ROOT, 'bench.py'), os.path.join(ROOT, '__graft_entry__.py')]
  for pat in ('tools/*.py', 'dqn_zoo_amd/*.py', 'dqn_zoo_amd/*/*.py', 'oracle/*.py',
              'tests/*.py', 'tests/golden/*.py'):
    files += glob.glob(os.path.join(ROOT, pat))
  assert len(files) > 30
  for f in files:
    with open(f) as fh:
      ast.parse(fh.read(), filename=f)


def test_bench_contract_flags_and_fields():
  src = open(os.path.join(ROOT, 'bench.py')).read()
  for flag in ("'--gpus'", "'--steps'", "'--warmup'"):
    assert flag in src
  for key in ("'metric'", "'value'", "'unit'", "'n_gpus'", "'steps'", "'warmup'",
              "'ms_per_step'", "'higher_is_better'", "'scaling'", "'vs_baseline'", "'dtype'",
              "'data'", "'config'", "'roofline'", "'cpu_baseline'"):
    assert key in src, key
  entry = open(os.path.join(ROOT, '__graft_entry__.py')).read()
  assert 'def build(' in entry and 'def smoke(' in entry
