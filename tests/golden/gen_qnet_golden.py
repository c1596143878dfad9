"""Writes tests/golden/qnet_*.npz: float64 outputs of oracle/qnet_oracle.py on
the seeded cases of qnet_cases.py (see that module's docstring for what this
does and does not pin).

    python tests/golden/gen_qnet_golden.py
"""

import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, '..', '..'))

from oracle import qnet_oracle as qo  # noqa: E402
from tests.golden import qnet_cases as qc  # noqa: E402


def main():
  for name in qc.CASES:
    inp = qc.make_inputs(name)
    out = qc.pack(qc.oracle_step(name, inp))
    np.savez_compressed(os.path.join(HERE, 'qnet_%s.npz' % name), **out)
    print('wrote qnet_%s.npz  loss=%.12g gnorm=%.6g' % (name, out['loss'],
                                                        out['gnorm']))
  proj = np.stack([qo.categorical_l2_project(zp, p, qc.SUPPORT)
                   for zp, p in qc.projection_cases()])
  np.savez_compressed(os.path.join(HERE, 'qnet_projection.npz'), m=proj)
  print('wrote qnet_projection.npz')


if __name__ == '__main__':
  main()
