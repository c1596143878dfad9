"""Builds libdqnzoo_hip.so in-tree with hipcc for gfx950 (no JIT cache).

    python -m dqn_zoo_amd.build [--force]

Each .hip file is compiled to an object under dqn_zoo_amd/csrc/_obj/ (only when
it or a header changed) and linked into dqn_zoo_amd/libdqnzoo_hip.so.  The
shared object is git-ignored but travels to the GPU box with the snapshot.
"""

import concurrent.futures
import glob
import os
import shutil
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, 'csrc')
OBJ = os.path.join(CSRC, '_obj')
LIB = os.path.join(PKG, 'libdqnzoo_hip.so')
INCLUDE = os.path.join(os.path.dirname(PKG), 'include')

ARCH = 'gfx950'
# -ffp-contract=off: the replay arithmetic must round exactly like NumPy
# (mul, mul, add -- never an FMA); GEMM kernels use MFMA / explicit fmaf.
CXXFLAGS = [
    '--offload-arch=' + ARCH, '-O3', '-std=c++17', '-fPIC',
    '-ffp-contract=off', '-fno-fast-math', '-Wall', '-Wno-unused-function',
    '-I', INCLUDE,
]


def _hipcc():
  exe = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
  if not os.path.exists(exe):
    raise RuntimeError('hipcc not found; cannot build libdqnzoo_hip.so')
  return exe


def _newest_header_mtime():
  hs = glob.glob(os.path.join(CSRC, '*.h')) + glob.glob(
      os.path.join(INCLUDE, '*.h'))
  return max(os.path.getmtime(h) for h in hs)


def _compile(src, obj, verbose):
  cmd = [_hipcc()] + CXXFLAGS + ['-c', src, '-o', obj]
  if verbose:
    print(' '.join(cmd), flush=True)
  r = subprocess.run(cmd, capture_output=True, text=True)
  if r.returncode != 0:
    raise RuntimeError('hipcc failed for %s:\n%s\n%s' % (src, r.stdout, r.stderr))
  if r.stderr.strip() and verbose:
    print(r.stderr, file=sys.stderr)


def build(force=False, verbose=False):
  """Compiles every csrc/*.hip for gfx950 and links the shared library."""
  os.makedirs(OBJ, exist_ok=True)
  srcs = sorted(glob.glob(os.path.join(CSRC, '*.hip')))
  if not srcs:
    raise RuntimeError('no HIP sources under %s' % CSRC)
  hdr = _newest_header_mtime()
  jobs, objs = [], []
  for s in srcs:
    o = os.path.join(OBJ, os.path.basename(s)[:-4] + '.o')
    objs.append(o)
    stale = (force or not os.path.exists(o) or
             os.path.getmtime(o) < max(os.path.getmtime(s), hdr))
    if stale:
      jobs.append((s, o))
  if jobs:
    with concurrent.futures.ThreadPoolExecutor(max_workers=8) as ex:
      list(ex.map(lambda so: _compile(so[0], so[1], verbose), jobs))
  need_link = (force or bool(jobs) or not os.path.exists(LIB) or
               any(os.path.getmtime(o) > os.path.getmtime(LIB) for o in objs))
  if need_link:
    cmd = [_hipcc(), '--offload-arch=' + ARCH, '-shared', '-fPIC', '-o', LIB
          ] + objs
    if verbose:
      print(' '.join(cmd), flush=True)
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
      raise RuntimeError('link failed:\n%s\n%s' % (r.stdout, r.stderr))
  return LIB


if __name__ == '__main__':
  print(build(force='--force' in sys.argv, verbose=True))
