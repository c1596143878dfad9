ulimit -c 0
for v in "$@"; do
  timeout 200 python tools/agent_bench.py 13=$v iqn 2>&1 | grep -v amdgpu | cut -c1-110
  timeout 200 python -c "
from dqn_zoo_amd import _lib
_lib.load().dz_set_tuning(13, $v)
import pytest, sys
sys.exit(pytest.main(['tests/test_iqn_gpu.py', '-x', '-q', '-m', 'gpu', '-k', 'test_iqn_step']))" 2>&1 | tail -1
done
