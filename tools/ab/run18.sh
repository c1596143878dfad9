python -m pytest tests/test_iqn_gpu.py -x -q 2>&1 | tail -4
python tools/iqn_probe.py prof 2>&1 | grep -v amdgpu | tail -22
python tools/iqn_probe.py time 2>&1 | grep "learn us"
python tools/iqn_probe.py 2>&1 | grep -E "loop us|learn us|decision"
