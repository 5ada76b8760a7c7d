cd /root/repo
timeout 600 python tools/soak_p3m.py 0.04 2>&1 | tail -4 | head -1
timeout 600 python tools/soak_p3m.py 0.04 2>&1 | tail -4 | head -1
SOAK_DIST=clustered timeout 900 python tools/soak_p3m.py 0.025 2>&1 | tail -4 | head -1
timeout 900 python -m pytest tests/test_gpu_trajectory.py -x -q -m gpu 2>&1 | tail -2
