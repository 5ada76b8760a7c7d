cd /root/repo
timeout 900 python -m pytest tests/test_gpu_trajectory.py -x -q -m gpu -k two_passes 2>&1 | tail -15
