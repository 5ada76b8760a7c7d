cd /root/repo
timeout 1200 python -m pytest tests/test_gpu_p3m.py tests/test_gpu_trajectory.py tests/test_gpu_known_answers.py -x -q -m gpu 2>&1 | tail -4
timeout 600 python tools/soak_p3m.py 0.04 2>&1 | tail -4 | head -1
