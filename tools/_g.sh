cd /root/repo
timeout 1200 python -m pytest tests/test_gpu_trajectory.py tests/test_gpu_p3m.py -x -q -m gpu 2>&1 | tail -2
python tools/sr_rung_cost.py uniform 2>&1 | grep 'by cell' | cut -c1-150
timeout 600 python tools/soak_p3m.py 0.04 2>&1 | tail -4 | head -1
SOAK_DIST=clustered timeout 900 python tools/soak_p3m.py 0.025 2>&1 | tail -4 | head -2
