cd /root/repo
timeout 900 python -m pytest tests/test_gpu_bench.py -x -q -m gpu -k "rung_loop or config4" 2>&1 | tail -5
timeout 600 python bench.py --rung-loop uniform --steps 12 2>&1 | tail -1
timeout 600 python bench.py --rung-loop clustered --steps 10 2>&1 | tail -1
