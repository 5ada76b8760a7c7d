cd /root/repo
timeout 1500 python -m pytest "tests/test_gpu_bench_kernels_parity.py::test_fused_pass_config3_size_properties" -x -q -m gpu 2>&1 | grep -v Warn | tail -30
