set -e
timeout -k 10 900 python -m pytest tests/test_ops_gpu.py tests/test_model_gpu.py tests/test_ft_gpu.py tests/test_r3d_gpu.py -x -q 2>&1 | tail -2
for i in 1 2; do CSTP_DEBUG=1 timeout -k 10 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | tail -2 | cut -c1-200; done
