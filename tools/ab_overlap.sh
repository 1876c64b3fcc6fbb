set -e
for i in 1 2; do timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | grep "^{" | cut -c60-180; done
timeout -k 10 600 python -m pytest tests/test_ops_gpu.py tests/test_model_gpu.py -x -q 2>&1 | tail -1
