mkdir -p gpurun_out/r4s
timeout -k 10 900 python -m pytest tests/test_ops_gpu.py tests/test_split_gpu.py tests/test_fused_bn_gpu.py -x -q -m gpu > gpurun_out/r4s/tests.log 2>&1; echo "tests rc=$?"; tail -5 gpurun_out/r4s/tests.log
