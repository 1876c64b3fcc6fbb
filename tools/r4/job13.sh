mkdir -p gpurun_out/r4m
timeout -k 10 900 python -m pytest tests/test_tpatch_gpu.py tests/test_ops_gpu.py tests/test_fused_bn_gpu.py -x -q > gpurun_out/r4m/tests.log 2>&1 || { tail -40 gpurun_out/r4m/tests.log; exit 1; }
tail -2 gpurun_out/r4m/tests.log
timeout -k 10 200 python tools/ab_wgrad.py tile=1,8,8,0 T3 2>/dev/null | grep -v amdgpu > gpurun_out/r4m/wgrad.log
timeout -k 10 200 python tools/ab_wgrad.py tile=2,9,1,0 T3 2>/dev/null | grep -v amdgpu >> gpurun_out/r4m/wgrad.log
cat gpurun_out/r4m/wgrad.log
