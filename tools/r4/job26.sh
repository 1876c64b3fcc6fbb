mkdir -p gpurun_out/r5b
timeout -k 10 900 python -m pytest tests/test_split_gpu.py tests/test_ops_gpu.py tests/test_fused_bn_gpu.py tests/test_tpatch_gpu.py -x -q -m gpu > gpurun_out/r5b/tests.log 2>&1; rc=$?; tail -3 gpurun_out/r5b/tests.log; [ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python tools/step_table.py > gpurun_out/r5b/table18.log 2>&1; head -12 gpurun_out/r5b/table18.log | cut -c1-110
CSTP_LIB_PATH=build_var/nodeep.so timeout -k 10 300 python tools/step_table.py > gpurun_out/r5b/table18_nodeep.log 2>&1; head -12 gpurun_out/r5b/table18_nodeep.log | cut -c1-110
bash tools/ab_same_box.sh r5b build_var/nodeep.so 3 20
