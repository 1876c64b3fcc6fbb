mkdir -p gpurun_out/r4k
timeout -k 10 900 python -m pytest tests/test_split_gpu.py tests/test_ops_gpu.py tests/test_fused_bn_gpu.py -x -q > gpurun_out/r4k/tests.log 2>&1 || { tail -40 gpurun_out/r4k/tests.log; exit 1; }
tail -2 gpurun_out/r4k/tests.log
L=gpurun_out/r4k/diag.log
for layer in S1 S3 S5; do
  timeout -k 10 200 python tools/diag_patch.py $layer dgrad 2>/dev/null | grep -v amdgpu.ids >> $L || exit 1
  CSTP_LIB_PATH=$PWD/build_var/g0.so timeout -k 10 200 python tools/diag_patch.py $layer dgrad 2>/dev/null | grep -v amdgpu.ids >> $L || exit 1
done
cat $L
bash tools/ab_same_box.sh r4k_ab $PWD/build_var/g0.so 2 20
