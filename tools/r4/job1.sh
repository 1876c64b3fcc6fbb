mkdir -p gpurun_out/r4a
python -m pytest tests/test_split_gpu.py tests/test_fused_bn_gpu.py tests/test_tpatch_gpu.py tests/test_ops_gpu.py -x -q > gpurun_out/r4a/tests.log 2>&1 || { tail -30 gpurun_out/r4a/tests.log; exit 1; }
tail -2 gpurun_out/r4a/tests.log
for mode in fwdbn fwd; do
for v in default r3base diag16 diag144 d128; do
  if [ $v = default ]; then timeout -k 10 200 python tools/diag_patch.py S1 $mode >> gpurun_out/r4a/diag.log 2>&1 || exit 1
  else CSTP_LIB_PATH=$PWD/build_var/$v.so timeout -k 10 200 python tools/diag_patch.py S1 $mode >> gpurun_out/r4a/diag.log 2>&1 || exit 1; fi
done; done
cat gpurun_out/r4a/diag.log
bash tools/ab_same_box.sh r4a_ab $PWD/build_var/r3base.so 2 20
