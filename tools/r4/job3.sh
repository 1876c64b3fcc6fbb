# QUAD staging: parity of the patch kernels, then S1 / S3 forward + data gradient with and without it; priority / split variants (old staging)
mkdir -p gpurun_out/r4c
python -m pytest tests/test_split_gpu.py tests/test_fused_bn_gpu.py tests/test_ops_gpu.py -x -q > gpurun_out/r4c/tests.log 2>&1 || { tail -30 gpurun_out/r4c/tests.log; exit 1; }
tail -2 gpurun_out/r4c/tests.log
L=gpurun_out/r4c/diag.log
run() { timeout -k 10 200 python tools/diag_patch.py "$@" 2>/dev/null | grep -v amdgpu.ids >> $L || exit 1; }
for layer in S1 S3; do for mode in fwd fwdbn dgrad; do
  echo "# quad on" >> $L; run $layer $mode
  echo "# quad off" >> $L; CSTP_K1P_QUAD=0 run $layer $mode
done; done
for v in p0 p0s2 p1s3 x5 x6 x3; do
  echo "# variant $v (dword staging)" >> $L
  CSTP_K1P_QUAD=0 CSTP_LIB_PATH=$PWD/build_var/$v.so run S1 fwd
  CSTP_K1P_QUAD=0 CSTP_LIB_PATH=$PWD/build_var/$v.so run S1 dgrad
done
cat $L
