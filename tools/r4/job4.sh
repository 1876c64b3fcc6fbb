# igemm_k1w: parity, then T1 forward timings (gather AFF vs ring vs weight-resident), plain and as the step issues it
mkdir -p gpurun_out/r4d
python -m pytest tests/test_tpatch_gpu.py tests/test_fused_bn_gpu.py -x -q > gpurun_out/r4d/tests.log 2>&1 || { tail -40 gpurun_out/r4d/tests.log; exit 1; }
tail -2 gpurun_out/r4d/tests.log
L=gpurun_out/r4d/diag.log
for tile in 1,4,0,0 2,4,0,0 2,4,2,0; do
  timeout -k 10 200 python tools/diag_patch.py T1 fwd $tile 2>/dev/null | grep -v amdgpu.ids >> $L || exit 1
done
timeout -k 10 300 python tools/time_t1.py >> $L 2>&1 || exit 1
cat $L
