# decomposition of the S1 forward / data-gradient patch launches: which half (weight DMA, patch staging, products) sets the K-tile time
mkdir -p gpurun_out/r4b
for mode in fwd dgrad; do
for v in default k1 k2 k3 k4 k5 k6 k7; do
  [ -f build_var/$v.so ] || [ $v = default ] || continue
  if [ $v = default ]; then timeout -k 10 200 python tools/diag_patch.py S1 $mode 2>/dev/null | grep -v amdgpu.ids >> gpurun_out/r4b/diag.log || exit 1
  else CSTP_LIB_PATH=$PWD/build_var/$v.so timeout -k 10 200 python tools/diag_patch.py S1 $mode 2>/dev/null | grep -v amdgpu.ids >> gpurun_out/r4b/diag.log || exit 1; fi
done; done
cat gpurun_out/r4b/diag.log
