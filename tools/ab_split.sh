#!/bin/bash
# A/B of the native-f32 tiles vs the 3xbf16-split tiles on chosen layers, then the conv parity tests under forced split tiles
set -e
O=gpurun_out/split
mkdir -p $O
rm -f $O/time.log
for T in ${TILES:-"" s9 s8 s4}; do
  echo "== CSTP_TILE=$T" >> $O/time.log
  CSTP_TILE=$T timeout -k 10 300 python tools/time_k1.py --only "${ONLY:-c2}" 2>&1 | grep -v amdgpu.ids >> $O/time.log
done
cat $O/time.log
if [ -z "$NOTEST" ]; then
for T in s9 s8 s4; do
  echo "== tests with CSTP_TILE=$T"
  CSTP_TILE=$T CSTP_AUTOTUNE=0 timeout -k 10 300 python -m pytest tests/test_ops_gpu.py -x -q -k "conv3d or linear" 2>&1 | tail -3
done
fi
