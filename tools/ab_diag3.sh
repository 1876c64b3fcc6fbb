set -e
for V in base diag3; do
  for T in "" s9x; do
    echo "== $V CSTP_TILE=$T"
    CSTP_LIB_PATH=$PWD/build_ab/$V.so CSTP_TILE=$T timeout -k 10 200 python tools/time_k1.py 2>&1 | grep -v amdgpu.ids | cut -c1-120
  done
done
