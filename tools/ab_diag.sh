set -e
for V in "" d6; do for T in s9 s4; do
  echo "== variant=$V $T"
  L=""; [ -n "$V" ] && L=$PWD/build_ab/$V.so
  CSTP_LIB_PATH=$L CSTP_TILE=$T timeout -k 10 200 python tools/time_k1.py --only .S 2>&1 | grep -v amdgpu.ids | grep "c2\|c3.same\|c4.same\|total" | cut -c1-130
done; done
