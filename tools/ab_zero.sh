set -e
for Z in "" 1; do for G in bf16x3 f16x2; do
  echo "== zero=$Z CSTP_GEMM=$G s9x"
  CSTP_TIME_ZERO=$Z CSTP_GEMM=$G CSTP_TILE=s9x timeout -k 10 200 python tools/time_k1.py --only c2.same.S --iters 20 2>&1 | grep -v amdgpu.ids | cut -c1-120
done; done
