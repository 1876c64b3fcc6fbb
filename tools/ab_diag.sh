set -e
for G in bf16x3 f16x2; do
  echo "== CSTP_GEMM=$G s9x"
  CSTP_GEMM=$G CSTP_TILE=s9x timeout -k 10 200 python tools/time_k1.py 2>&1 | grep -v amdgpu.ids | cut -c1-150
done
timeout -k 10 600 python -m pytest tests/test_split_gpu.py -x -q 2>&1 | tail -3
