set -e
for T in s9x s9 s4; do
  echo "== CSTP_GEMM=f16x2 $T"
  CSTP_TILE=$T timeout -k 10 200 python tools/time_k1.py 2>&1 | grep -v amdgpu.ids | grep "c2.same\|total" | cut -c1-150
done
timeout -k 10 600 python -m pytest tests/test_split_gpu.py -x -q 2>&1 | tail -3
CSTP_DEBUG=1 timeout -k 10 400 python bench.py --steps 8 --warmup 3 --no-cpu-baseline 2>&1 | tail -2 | cut -c1-330
