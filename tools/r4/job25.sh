mkdir -p gpurun_out/r5a
timeout -k 10 900 python -m pytest tests/test_split_gpu.py tests/test_ops_gpu.py tests/test_fused_bn_gpu.py tests/test_tpatch_gpu.py -x -q -m gpu > gpurun_out/r5a/tests.log 2>&1; rc=$?; tail -3 gpurun_out/r5a/tests.log; [ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python tools/step_table.py > gpurun_out/r5a/table18.log 2>&1; head -12 gpurun_out/r5a/table18.log | cut -c1-110
L=gpurun_out/r5a/ab_k1p_wide.log
echo "# bench.py --no-cpu-baseline --no-extras --steps 20 (R18 cfg2) / --steps 10 (R34), one box, alternating, tuned table read-only" > $L
echo "# new = in-tree library (64-row patch launches on six staging waves, 768 threads); old = CSTP_K1P_WIDE=0 (two staging waves, 512 threads)" >> $L
for i in 1 2 3; do
  CSTP_TUNE_TABLE_RO=1 timeout -k 10 300 python bench.py --no-cpu-baseline --no-extras --steps 20 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('new', round(d['ms_per_step'],3))" >> $L
  CSTP_K1P_WIDE=0 CSTP_TUNE_TABLE_RO=1 timeout -k 10 300 python bench.py --no-cpu-baseline --no-extras --steps 20 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('old', round(d['ms_per_step'],3))" >> $L
done
for cfg in "--depth 34" "--depth 34 --batch 8 --frames 32"; do
    CSTP_TUNE_TABLE_RO=1 timeout -k 10 300 python bench.py $cfg --no-cpu-baseline --no-extras --steps 10 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('new', '$cfg', round(d['ms_per_step'],3))" >> $L
    CSTP_K1P_WIDE=0 CSTP_TUNE_TABLE_RO=1 timeout -k 10 300 python bench.py $cfg --no-cpu-baseline --no-extras --steps 10 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('old', '$cfg', round(d['ms_per_step'],3))" >> $L
done
cat $L
