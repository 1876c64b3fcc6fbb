mkdir -p gpurun_out/r4u; L=gpurun_out/r4u/ab_ksplit.log
echo "# bench.py --no-cpu-baseline --no-extras --steps 20 (R18 cfg2) / --steps 10 (R34), one box, alternating, tuned table read-only" > $L
echo "# new = in-tree library; old = CSTP_KSPLIT=0 (one split of the reduction axis in igemm_k1s)" >> $L
for i in 1 2 3; do
  CSTP_TUNE_TABLE_RO=1 timeout -k 10 300 python bench.py --no-cpu-baseline --no-extras --steps 20 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('new', round(d['ms_per_step'],3))" >> $L
  CSTP_KSPLIT=0 CSTP_TUNE_TABLE_RO=1 timeout -k 10 300 python bench.py --no-cpu-baseline --no-extras --steps 20 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('old', round(d['ms_per_step'],3))" >> $L
done
for cfg in "--depth 34" "--depth 34 --batch 8 --frames 32"; do
    CSTP_TUNE_TABLE_RO=1 timeout -k 10 300 python bench.py $cfg --no-cpu-baseline --no-extras --steps 10 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('new', '$cfg', round(d['ms_per_step'],3))" >> $L
    CSTP_KSPLIT=0 CSTP_TUNE_TABLE_RO=1 timeout -k 10 300 python bench.py $cfg --no-cpu-baseline --no-extras --steps 10 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('old', '$cfg', round(d['ms_per_step'],3))" >> $L
done
cat $L
