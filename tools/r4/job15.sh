mkdir -p gpurun_out/r4o
for i in 1 2; do
  for cfg in "--depth 34" "--depth 34 --batch 8 --frames 32"; do
    CSTP_TUNE_TABLE_RO=1 timeout -k 10 300 python bench.py $cfg --no-cpu-baseline --no-extras --steps 10 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('plan  ', '$cfg', round(d['ms_per_step'],3))"
    CSTP_PACK_PLAN=0 CSTP_TUNE_TABLE_RO=1 timeout -k 10 300 python bench.py $cfg --no-cpu-baseline --no-extras --steps 10 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('noplan', '$cfg', round(d['ms_per_step'],3))"
  done
done
