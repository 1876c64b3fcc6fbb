mkdir -p gpurun_out/r5d; L=gpurun_out/r5d/ab_wgrad_late.log
echo "# bench.py --steps 20: CSTP_WGRAD_LATE=1 (temporal weight gradient behind the data gradient, beside the BatchNorm backward) vs default" > $L
for i in 1 2 3; do
  CSTP_WGRAD_LATE=1 CSTP_TUNE_TABLE_RO=1 timeout -k 10 300 python bench.py --no-cpu-baseline --no-extras --steps 20 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('late', round(d['ms_per_step'],3))" >> $L
  CSTP_TUNE_TABLE_RO=1 timeout -k 10 300 python bench.py --no-cpu-baseline --no-extras --steps 20 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('early', round(d['ms_per_step'],3))" >> $L
done
cat $L
