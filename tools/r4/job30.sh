mkdir -p gpurun_out/r5j; L=gpurun_out/r5j/ab_env.log
echo "# bench.py --steps 20, one box, alternating: HIP_FORCE_DEV_KERNARG=1 vs default; then GPU_MAX_HW_QUEUES=8 vs default" > $L
one() { python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', round(d['ms_per_step'],3))"; }
for i in 1 2 3; do
  HIP_FORCE_DEV_KERNARG=1 CSTP_TUNE_TABLE_RO=1 timeout -k 10 300 python bench.py --no-cpu-baseline --no-extras --steps 20 2>/dev/null | tail -1 | one kernarg >> $L
  CSTP_TUNE_TABLE_RO=1 timeout -k 10 300 python bench.py --no-cpu-baseline --no-extras --steps 20 2>/dev/null | tail -1 | one default >> $L
done
for i in 1 2; do
  GPU_MAX_HW_QUEUES=8 CSTP_TUNE_TABLE_RO=1 timeout -k 10 300 python bench.py --no-cpu-baseline --no-extras --steps 20 2>/dev/null | tail -1 | one hwq8 >> $L
  GPU_MAX_HW_QUEUES=2 CSTP_TUNE_TABLE_RO=1 timeout -k 10 300 python bench.py --no-cpu-baseline --no-extras --steps 20 2>/dev/null | tail -1 | one hwq2 >> $L
done
cat $L
