mkdir -p gpurun_out/r4n
timeout -k 10 600 python -m pytest tests/test_model_gpu.py -x -q -k "pack_plan" > gpurun_out/r4n/tests.log 2>&1 || { tail -50 gpurun_out/r4n/tests.log; exit 1; }
tail -2 gpurun_out/r4n/tests.log
for i in 1 2; do
  CSTP_TUNE_TABLE_RO=1 timeout -k 10 300 python bench.py --no-cpu-baseline --no-extras --steps 20 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('plan', round(d['ms_per_step'],3), d['roofline']['avg_launch_ms'])"
  CSTP_PACK_PLAN=0 CSTP_TUNE_TABLE_RO=1 timeout -k 10 300 python bench.py --no-cpu-baseline --no-extras --steps 20 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('noplan', round(d['ms_per_step'],3), d['roofline']['avg_launch_ms'])"
done
