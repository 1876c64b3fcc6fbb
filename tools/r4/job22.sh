mkdir -p gpurun_out/r4v
timeout -k 10 1000 python -m pytest tests/ -x -q -m gpu > gpurun_out/r4v/tests.log 2>&1; rc=$?; tail -4 gpurun_out/r4v/tests.log; [ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python tools/step_table.py > gpurun_out/r4v/table18.log 2>&1
CSTP_TUNE_TABLE_RO=1 timeout -k 10 300 python bench.py --no-cpu-baseline --no-extras --steps 20 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('bench', round(d['ms_per_step'],3), d['roofline']['frac'], d['roofline']['avg_launch_ms'])"
