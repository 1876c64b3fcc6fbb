# PMC on T1 forward (gather vs weight-resident), then a freshly tuned bench run (new tile table) and the per-kernel serial trace
mkdir -p gpurun_out/r4e
bash tools/pmc_s1.sh fwd T1 tile=1,4,0,0 > gpurun_out/r4e/pmc_T1_fwd_gather.txt 2>&1
bash tools/pmc_s1.sh fwd T1 tile=2,4,2,0 > gpurun_out/r4e/pmc_T1_fwd_k1w.txt 2>&1
CSTP_TUNE_TABLE=$PWD/gpurun_out/r4e/table.json timeout -k 10 900 python bench.py --no-cpu-baseline --no-extras --steps 20 > gpurun_out/r4e/bench_tune.log 2>&1 || { tail -5 gpurun_out/r4e/bench_tune.log; exit 1; }
CSTP_TUNE_TABLE=$PWD/gpurun_out/r4e/table.json CSTP_TUNE_TABLE_RO=1 timeout -k 10 300 python bench.py --no-cpu-baseline --no-extras --steps 20 > gpurun_out/r4e/bench2.log 2>&1
python3 -c "
import json
for f in ('bench_tune','bench2'):
    for l in open('gpurun_out/r4e/%s.log'%f):
        if l.startswith('{'):
            d=json.loads(l); print(f, round(d['ms_per_step'],3), d['tuned_tiles'])
"
cd /tmp; export TMPDIR=/tmp
CSTP_TUNE_TABLE=$GRAFT_REPO_ROOT/gpurun_out/r4e/table.json CSTP_TUNE_TABLE_RO=1 CSTP_OVERLAP_WGRAD=0 CSTP_OVERLAP_TARGET=0 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r4e/trace_serial -o run -- python3 $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-extras > $GRAFT_REPO_ROOT/gpurun_out/r4e/trace_serial.log 2>&1
cd $GRAFT_REPO_ROOT
python3 profiles/summarize.py --trace gpurun_out/r4e/trace_serial/run_kernel_trace.csv 3 > gpurun_out/r4e/bench_last3steps_serial.txt
rm -rf gpurun_out/r4e/trace_serial
head -45 gpurun_out/r4e/bench_last3steps_serial.txt
cat gpurun_out/r4e/pmc_T1_fwd_gather.txt gpurun_out/r4e/pmc_T1_fwd_k1w.txt
