# serial per-kernel trace of the bench step at HEAD + PMC on T1 weight gradient (k2t, AFF as the step runs it is timed in the trace)
mkdir -p gpurun_out/r4l
O=$GRAFT_REPO_ROOT/gpurun_out/r4l
cd /tmp; export TMPDIR=/tmp
CSTP_TUNE_TABLE_RO=1 CSTP_OVERLAP_WGRAD=0 CSTP_OVERLAP_TARGET=0 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_serial -o run -- python3 $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-extras > $O/trace_serial.log 2>&1
cd $GRAFT_REPO_ROOT
python3 profiles/summarize.py --trace $O/trace_serial/run_kernel_trace.csv 3 > $O/bench_last3steps_serial.txt
cp $O/trace_serial/run_kernel_stats.csv $O/bench_serial_kernel_stats.csv
rm -rf $O/trace_serial
head -60 $O/bench_last3steps_serial.txt
