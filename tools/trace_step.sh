#!/bin/bash
# kernel traces of the bench step (last 3 of 4 steps): default (overlapped streams) and serial (per-kernel costs)
#   usage (GPU box, repo root): bash tools/trace_step.sh <outdir-under-gpurun_out>
set -e
R=$PWD
O=$R/gpurun_out/$1
mkdir -p $O
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_overlap -o run -- python3 $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-extras > $O/trace_overlap.log 2>&1
CSTP_OVERLAP_WGRAD=0 CSTP_OVERLAP_TARGET=0 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_serial -o run -- python3 $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-extras > $O/trace_serial.log 2>&1
cd $R
python3 profiles/summarize.py --trace $O/trace_overlap/run_kernel_trace.csv 3 > $O/bench_last3steps_overlap.txt
python3 profiles/summarize.py --trace $O/trace_serial/run_kernel_trace.csv 3 > $O/bench_last3steps_serial.txt
cp $O/trace_overlap/run_kernel_stats.csv $O/bench_overlap_kernel_stats.csv
cp $O/trace_serial/run_kernel_stats.csv $O/bench_serial_kernel_stats.csv
rm -rf $O/trace_overlap $O/trace_serial
