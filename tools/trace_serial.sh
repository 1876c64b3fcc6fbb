#!/bin/bash
# serial kernel trace of the bench step (last 3 of 4 steps) for one library variant:
#   usage (GPU box, repo root): bash tools/trace_serial.sh <outdir-under-gpurun_out> [lib.so]
set -e
R=$PWD
O=$R/gpurun_out/$1
mkdir -p $O
[ -n "$2" ] && export CSTP_LIB_PATH=$R/$2
cd /tmp; export TMPDIR=/tmp
CSTP_OVERLAP_WGRAD=0 CSTP_OVERLAP_TARGET=0 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_serial -o run -- python3 $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-extras > $O/trace_serial.log 2>&1
cd $R
python3 profiles/summarize.py --trace $O/trace_serial/run_kernel_trace.csv 3 > $O/bench_last3steps_serial.txt
rm -rf $O/trace_serial
