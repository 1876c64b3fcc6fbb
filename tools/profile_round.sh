#!/bin/bash
# end-of-round evidence: kernel traces of the bench (default = overlapped streams; serial for per-kernel costs), the PMC passes
# on the S1 forward / weight-gradient kernels, and the default bench line.  Run from the repo root on the GPU box.
set -e
R=$PWD
O=$R/gpurun_out/round
rm -rf $O; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_overlap -o run -- python3 $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline > $O/trace_overlap.log 2>&1
CSTP_OVERLAP_WGRAD=0 CSTP_OVERLAP_TARGET=0 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_serial -o run -- python3 $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline > $O/trace_serial.log 2>&1
cd $R
python3 profiles/summarize.py --trace $O/trace_overlap/run_kernel_trace.csv 3 > $O/bench_last3steps_f16_overlap.txt
python3 profiles/summarize.py --trace $O/trace_serial/run_kernel_trace.csv 3 > $O/bench_last3steps_f16_serial.txt
cp $O/trace_overlap/run_kernel_stats.csv $O/bench_f16_overlap_kernel_stats.csv
cp $O/trace_serial/run_kernel_stats.csv $O/bench_f16_serial_kernel_stats.csv
rm -rf $O/trace_overlap $O/trace_serial
bash tools/pmc_s1.sh fwd > $O/pmc_fwd.txt 2>&1
bash tools/pmc_s1.sh wgrad > $O/pmc_wgrad.txt 2>&1
python3 bench.py > $O/bench_default_run.json.log 2>$O/bench_default_run.err
tail -1 $O/bench_default_run.json.log | cut -c1-300
head -12 $O/bench_last3steps_f16_serial.txt
