#!/bin/bash
# A/B harness: run a command once per library variant in build_ab/ (each in a fresh process).
# usage: tools/ab_run.sh <log-prefix> <cmd...>
pref=$1; shift
cp cstp_amd/lib/libcstp_hip.so /tmp/lib_orig.so
for v in build_ab/*.so; do
  name=$(basename $v .so)
  cp $v cstp_amd/lib/libcstp_hip.so
  echo "=== $name"
  "$@" > gpurun_out/${pref}_${name}.log 2>&1
  grep -E "c2.same|c3.same|c4.same|c5.same|c3.b1c1|per enc|ms_per_step" gpurun_out/${pref}_${name}.log | sed 's/.*"ms_per_step": \([0-9.]*\).*/ms_per_step \1/'
done
cp /tmp/lib_orig.so cstp_amd/lib/libcstp_hip.so
