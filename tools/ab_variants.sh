#!/bin/bash
# A/B kernel-library variants from build_var/ on the patch-kernel layer shapes, plus the WRITE_SIZE counter of the S1 forward
# launch per variant.  usage (on the GPU box): tools/ab_variants.sh "S1 S3" name1 name2 ...   -> gpurun_out/abv/
layers=$1; shift
R=$PWD; mkdir -p gpurun_out/abv
for v in "$@"; do CSTP_LIB_PATH=$R/build_var/$v.so timeout -k 10 180 python tools/ab_patch.py $layers > gpurun_out/abv/ab_$v.log 2>&1 || exit 1; done
cd /tmp; export TMPDIR=/tmp
for v in "$@"; do
  export CSTP_LIB_PATH=$R/build_var/$v.so
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/abv/w_$v -- python3 $R/tools/one_conv.py fwd 32 > $R/gpurun_out/abv/w_$v.log 2>&1
done
cd $R
python3 - "$@" <<'PY'
import csv, glob, sys
for v in sys.argv[1:]:
    f = glob.glob("gpurun_out/abv/w_%s/*/*counter_collection.csv" % v)
    if f:
        rows = [float(r["Counter_Value"]) for r in csv.DictReader(open(f[0])) if "k1p" in r["Kernel_Name"]][-4:]
        print("%-8s WRITE_SIZE KB (last launches) %s" % (v, ["%.0f" % x for x in rows]))
    for line in open("gpurun_out/abv/ab_%s.log" % v):
        if "patch" in line: print("%-8s %s" % (v, line.rstrip()))
PY
rm -rf gpurun_out/abv/w_*/
