#!/bin/bash
# usage: pmc_var.sh <lib.so> <mode> <layer> [tile]
R=$PWD; out=$R/gpurun_out/pmcvar; rm -rf $out; mkdir -p $out
export CSTP_LIB_PATH=$1
cd /tmp; export TMPDIR=/tmp
for set in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_VALU_MFMA_COEXEC_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_MFMA"; do
  tag=$(echo $set | tr ' ' '+')
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $out/$tag -- python3 $R/tools/one_conv.py $2 32 $3 $4 > $out/$tag.log 2>&1
  f=$(ls $out/$tag/*/*counter_collection.csv 2>/dev/null | head -1)
  python3 - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    k = r["Kernel_Name"]
    if "igemm_k1p" not in k and "igemm_k2p" not in k: continue
    acc[k.split("(")[0][:40]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    for c, v in d.items():
        tail = v[-4:]
        print("%-42s %-28s %.5g" % (k, c, sum(tail) / len(tail)))
PY
done
