#!/bin/bash
# PMC passes (one rocprofv3 run per counter set) on the S1-shape forward convolution; prints the per-dispatch
# counter values of the convolution kernel the autotuner picked (or the pinned one).
# usage: tools/pmc_s1.sh [fwd|dgrad|wgrad] [layer: S1 T1 S3 T3 S5 T5] [tile=sp,mt,x,y]     (see tools/one_conv.py)
mode=${1:-fwd}; layer=${2:-S1}; pin=$3
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
out=$R/gpurun_out/pmc_${mode}_${layer}
rm -rf $out; mkdir -p $out
cd /tmp; export TMPDIR=/tmp
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "TCC_HIT_sum TCC_MISS_sum" "SQ_VALU_MFMA_COEXEC_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM_RD" "TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum"; do
  tag=$(echo $set | tr ' ' '+')
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $out/$tag -- python3 $R/tools/one_conv.py $mode 32 $layer $pin > $out/$tag.log 2>&1
  f=$(ls $out/$tag/*/*counter_collection.csv 2>/dev/null | head -1)
  [ -z "$f" ] && { echo "$tag: no counter file"; tail -3 $out/$tag.log; continue; }
  python3 - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    k = r["Kernel_Name"]
    if "igemm" not in k: continue
    acc[k.split("(")[0][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    for c, v in d.items():
        tail = v[-4:]                       # the last launches (after the autotune candidates)
        print("%-62s %-28s n=%3d  last4 avg %.6g" % (k, c, len(v), sum(tail) / len(tail)))
PY
done
rm -rf $out/*/
