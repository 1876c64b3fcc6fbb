#!/bin/bash
# PMC passes (one rocprofv3 run per counter set) on one bf16-storage convolution layer of the 3D-ResNet-50 step: forward
# (conv_b16_kernel), data gradient and weight gradient (wgrad_b16_kernel) launches of tools/one_conv_b16.py.
# usage: tools/pmc_b16.sh [layer]      -> prints per-kernel counter averages (last launches)
layer=${1:-L1_3x3x3}
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
out=$R/gpurun_out/pmc_b16_$layer
rm -rf $out; mkdir -p $out
cd /tmp; export TMPDIR=/tmp
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_VALU_MFMA_COEXEC_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD"; do
  tag=$(echo $set | tr ' ' '+')
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $out/$tag -- python3 $R/tools/one_conv_b16.py $layer > $out/$tag.log 2>&1
  f=$(ls $out/$tag/*/*counter_collection.csv 2>/dev/null | head -1)
  [ -z "$f" ] && { echo "$tag: no counter file"; tail -3 $out/$tag.log; continue; }
  python3 - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    k = r["Kernel_Name"]
    if "b16" not in k: continue
    acc[k.split("(")[0][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    for c, v in d.items():
        tail = v[-2:]
        print("%-72s %-28s n=%3d  last2 avg %.6g" % (k, c, len(v), sum(tail) / len(tail)))
PY
done
grep -h "^done" $out/FETCH_SIZE.log
rm -rf $out/*/
