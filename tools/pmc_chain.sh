#!/bin/bash
# PMC passes (one rocprofv3 run per counter set) over tools/one_chain_t1.py: the S1 / T1 kernels in the instantiations the
# training step runs (BatchNorm sums in the epilogues, the BatchNorm in front of T1 inside its staging).
# usage: tools/pmc_chain.sh          -> per-kernel counter averages (last launches) on stdout
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$PWD
out=$R/gpurun_out/pmc_chain
rm -rf $out; mkdir -p $out
cd /tmp; export TMPDIR=/tmp
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "TCC_HIT_sum TCC_MISS_sum" "SQ_VALU_MFMA_COEXEC_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD"; do
  tag=$(echo $set | tr ' ' '+')
  CSTP_TUNE_TABLE_RO=1 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $out/$tag -- python3 $R/tools/one_chain_t1.py > $out/$tag.log 2>&1
  f=$(ls $out/$tag/*/*counter_collection.csv 2>/dev/null | head -1)
  [ -z "$f" ] && { echo "$tag: no counter file"; tail -3 $out/$tag.log; continue; }
  python3 - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    k = r["Kernel_Name"]
    if "igemm" not in k and "bn_" not in k: continue
    acc[k.split("(")[0][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    for c, v in d.items():
        tail = v[-2:]
        print("%-62s %-28s n=%3d  last2 avg %.6g" % (k, c, len(v), sum(tail) / len(tail)))
PY
done
rm -rf $out/*/
