#!/bin/bash
# same-box A/B of the bench step between library variants:  bash tools/ab_lib.sh <outdir> <variant.so|default> <variant.so|default> [rounds]
# (variants from tools/build_variant.sh; "default" = the in-tree library).  Every variant tunes its own tile table first.
# Prints ms_per_step per run.
O=gpurun_out/$1; mkdir -p $O
A=$2; B=$3; R=${4:-3}
run() {  # name lib table steps
  if [ "$2" = default ]; then CSTP_TUNE_TABLE=$O/$3.json timeout -k 10 300 python bench.py --no-cpu-baseline --no-extras --steps $4 > $O/$1.log 2>&1
  else CSTP_TUNE_TABLE=$O/$3.json CSTP_LIB_PATH=$2 timeout -k 10 300 python bench.py --no-cpu-baseline --no-extras --steps $4 > $O/$1.log 2>&1; fi
}
run warm_a $A table_a 3 || exit 1
run warm_b $B table_b 3 || exit 1
for i in $(seq 1 $R); do run a$i $A table_a 20 || exit 1; run b$i $B table_b 20 || exit 1; done
python3 - <<PY
import json,glob
for f in sorted(glob.glob("$O/[ab][0-9]*.log")):
    for l in open(f):
        if l.startswith("{"):
            d=json.loads(l); print(f, round(d["ms_per_step"],3), d["tuned_tiles"]["from_table"])
PY
