#!/bin/bash
# same-box A/B of the bench step between two tile tables (read-only):  bash tools/ab_table.sh <outdir> <tableA.json> <tableB.json> [rounds]
O=gpurun_out/$1; mkdir -p $O
R=${4:-3}
for i in $(seq 1 $R); do
  CSTP_TUNE_TABLE=$2 CSTP_TUNE_TABLE_RO=1 timeout -k 10 300 python bench.py --no-cpu-baseline --no-extras --steps 20 > $O/a$i.log 2>&1 || exit 1
  CSTP_TUNE_TABLE=$3 CSTP_TUNE_TABLE_RO=1 timeout -k 10 300 python bench.py --no-cpu-baseline --no-extras --steps 20 > $O/b$i.log 2>&1 || exit 1
done
python3 - <<PY
import json,glob
for f in sorted(glob.glob("$O/[ab][0-9]*.log")):
    for l in open(f):
        if l.startswith("{"):
            d=json.loads(l); print(f, round(d["ms_per_step"],3), d["tuned_tiles"])
PY
