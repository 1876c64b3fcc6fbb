#!/bin/bash
# usage: ab.sh <outdir> ; runs bench alternately unfused/fused, prints ms_per_step
O=gpurun_out/$1; mkdir -p $O
timeout -k 10 300 python bench.py --no-cpu-baseline --no-extras --steps 3 --warmup 2 > $O/warm.log 2>&1
for i in 1 2 3; do
  CSTP_FUSE_BN_T=0 timeout -k 10 300 python bench.py --no-cpu-baseline --no-extras --steps 20 > $O/u$i.log 2>&1 || exit 1
  timeout -k 10 300 python bench.py --no-cpu-baseline --no-extras --steps 20 > $O/f$i.log 2>&1 || exit 1
done
python3 - <<PY
import json,glob
for f in sorted(glob.glob("$O/[uf]*.log")):
    for l in open(f):
        if l.startswith("{"):
            d=json.loads(l); print(f, round(d["ms_per_step"],3), d["tuned_tiles"]["from_table"])
PY
