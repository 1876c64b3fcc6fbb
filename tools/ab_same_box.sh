#!/bin/bash
# same-box A/B of the bench step: the in-tree library against another build (CSTP_LIB_PATH), alternating, the in-tree tile table
# read-only for both.   bash tools/ab_same_box.sh <outdir> <other.so> [rounds] [steps]
O=gpurun_out/$1; mkdir -p $O; B=$2; R=${3:-3}; S=${4:-20}
export CSTP_TUNE_TABLE_RO=1
one() { python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', round(d['ms_per_step'],3), d['tuned_tiles']['from_table'], d['tuned_tiles']['timed'])"; }
for i in $(seq 1 $R); do
  python3 bench.py --no-cpu-baseline --no-extras --steps $S 2>/dev/null | tail -1 | one new | tee -a $O/ab.log
  CSTP_LIB_PATH=$B python3 bench.py --no-cpu-baseline --no-extras --steps $S 2>/dev/null | tail -1 | one old | tee -a $O/ab.log
done
