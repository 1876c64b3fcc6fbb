mkdir -p gpurun_out/r4j
timeout -k 10 300 python -m pytest tests/test_tpatch_gpu.py -x -q -k "stream" > gpurun_out/r4j/tests.log 2>&1 || { tail -40 gpurun_out/r4j/tests.log; exit 1; }
tail -2 gpurun_out/r4j/tests.log
O=gpurun_out/r4j
for i in 1 2; do
  for t in base18 k2t k2t_stem; do
    CSTP_TUNE_TABLE=$PWD/tools/r4/data/table_$t.json CSTP_TUNE_TABLE_RO=1 timeout -k 10 300 python bench.py --no-cpu-baseline --no-extras --steps 20 > $O/${t}_$i.log 2>&1 || exit 1
  done
done
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r4j/*_[12].log")):
    for l in open(f):
        if l.startswith("{"):
            d=json.loads(l); print(f, round(d["ms_per_step"],3), d["tuned_tiles"]["from_table"], d["tuned_tiles"]["timed"])
PY
