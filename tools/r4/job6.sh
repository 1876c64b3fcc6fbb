# in-step A/B: T1 forward on the gather kernel (tuner's pick) against igemm_k1w pinned through the tile table
mkdir -p gpurun_out/r4f
python3 - <<'PY'
import json
d=json.load(open('tools/r4/data/table_r4e.json'))
d['tiles']['2|0|32,144,16,56,56,64,3,1,1,1,1,1,1,0,0']=[2,4,2,1]
json.dump(d,open('gpurun_out/r4f/table_k1w.json','w'))
PY
cp tools/r4/data/table_r4e.json gpurun_out/r4f/table_gather.json
bash tools/ab_table.sh r4f $PWD/gpurun_out/r4f/table_gather.json $PWD/gpurun_out/r4f/table_k1w.json 3
