# the multi-rank control flow with this round's PackPlan / StagedAllReduce changes: world-2 rehearsal (gloo, both ranks on cuda:0, 8 steps so
# that the pack plan records and replays) and the driver's launch line at world 1 next to a plain run
mkdir -p gpurun_out/r4x
timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 tools/rehearse_world2.py 18 4 8 > gpurun_out/r4x/world2_rehearsal.log 2>&1; echo "world2 rc=$?"; tail -4 gpurun_out/r4x/world2_rehearsal.log
CSTP_TUNE_TABLE_RO=1 timeout -k 10 300 python bench.py --no-cpu-baseline --no-extras --steps 20 > gpurun_out/r4x/bench_world1_plain.json.log 2>/dev/null; echo "plain rc=$?"
CSTP_TUNE_TABLE_RO=1 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29518 bench.py --gpus 1 --no-cpu-baseline --no-extras --steps 20 > gpurun_out/r4x/bench_world1_torchrun.json.log 2>gpurun_out/r4x/torchrun.err; echo "torchrun rc=$?"
python3 - <<'PY'
import json
for f in ("plain", "torchrun"):
    d = json.loads(open("gpurun_out/r4x/bench_world1_%s.json.log" % f).read().strip().splitlines()[-1])
    print(f, round(d["ms_per_step"], 3), round(d["value"], 2), d["n_gpus"])
PY
