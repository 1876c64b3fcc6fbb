# cfg5 evidence: the bf16-storage 3D-ResNet-50 step with its roofline block, PMC passes on three bf16 layers, and the golden test that changed
mkdir -p gpurun_out/r4h
python tools/bench_r3d.py --depth 50 --batch 4 --size 224 --steps 10 --act_dtype bf16 > gpurun_out/r4h/bench_r3d50_bf16.json.log 2>gpurun_out/r4h/bench_r3d50.err || { tail -20 gpurun_out/r4h/bench_r3d50.err; exit 1; }
python3 -c "
import json
d=json.loads(open('gpurun_out/r4h/bench_r3d50_bf16.json.log').read().strip().splitlines()[-1])
print(d['ms_per_step'], d['clips_per_s']); r=d['roofline']; print({k:r[k] for k in r if k!='classes'})
for c in r['classes'][:12]: print(c)
"
for l in L1_3x3x3 L1_pw_out L2_3x3x3; do bash tools/pmc_b16.sh $l > gpurun_out/r4h/pmc_b16_$l.txt 2>&1; tail -30 gpurun_out/r4h/pmc_b16_$l.txt; done
python -m pytest tests/test_model_gpu.py -x -q -k "r34" > gpurun_out/r4h/tests_r34.log 2>&1 || { tail -30 gpurun_out/r4h/tests_r34.log; exit 1; }
tail -3 gpurun_out/r4h/tests_r34.log
