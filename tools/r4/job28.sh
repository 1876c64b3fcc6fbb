mkdir -p gpurun_out/r5f
timeout -k 10 600 python -m pytest tests/test_b16_gpu.py -x -q -m gpu > gpurun_out/r5f/tests.log 2>&1; rc=$?; tail -3 gpurun_out/r5f/tests.log; [ $rc -eq 0 ] || exit $rc
L=gpurun_out/r5f/ab_bn_small_b16.log
echo "# tools/bench_r3d.py --depth 50 --batch 4 --size 224 --steps 10 --act_dtype bf16, one box, alternating; new = single-launch BatchNorm on small bf16 tensors, old = CSTP_BN_SMALL=0" > $L
for i in 1 2 3; do
  timeout -k 10 300 python tools/bench_r3d.py --depth 50 --batch 4 --size 224 --steps 10 --act_dtype bf16 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('new', d['ms_per_step'], d['clips_per_s'])" >> $L
  CSTP_BN_SMALL=0 timeout -k 10 300 python tools/bench_r3d.py --depth 50 --batch 4 --size 224 --steps 10 --act_dtype bf16 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('old', d['ms_per_step'], d['clips_per_s'])" >> $L
done
cat $L
