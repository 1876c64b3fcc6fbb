mkdir -p gpurun_out/r5h
timeout -k 10 900 python -m pytest tests/test_b16_gpu.py tests/test_r3d_gpu.py -x -q -m gpu > gpurun_out/r5h/tests.log 2>&1; rc=$?; tail -3 gpurun_out/r5h/tests.log; [ $rc -eq 0 ] || exit $rc
L=gpurun_out/r5h/ab_r3d_gradjoin.log
echo "# tools/bench_r3d.py --depth 50 --batch 4 --size 224 --steps 10, one box, alternating; join = the residual connections' two gradients summed in the kernels' epilogues (ops.GradJoin), nojoin = CSTP_R3D_JOIN=0 (autograd's add pass)" > $L
for dt in bf16 fp32; do
for i in 1 2 3; do
  timeout -k 10 300 python tools/bench_r3d.py --depth 50 --batch 4 --size 224 --steps 10 --act_dtype $dt 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$dt join  ', d['ms_per_step'], d['clips_per_s'])" >> $L
  CSTP_R3D_JOIN=0 timeout -k 10 300 python tools/bench_r3d.py --depth 50 --batch 4 --size 224 --steps 10 --act_dtype $dt 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$dt nojoin', d['ms_per_step'], d['clips_per_s'])" >> $L
done
done
cat $L
