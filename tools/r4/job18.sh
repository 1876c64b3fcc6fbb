# same-box A/B of ops.PackPlan (CSTP_PACK_PLAN=0 packs inside every call, as in round 3); output kept as profiles/r04/ab_packplan.log
mkdir -p gpurun_out/r4q; L=gpurun_out/r4q/ab_packplan.log
echo "# bench.py --no-cpu-baseline --no-extras --steps 20 (R18 cfg2) / --steps 10 (R34), one box, alternating, tuned table read-only" > $L
echo "# columns: variant, [config,] ms/step, [S1 forward ms per launch by HIP events]" >> $L
bash tools/r4/job17.sh >> $L 2>&1
cat $L
