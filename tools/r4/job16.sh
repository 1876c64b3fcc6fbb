mkdir -p gpurun_out/r4p
O=$GRAFT_REPO_ROOT/gpurun_out/r4p
cd /tmp; export TMPDIR=/tmp
for d in 18 34; do
CSTP_TUNE_TABLE_RO=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/t$d -o run -- python3 $GRAFT_REPO_ROOT/tools/r4/plan_stats.py $d > $O/t$d.log 2>&1
grep -E "pack_|Name" $O/t$d/run_kernel_stats.csv | cut -c1-160
done
rm -rf $O/t18 $O/t34
