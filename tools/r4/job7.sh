mkdir -p gpurun_out/r4g
python -m pytest tests -m gpu -x -q > gpurun_out/r4g/tests.log 2>&1 || { tail -40 gpurun_out/r4g/tests.log; exit 1; }
tail -3 gpurun_out/r4g/tests.log
python bench.py --no-cpu-baseline --no-extras --steps 20 > gpurun_out/r4g/bench.log 2>&1
python3 -c "
import json
for l in open('gpurun_out/r4g/bench.log'):
    if l.startswith('{'):
        d=json.loads(l); print(round(d['ms_per_step'],3), d['tuned_tiles'], d['roofline']['frac'], d['roofline']['avg_launch_ms'])
"
