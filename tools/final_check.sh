#!/bin/bash
# end-of-round verification on the GPU box: full GPU suite, smoke(), the secondary benches, the launcher path at world 1
set -e
O=gpurun_out/final
mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1
tail -1 $O/gpu_tests.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout -k 10 300 python tools/bench_ft.py > $O/bench_ft.log 2>&1; tail -4 $O/bench_ft.log | cut -c1-250
timeout -k 10 300 python tools/bench_r3d.py > $O/bench_r3d.log 2>&1; tail -4 $O/bench_r3d.log | cut -c1-250
timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_ddp1.log 2>&1; grep "^{" $O/bench_ddp1.log | cut -c1-200
