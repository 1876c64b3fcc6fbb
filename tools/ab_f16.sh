#!/bin/bash
set -e
O=gpurun_out/f16
mkdir -p $O
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > $O/full_gpu.log 2>&1
tail -4 $O/full_gpu.log
