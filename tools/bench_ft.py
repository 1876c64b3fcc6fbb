#!/usr/bin/env python3
"""Fine-tune / validation throughput of R21DBYOL(pretrain=False) on one MI355X (synthetic clips resident in HBM).

    python tools/bench_ft.py --depth 18 --batch 16 --steps 10

Prints ms/step and clips/s for: ft_all training step, ft_fc training step (frozen encoder: forward only + classifier
backward) and model.eval() validation forward.  Not the headline metric (bench.py is) -- a sizing aid for the
fine-tune path."""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from cstp_amd import ops  # noqa: E402
from cstp_amd.optim import FlatSGD  # noqa: E402
from cstp_amd.r21d_byol import R21DBYOL, get_fine_tuning_parameters, layer_sizes_for_depth  # noqa: E402
from cstp_amd.train import FineTuneStep  # noqa: E402


def timed(fn, steps, warmup):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--depth", type=int, default=18)
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--frames", type=int, default=16)
    ap.add_argument("--size", type=int, default=112)
    ap.add_argument("--classes", type=int, default=101)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    a = ap.parse_args()
    torch.manual_seed(1)
    dev = torch.device("cuda", 0)
    x = torch.rand((a.batch, 3, a.frames, a.size, a.size), device=dev) * 2 - 1
    y = torch.randint(0, a.classes, (a.batch,), device=dev)
    out = {"config": {"workload": "R(2+1)D-%d fine-tune, B=%d, 3x%dx%dx%d, %d classes" % (a.depth, a.batch, a.frames, a.size,
                                                                                         a.size, a.classes)}}
    for task in ("ft_all", "ft_fc"):
        model = R21DBYOL(pretrain=False, num_classes=a.classes, cls_bn=True, layer_sizes=layer_sizes_for_depth(a.depth)).cuda()
        arenas = model.flatten_parameters()
        params = get_fine_tuning_parameters(model, 0 if task == "ft_all" else 5)
        opt = FlatSGD(params, lr=0.01, momentum=0.9, weight_decay=5e-4, arenas=arenas)
        step = FineTuneStep(model, opt, task)
        model.train()
        ms = timed(lambda: step(x, y), a.steps, a.warmup + 1)
        out[task] = {"ms_per_step": round(ms, 3), "clips_per_s": round(a.batch / ms * 1e3, 1)}
        if task == "ft_all":
            model.eval()
            with torch.no_grad():
                ms = timed(lambda: ops.cross_entropy(model(x, o_type="test"), y), a.steps, a.warmup)
            out["eval"] = {"ms_per_step": round(ms, 3), "clips_per_s": round(a.batch / ms * 1e3, 1)}
        del model, opt, step
    print(json.dumps(out))


if __name__ == "__main__":
    main()
