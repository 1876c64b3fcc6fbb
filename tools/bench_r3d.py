#!/usr/bin/env python3
"""Pre-training throughput of the 3D-ResNet-BYOL wrapper on one MI355X (synthetic clips resident in HBM): the per-GPU share of
BASELINE.json configs[4] (B = 32 over 8 GPUs -> 4 clip pairs per GPU, 3x16x224x224) on the BasicBlock depths the reference can
can run (10 / 18 / 34) and on the Bottleneck depth 50 that configs[4] names (corrected wrapper: cstp_amd/r3d_byol.py), with fp32 or
bf16 activation storage (--act_dtype; configs[4] says bf16).

    python tools/bench_r3d.py --depth 50 --batch 4 --size 224 --steps 10 --act_dtype bf16

Not the headline metric (bench.py is) -- a sizing aid for the backbone swap."""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from cstp_amd.optim import FlatSGD  # noqa: E402
from cstp_amd.r3d_byol import R3DBYOL  # noqa: E402
from cstp_amd.synthetic import device_batch  # noqa: E402
from cstp_amd.train import PretrainStep  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--depth", type=int, default=18)
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--frames", type=int, default=16)
    ap.add_argument("--size", type=int, default=224)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--act_dtype", default="fp32", choices=("fp32", "bf16"))
    a = ap.parse_args()
    torch.manual_seed(1)
    dev = torch.device("cuda", 0)
    opts = argparse.Namespace(model_depth=a.depth, sample_size=a.size, sample_duration=a.frames, sc_type="B", n_classes=400, act_dtype=a.act_dtype)
    model = R3DBYOL(pretrain=True, opts=opts).cuda()
    arenas = model.flatten_parameters()
    model.train()
    opt = FlatSGD(model.parameters(), lr=0.01, momentum=0.9, weight_decay=5e-4, arenas=arenas)
    step = PretrainStep(model, opt, (0.1, 1.0, 1.0, 1.0, 1.0), clip_grad_norm=True)
    x1, x2, lab = device_batch(a.batch, a.frames, a.size, dev, seed=1)

    def run(n):
        for _ in range(n):
            step(x1, x2, lab["spa"], lab["tem"], lab["pb"], lab["rot1"], lab["rot2"]).to_host()
    run(1 + a.warmup)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(a.steps)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / a.steps * 1e3
    print(json.dumps({"config": {"workload": "r3d_byol 3D-ResNet-%d, B=%d clip pairs 3x%dx%dx%d, full loss_com, clip 18, SGD; %s activation storage"
                                 % (a.depth, a.batch, a.frames, a.size, a.size, a.act_dtype)},
                      "ms_per_step": round(ms, 2), "clips_per_s": round(a.batch / ms * 1e3, 2),
                      "max_mem_GiB": round(torch.cuda.max_memory_allocated() / 2 ** 30, 2)}))


if __name__ == "__main__":
    main()
