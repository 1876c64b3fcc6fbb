#!/usr/bin/env python3
"""How long does the HOST need to enqueue one cfg2 training step (Python + ctypes + launches), next to how long the GPU needs
to run it?  If the two are close the step is launch-bound and shortening kernels buys nothing.
usage: python tools/cpu_enqueue_time.py [steps]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cstp_amd.ntxent import NTXentLoss  # noqa: E402
from cstp_amd.optim import FlatSGD  # noqa: E402
from cstp_amd.r21d_byol import R21DBYOL, layer_sizes_for_depth  # noqa: E402
from cstp_amd.synthetic import device_batch  # noqa: E402
from cstp_amd.train import LaggedScalars, PretrainStep  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
dev = torch.device("cuda", 0)
torch.manual_seed(1)
model = R21DBYOL(pretrain=True, layer_sizes=layer_sizes_for_depth(18)).cuda()
arenas = model.flatten_parameters()
model.train()
opt = FlatSGD(model.parameters(), lr=0.09, momentum=0.9, weight_decay=5e-4, arenas=arenas)
ntx = NTXentLoss(device=dev, batch_size=16, temperature=0.5, use_cosine_similarity=True)
step = PretrainStep(model, opt, (0.1, 1.0, 1.0, 0.0, 0.0), clip_grad_norm=True, ntxent=ntx, ntxent_weight=1.0)
x1, x2, lab = device_batch(16, 16, 112, dev, seed=1)
lagged = LaggedScalars(dev, 1)


def run(n, log=True):
    for _ in range(n):
        out = step(x1, x2, lab["spa"], lab["tem"], lab["pb"], lab["rot1"], lab["rot2"])
        if log:
            lagged.push(out)          # waits for the PREVIOUS step's scalars: the host runs at most one step ahead


run(4)
for log in (True, False):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(steps, log)
    t_enq = time.perf_counter() - t0          # the host is done enqueueing
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    print("%s: enqueue %.2f ms/step (host), complete %.2f ms/step (GPU drained): host share %.0f %%"
          % ("with the lagged log read" if log else "no host read at all    ", t_enq / steps * 1e3, t_all / steps * 1e3,
             100 * t_enq / t_all))
