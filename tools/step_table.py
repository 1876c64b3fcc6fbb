#!/usr/bin/env python3
"""Where does the cfg2 training step spend its GPU time, call site by call site?  Every spanned C-ABI call (convolutions by
direction, BatchNorm forward / backward) of two steps under a HIP-event pair on its launch stream, stream overlaps off, summed per
(operation, shape) and printed with the call's ALGORITHMIC work (2*k*c*taps FLOP per output position; every operand touched once
at 4 B) as TFLOP/s and TB/s.  Fused forms are counted where they run: a BatchNorm whose apply runs in the next convolution's staging
shows up in that convolution's row.
usage: python tools/step_table.py [depth=18] [batch=16] [frames=16]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cstp_amd import ops, r21d_byol as rb  # noqa: E402
from cstp_amd.ntxent import NTXentLoss  # noqa: E402
from cstp_amd.optim import FlatSGD  # noqa: E402
from cstp_amd.r21d_byol import R21DBYOL, layer_sizes_for_depth  # noqa: E402
from cstp_amd.synthetic import device_batch  # noqa: E402
from cstp_amd.train import PretrainStep  # noqa: E402
from tools.bench_r3d import AllTimers  # noqa: E402

depth = int(sys.argv[1]) if len(sys.argv) > 1 else 18
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 16
frames = int(sys.argv[3]) if len(sys.argv) > 3 else 16
dev = torch.device("cuda", 0)
torch.manual_seed(1)
model = R21DBYOL(pretrain=True, layer_sizes=layer_sizes_for_depth(depth)).cuda()
arenas = model.flatten_parameters()
model.train()
opt = FlatSGD(model.parameters(), lr=0.09, momentum=0.9, weight_decay=5e-4, arenas=arenas)
ntx = NTXentLoss(device=dev, batch_size=batch, temperature=0.5, use_cosine_similarity=True)
step = PretrainStep(model, opt, (0.1, 1.0, 1.0, 0.0, 0.0), clip_grad_norm=True, ntxent=ntx, ntxent_weight=1.0)
x1, x2, lab = device_batch(batch, frames, 112, dev, seed=1)


def run(n):
    for _ in range(n):
        step(x1, x2, lab["spa"], lab["tem"], lab["pb"], lab["rot1"], lab["rot2"]).to_host()


run(6)
torch.cuda.synchronize()
tm = AllTimers()
ops.kernel_timer = tm
ops.OVERLAP_WGRAD = False
rb.OVERLAP_TARGET_FORWARD = False
run(1)
tm.enabled = True
nrep = 2
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
run(nrep)
b.record()
torch.cuda.synchronize()
tm.enabled = False
rows = []
for (what, key), pairs in tm.pairs.items():
    t = sum(x.elapsed_time(y) for x, y in pairs) / nrep
    if what.startswith("conv3d"):
        n, c, d, h, w, ko, kt, kh, kw, st, sh, sw, pt, ph, pw = key[:15]
        do, ho, wo = (d + 2 * pt - kt) // st + 1, (h + 2 * ph - kh) // sh + 1, (w + 2 * pw - kw) // sw + 1
        pos = n * do * ho * wo
        flop = 2.0 * pos * ko * c * kt * kh * kw
        nbytes = 4.0 * (n * c * d * h * w + pos * ko + ko * c * kt * kh * kw)
        name = "%s %dx%d->%d k%d%d%d s%d%d%d @%dx%dx%d" % (what[7:].replace("backward_", "d"), n, c, ko, kt, kh, kw, st, sh, sw, d, h, w)
    else:
        n, c, s = key[0], key[1], key[2]
        res = bool(key[4])
        flop = 0.0
        nbytes = 4.0 * n * c * s * ((3 + res) if what == "bn_forward" else (5 + res))
        name = "%s %dx%dx%d g%d res%d relu%d" % (what, n, c, s, key[3], res, key[5])
    rows.append((t, len(pairs) / nrep, name, flop, nbytes))
rows.sort(key=lambda r: -r[0])
tot = sum(r[0] for r in rows)
print("step under the timers: %.2f ms (overlaps off, event pairs add bubbles); spanned calls sum to %.2f ms" % (a.elapsed_time(b) / nrep, tot))
print("%-58s %5s %8s %8s %7s %6s" % ("call site", "n/st", "ms/step", "avg_ms", "TF/s", "TB/s"))
for t, n, name, flop, nbytes in rows:
    avg = t / n
    print("%-58s %5.1f %8.3f %8.4f %7.1f %6.2f" % (name, n, t, avg, flop / avg / 1e9, nbytes / avg / 1e9))
by = {}
for t, n, name, flop, nbytes in rows:
    k = name.split()[0]
    by[k] = by.get(k, 0.0) + t
print({k: round(v, 2) for k, v in sorted(by.items(), key=lambda kv: -kv[1])})
