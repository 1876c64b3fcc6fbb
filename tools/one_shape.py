#!/usr/bin/env python3
"""Time one conv shape (fwd / dgrad / wgrad) in isolation: one_shape.py <name> where name in tools/bench_convs.shapes()."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from bench_convs import shapes, timeit  # noqa: E402
from cstp_amd import ops  # noqa: E402

names = sys.argv[1:] or ["c2.same.T"]
for name, xs, k, ks, st, pd, cnt in shapes(32):
    if name not in names:
        continue
    x = torch.randn(xs, device="cuda")
    w = torch.randn((k, xs[1]) + ks, device="cuda") * 0.05
    y = ops.conv3d(x, w, None, st, pd)
    dy = torch.randn_like(y)
    gf = 2.0 * y.numel() * xs[1] * ks[0] * ks[1] * ks[2] / 1e9
    xr = x.clone().requires_grad_(True)
    t_f = timeit(lambda: ops.conv3d(x, w, None, st, pd), 10)

    def dgrad():
        yy = ops.conv3d(xr, w, None, st, pd)
        yy.backward(dy)
    t_d = timeit(dgrad, 10) - t_f
    print("%-10s fwd %.3f ms %.1f TF | dgrad %.3f ms %.1f TF" % (name, t_f, gf / t_f, t_d, gf / t_d))
